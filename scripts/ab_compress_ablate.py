#!/usr/bin/env python3
"""Timing-only ablations of k_compress_lanes on ONE workspace, launches interleaved (needs a library built with -DSNP_CL_ABLATE_RT=1:
SNAPPIER_HIP_LIB=...; SNAPPIER_HIP_CL_ABLATE is read per launch).  What owns the time above the exchange-only floor?
  1   the ip - 1 insert after a copy is not stored (the parse drifts: later probes miss what it would have put there)
  2   the candidate / probe / literal loads of a trip issued twice (a second set 64 bytes away): time(2) - time(0) = their cost
  4   the staged output runs are not stored
  8   the input-window reload issued twice
  32  one more exchange + store per probe elsewhere in the lane's table (net effect none): time(32) - time(0) = cost of 1.63 G more
      random exchanges + 1.63 G more random stores
  64 / 128  the ip - 1 insert as a non-temporal / an agent-scope (sc1) store -- experiments, results unchanged
  256 / 512  the probe exchange at workgroup / wavefront scope instead of agent scope (the table is lane-private) -- experiments, results unchanged
  8192  (round 5) the tables are zeroed TWICE at the head of the kernel: time(8192) - time(0) = what zeroing costs, the most an epoch tag in
        the entries could save (results unchanged)
  1024 / 2048 / 4096  the exchange through inline asm (waited for at once) with sc0 | sc0 nt | sc0 sc1 -- compare these three with each other
Masks 0, 2, 8, 32, 64, 128, 256, 512, 1024, 2048, 4096 must produce the reference bytes (checked); 1 and 4 are wrong by construction.
   python scripts/ab_compress_ablate.py [masks...]   DATA=html|mixed   ->  one JSON line"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
masks = [int(m) for m in sys.argv[1:]] or [0, 1, 2, 4, 8, 32, 5]
nb = int(os.environ.get("NB", "163840"))
td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
html = open(os.path.join(td, "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
if not os.environ.get("NO_RESERVE"):
    cd.ctx.reserve_compress(nb)
kind = os.environ.get("DATA", "html")
if kind == "mixed":
    names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
    raw = SD.corpus_blocks([html * 4 if n == "html_x_4" else open(os.path.join(td, n), "rb").read() for n in names], 0, nb, SD.MIXED_SEED, "cuda")
else:
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
in_off, in_len = cd.uniform_layout(nb)
comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
def run(mask):
    os.environ["SNAPPIER_HIP_CL_ABLATE"] = str(mask)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), r
run(0); run(0)
res = {m: [] for m in masks}
total = {}
ref = None
for rep in range(int(os.environ.get("REPS", "4"))):
    for m in (masks if rep % 2 == 0 else masks[::-1]):
        ms, (_, _, out_len, st) = run(m)
        res[m].append(round(ms, 2))
        tot = int(out_len.to(torch.int64).sum().item())
        total[m] = tot
        if m in (0, 2, 8, 32, 10, 34, 40, 42, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
            crcs = cd.crc32c(comp, comp_off, out_len)
            sig = (tot, int(crcs.to(torch.int64).sum().item()))
            ref = sig if ref is None else ref
            assert sig == ref and int((st != 0).sum()) == 0, (m, sig, ref)
base = sum(res[0]) / len(res[0]) if 0 in res else None
print(json.dumps({"data": kind, "blocks": nb, "workspace_search": {"candidates": S.lib().snp_ctx_counter(cd.ctx.handle, 3), "seconds": S.lib().snp_ctx_counter(cd.ctx.handle, 4) / 1e6},
                  "ablations": {str(m): {"ms": v, "mean": round(sum(v) / len(v), 2), "delta_vs_0": (round(sum(v) / len(v) - base, 2) if base else None),
                                         "compressed_bytes": total[m]} for m, v in res.items()}}))
