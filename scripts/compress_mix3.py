#!/usr/bin/env python3
"""Round 6 experiment: the headline batch (163 840 fragments) split between the lane compressor and the dual per-wavefront form, on two streams:
does the dual form's population find room beside 10 lane wavefronts per CU that saturate the memory system?   SHARES="0 0.02 0.04 0.06 0.08 0.1"
One JSON line per share (fraction of the fragments given to the dual form); bytes verified against the lane kernel's."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD, _native as N
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
shares = [float(x) for x in os.environ.get("SHARES", "0 0.02 0.04 0.06 0.08 0.1").split()]
html = open(os.path.join(ROOT, "tests", "golden", "testdata", "html"), "rb").read()
st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(st_a):
    lanes = SB.BlockCodec(0, S.HASH_CRC32C); lanes.ctx.set_option(N.OPT_COMPRESS_LAYOUT, N.COMPRESS_LANES)
    lanes.ctx.reserve_compress(nb)
with torch.cuda.stream(st_b):
    dual = SB.BlockCodec(0, S.HASH_CRC32C); dual.ctx.set_option(N.OPT_COMPRESS_LAYOUT, {"dual": N.COMPRESS_WINDOW_DUAL, "win": N.COMPRESS_WINDOW_LDS, "wing": N.COMPRESS_WINDOW_GLOBAL}[os.environ.get("FORM", "dual")])
    if os.environ.get("SLOTS"): dual.ctx.set_option(N.OPT_COMPRESS_WINDOW_GLOBAL_SLOTS, int(os.environ["SLOTS"]))
kind = os.environ.get("DATA", "html")
if kind == "html":
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
elif kind == "low":
    raw = SD.low_entropy_blocks(0, nb, "cuda")
else:
    td = os.path.join(ROOT, "tests", "golden", "testdata")
    names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
    raw = SD.corpus_blocks([(html * 4 if n == "html_x_4" else open(os.path.join(td, n), "rb").read()) for n in names], 0, nb, SD.MIXED_SEED, "cuda")
stride = lanes.comp_stride
in_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
in_len = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * stride
with torch.cuda.stream(st_a):
    ref, _, ref_len, _ = lanes.compress(raw, in_off, in_len)
torch.cuda.synchronize()
comp = torch.empty_like(ref)
for share in shares:
    n_d = int(round(nb * share)) // 64 * 64; n_l = nb - n_d
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lens = []
        with torch.cuda.stream(st_a):
            lens.append(lanes.compress(raw, in_off[:n_l], in_len[:n_l], out=comp, out_off=comp_off[:n_l])[2])
        if n_d:
            with torch.cuda.stream(st_b):
                lens.append(dual.compress(raw, in_off[n_l:], in_len[n_l:], out=comp, out_off=comp_off[n_l:])[2])
        torch.cuda.synchronize()
        if it: best = min(best, time.perf_counter() - t0)
    ol = torch.cat(lens)
    same = bool(torch.equal(ol, ref_len))
    print(json.dumps({"data": kind, "form": os.environ.get("FORM", "dual"), "blocks": nb, "share_dual": share, "ms": round(best * 1e3, 2), "GBps": round(nb * 65536 / best / 1e9, 2), "same_lengths": same}), flush=True)
