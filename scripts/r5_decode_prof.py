#!/usr/bin/env python3
"""Per-phase shader-clock budget of k_decode_chains (a -DSNP_DC_PROF=1 variant: scripts/build_variant.sh prof -DSNP_DC_PROF=1).
   SNAPPIER_HIP_LIB=snappier_amd/variants/libsnappier_hip_prof.so DATA=html python scripts/r5_decode_prof.py [blocks]
   Every wavefront adds the cycles (s_memtime, 100 MHz-independent shader clock) it spent per phase; the sums divided by the number of
   wavefronts give the mean chain of ONE wavefront, which is what bounds the kernel (time is linear in wavefronts per SIMD: r05_decode_occupancy.jsonl)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD, _native as N

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
kind = os.environ.get("DATA", "html")
td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
html = open(os.path.join(td, "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
if kind == "html":
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
elif kind == "low":
    raw = SD.low_entropy_blocks(0, nb, "cuda")
else:
    names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
    raw = SD.corpus_blocks([open(os.path.join(td, n), "rb").read() for n in names if os.path.exists(os.path.join(td, n))], 0, nb, SD.MIXED_SEED, "cuda")
in_off, in_len = cd.uniform_layout(nb)
out, out_off, out_len, st = cd.compress(raw, in_off, in_len)
back = torch.zeros_like(raw)
L = cd.ctx.lib
prof = (C.c_ulonglong * 16)()
L.snp_debug_decode_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
ms = []
for i in range(3):
    torch.cuda.synchronize()
    L.snp_debug_decode_prof(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    cd.decompress(out, out_off, out_len, back, in_off, in_len)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
L.snp_debug_decode_prof(prof, 0)
ok = bool(torch.equal(back, raw))
names = {0: "glue (loop top, exits)", 1: "S stage the window, advance table", 2: "A walk", 3: "A' overrun walk", 4: "R reach / resolve", 5: "T tag list",
         6: "batch top (tag bytes, decode, prefix sum, checks)", 14: "new batch: classes, fence, tag-byte prefetch, far / literal loads ISSUED",
         8: "held batch: finish (in order)", 13: "new batch: loads waited for (vmcnt(0))", 9: "held batch: write-out",
         7: "new batch: near copies from the stage, history, stage stores, hold", 10: "drain before a window build / exit (finish + write-out)"}
order = (0, 1, 2, 3, 4, 5, 6, 14, 8, 13, 9, 7, 10)
tot = sum(prof[i] for i in names)
rows = [{"phase": names[i], "cycles_per_block": round(prof[i] / nb), "share": round(prof[i] / tot, 4)} for i in order]
print(json.dumps({"data": kind, "blocks": nb, "ms": [round(m, 2) for m in ms], "roundtrip_ok": ok, "batches_per_block": round(prof[11] / nb, 1),
                  "windows_per_block": round(prof[12] / nb, 2), "drains_per_block": round(prof[15] / nb, 2), "cycles_per_block": round(tot / nb), "cycles_per_batch": round(tot / max(prof[11], 1)), "phases": rows}))
