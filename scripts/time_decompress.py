#!/usr/bin/env python3
"""Times snp_decompress_batch alone (kernel variants via SNAPPIER_HIP_DECODE / SNAPPIER_HIP_LIB) and checks the round trip.
   python scripts/time_decompress.py [blocks]      DATA=html|low|mixed.  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

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
# PAD_GIB=n: n GiB allocated first, so that the output buffer lands elsewhere in device memory (the decoder does not care:
# 10.85-10.94 ms with 0-160 GiB of padding, profiles/r03y_decode_placement.jsonl -- unlike the compressor's hash tables)
pads = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(int(os.environ.get("PAD_GIB", "0")))]
back = torch.zeros_like(raw)
ms = []
for i in range(int(os.environ.get("REPS", "4"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
ok = bool(torch.equal(back, raw)) and int((dst != 0).sum()) == 0
print(json.dumps({"decode": os.environ.get("SNAPPIER_HIP_DECODE", "default"), "data": kind, "blocks": nb, "decompress_ms": [round(m, 2) for m in ms],
                  "GBps": round(nb * 65536 / min(ms) / 1e6, 1), "roundtrip_ok": ok}))
