#!/usr/bin/env python3
"""Why is the decoder 8 % slower inside the bench step (after a compress launch) than in a loop of its own?  Times snp_decompress_batch after
   (a) itself, (b) a compress launch, (c) a 20 GB fill (caches and TLBs swept), (d) a 20 GB fill followed by a 200 MB touch of the decoder's own
   input.  python scripts/decode_cold_warm.py [blocks]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
html = open(os.path.join(td, "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
raw = SD.html_like_blocks(html, 0, nb, "cuda")
in_off, in_len = cd.uniform_layout(nb)
out, out_off, out_len, st = cd.compress(raw, in_off, in_len)
back = torch.zeros_like(raw)
sweep = torch.empty(20 << 30, dtype=torch.uint8, device="cuda")

def decode():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    cd.decompress(out, out_off, out_len, back, in_off, in_len)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1), 2)

res = {"blocks": nb}
decode(); decode()
res["after_itself"] = [decode() for _ in range(3)]
r = []
for _ in range(3):
    cd.compress(raw, in_off, in_len, out=out, out_off=out_off)
    r.append(decode())
res["after_compress"] = r
r = []
for _ in range(3):
    sweep.fill_(1)
    r.append(decode())
res["after_20GB_fill"] = r
r = []
for _ in range(3):
    sweep.fill_(1)
    torch.cuda.synchronize()
    import time; time.sleep(0.2)
    r.append(decode())
res["after_fill_and_200ms_idle"] = r
r = []
for _ in range(3):
    decode()
    torch.cuda.synchronize()
    time.sleep(0.2)
    r.append(decode())
res["after_itself_and_200ms_idle"] = r
print(json.dumps(res))
