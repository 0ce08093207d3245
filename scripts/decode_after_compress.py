#!/usr/bin/env python3
"""Round 6: why is the decode launch that follows a compress launch 7 % slower than a decode launch that follows a decode launch?  Not the clocks (bench line).
Hypothesis: address translations for the 23 GB the launch touches.  Test: between the compress and the decode launch, a kernel that reads ONE byte per
STRIDE bytes of both buffers (no data worth caching: it can only warm translations).   python scripts/decode_after_compress.py   -> one JSON line per variant"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
html = open(os.path.join(ROOT, "tests", "golden", "testdata", "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
cd.ctx.reserve_compress(nb)
raw = SD.html_like_blocks(html, 0, nb, "cuda")
in_off, in_len = cd.uniform_layout(nb)
comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
back = torch.empty_like(raw)
def ev(): return torch.cuda.Event(enable_timing=True)
def run(touch):
    ms = []
    for it in range(5):
        _o, _oo, out_len, st = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
        t0, t1, t2 = ev(), ev(), ev()
        t0.record()
        if touch:
            s1 = comp[::touch].sum(); s2 = back[::touch].sum()
        t1.record()
        dlen, dst = cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
        t2.record()
        torch.cuda.synchronize()
        if it: ms.append((t0.elapsed_time(t1), t1.elapsed_time(t2)))
    return ms
for touch in (0, 2 << 20, 64 << 10, 4 << 10):
    ms = run(touch)
    print(json.dumps({"touch_stride": touch, "touch_ms": round(min(a for a, _ in ms), 3), "decode_ms": [round(b, 3) for _, b in ms]}), flush=True)
# second hypothesis: the compress launch leaves ~290 MB of dirty table sectors in L2 / Infinity Cache; a streaming fill of FLUSH bytes between the launches pushes them out first
scratch = torch.empty(2 << 30, dtype=torch.uint8, device="cuda")
for flush in (256 << 20, 1 << 30, 2 << 30):
    ms = []
    for it in range(5):
        _o, _oo, out_len, st = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
        t0, t1, t2 = ev(), ev(), ev()
        t0.record(); scratch[:flush].zero_(); t1.record()
        cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
        t2.record(); torch.cuda.synchronize()
        if it: ms.append((t0.elapsed_time(t1), t1.elapsed_time(t2)))
    print(json.dumps({"flush_bytes": flush, "flush_ms": round(min(a for a, _ in ms), 3), "decode_ms": [round(b, 3) for _, b in ms]}), flush=True)
assert torch.equal(back, raw)
