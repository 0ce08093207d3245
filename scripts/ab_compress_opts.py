#!/usr/bin/env python3
"""Same-process A/B of the lane compressor's runtime options (SNAPPIER_HIP_CL_OPTS is read per launch): identical
workspace placement, launches interleaved.  Usage: ab_compress_opts.py 7 5 3 1  -> mean ms per option mask."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
# arguments: option masks, or NAME=VALUE pairs of any per-launch environment knob (e.g. SNAPPIER_HIP_LANES_PER_WAVE=32)
masks = sys.argv[1:] or ["7", "5", "3", "1"]
nb = int(os.environ.get("NB", "163840"))
html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
kind = os.environ.get("DATA", "html")
if kind == "low_entropy":
    raw = SD.low_entropy_blocks(0, nb, "cuda")
elif kind == "mixed":
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
    from conftest import CORPUS
    TD = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
    files = [open(os.path.join(TD, "html"), "rb").read() * 4 if n == "html_x_4" else open(os.path.join(TD, n), "rb").read() for n in CORPUS]
    raw = SD.corpus_blocks(files, 0, nb, SD.MIXED_SEED, "cuda")
else:
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
in_off, in_len = cd.uniform_layout(nb)
comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
def run():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), r
run()
res = {m: [] for m in masks}
ref = None
for rep in range(5):
    for m in masks:
        if "=" in m:
            k, v = m.split("=", 1)
            os.environ[k] = v
        else:
            os.environ["SNAPPIER_HIP_CL_OPTS"] = m
        ms, (_, _, out_len, st) = run()
        res[m].append(round(ms, 2))
        # every mask must produce the same BYTES: total length + a CRC-32C of each block's valid bytes (computed on the device)
        crcs = cd.crc32c(comp, comp_off, out_len)
        sig = (int(out_len.to(torch.int64).sum().item()), int(crcs.to(torch.int64).sum().item()), int((crcs.to(torch.int64) * (torch.arange(nb, device="cuda") % 251 + 1)).sum().item()))
        ref = sig if ref is None else ref
        assert sig == ref and int((st != 0).sum()) == 0, (m, sig, ref)
print(json.dumps({m: {"ms": v, "mean": round(sum(v) / len(v), 2)} for m, v in res.items()}))
