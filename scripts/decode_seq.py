#!/usr/bin/env python3
"""compress, then three decode launches, three times over (for rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace: cycles and duration per dispatch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
nb = 163840
html = open(os.path.join(ROOT, "tests", "golden", "testdata", "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
raw = SD.html_like_blocks(html, 0, nb, "cuda")
in_off, in_len = cd.uniform_layout(nb)
comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
back = torch.empty_like(raw)
for rep in range(3):
    _o, _oo, out_len, st = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
    for k in range(3):
        cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
torch.cuda.synchronize()
assert torch.equal(back, raw)
