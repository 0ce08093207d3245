#!/usr/bin/env python3
"""Times snp_compress_batch alone on configs[1] (10 GiB html-like) -- for kernel variants whose OUTPUT IS NOT CHECKED
(timing-only ablations selected with SNAPPIER_HIP_LIB).  Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
raw = SD.html_like_blocks(html, 0, nb, "cuda")
in_off, in_len = cd.uniform_layout(nb)
comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
ms = []
for i in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _, _, out_len, status = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print(json.dumps({"lib": os.environ.get("SNAPPIER_HIP_LIB", "default"), "blocks": nb, "compress_ms": [round(m, 2) for m in ms],
                  "GBps": round(nb * 65536 / min(ms) / 1e6, 2), "ratio": round(float(out_len.sum().item()) / (nb * 65536), 4)}))
