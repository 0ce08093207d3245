"""Compress throughput of both layouts vs batch size (sets the auto-selection threshold in capi_batch.hip)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
html = open("tests/golden/testdata/html", "rb").read()
res = {}
for layout in ("win", "lanes64", "lanes32", "lanes16", "lanes8"):
    os.environ["SNAPPIER_HIP_COMPRESS"] = layout[:5] if layout.startswith("lanes") else layout
    if layout.startswith("lanes"):
        os.environ["SNAPPIER_HIP_LANES_PER_WAVE"] = layout[5:]
    import snappier_amd as S
    from snappier_amd import batch as SB, datagen as SD
    cd = SB.BlockCodec(0, S.HASH_CRC32C)
    for nb in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 163840):
        raw = SD.html_like_blocks(html, 0, nb, "cuda")
        in_off, in_len = cd.uniform_layout(nb)
        out = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
        oo = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
        cd.compress(raw, in_off, in_len, out=out, out_off=oo); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(2): cd.compress(raw, in_off, in_len, out=out, out_off=oo)
        b.record(); torch.cuda.synchronize()
        res.setdefault(layout, {})[nb] = round(nb * 65536 * 2 / a.elapsed_time(b) / 1e6, 2)
print(json.dumps(res))
