#!/usr/bin/env python3
"""Many small blocks (the "many tiny chunks" pattern of SnappyStreamTests.cs:145-192): throughput of the
batch API for block sizes 256 B .. 16 KiB, both decoder layouts."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
total = 1 << 30
raw = SD.html_like_blocks(html, 0, total // 65536, "cuda")
cd = SB.BlockCodec(0, S.HASH_CRC32C)
for bs in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096, 16384, 65536]:
    nb = total // bs
    in_off, in_len = cd.uniform_layout(nb, bs)
    stride = (int(S.lib().snp_max_compressed_length(bs)) + 15) // 16 * 16
    comp = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * stride
    back = torch.empty_like(raw)
    def timed(f):
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = f(); b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b), r
    ms_c, (_, _, out_len, st) = timed(lambda: cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off))
    ms_d, (dlen, dst) = timed(lambda: cd.decompress(comp, comp_off, out_len, back, in_off, in_len))
    ok = int((st != 0).sum()) == 0 and int((dst != 0).sum()) == 0 and torch.equal(back[: nb * bs], raw[: nb * bs])
    print(json.dumps({"block_bytes": bs, "blocks": nb, "ok": ok, "ratio": round(float(out_len.sum().item()) / total, 3),
                      "compress_GBps": round(total / ms_c / 1e6, 1), "decompress_GBps": round(total / ms_d / 1e6, 1),
                      "decode_layout": os.environ.get("SNAPPIER_HIP_DECODE", "chains")}), flush=True)
