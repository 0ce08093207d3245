#!/usr/bin/env python3
"""Small-block compress (SURVEY 8f.4 target: 60 GB/s at 256 B): launch-shape sweep of the lane compressor on 1 GiB cut into equal
blocks -- probes per trip (1 = atomic-exchange probes, 2 = two speculative load/store probes), fragments per wavefront, fragments
per launch.  One JSON line per point; every point's bytes are checked against the first one's (CRC of each block's valid bytes).
   python scripts/small_compress_sweep.py 256 1024"""
import itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
total = 1 << 30
raw = SD.html_like_blocks(html, 0, total // 65536, "cuda")
for bs in [int(a) for a in sys.argv[1:]] or [256, 1024]:
    nb = total // bs
    stride = (int(S.lib().snp_max_compressed_length(bs)) + 15) // 16 * 16
    comp = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * stride
    ref = None
    for slice_, slots, lanes in itertools.product(["65536", "262144"], ["1", "2"], ["16", "32", "64"]):
        os.environ["SNAPPIER_HIP_SLICE"] = slice_
        os.environ["SNAPPIER_HIP_CL_SLOTS"] = slots
        os.environ["SNAPPIER_HIP_LANES_PER_WAVE"] = lanes
        cd = SB.BlockCodec(0, S.HASH_CRC32C)
        in_off, in_len = cd.uniform_layout(nb, bs)
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); _, _, out_len, st = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        crcs = cd.crc32c(comp, comp_off, out_len).to(torch.int64)
        sig = (int(out_len.to(torch.int64).sum().item()), int(crcs.sum().item()), int((crcs * (torch.arange(nb, device="cuda") % 251 + 1)).sum().item()))
        ref = ref or sig
        print(json.dumps({"block_bytes": bs, "slice": int(slice_), "probes_per_trip": int(slots), "fragments_per_wavefront": int(lanes),
                          "compress_ms": round(best, 2), "compress_GBps": round(total / best / 1e6, 1), "same_bytes": sig == ref and int((st != 0).sum()) == 0}), flush=True)
        del cd
