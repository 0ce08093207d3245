#!/usr/bin/env python3
"""Throughput of the HOST-pointer boundary (snp_try_compress / snp_try_decompress / snp_frame_encode / snp_frame_decode):
pageable host buffers in and out, so PCIe staging is inside the timed call.  These are the numbers DESIGN.md quotes as
"PCIe-inclusive"; they are never bench.py's `value`.  One JSON line per size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import snappier_amd as S
from snappier_amd import snappy as SP

html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
ctx = S.Context(0, S.HASH_CRC32C)
sizes = [int(a) for a in sys.argv[1:]] or [65536, 1 << 20, 16 << 20, 256 << 20, 1 << 30]
for n in sizes:
    reps = (n + len(html) - 1) // len(html)
    data = np.frombuffer((html * reps)[:n], dtype=np.uint8).copy()
    if n > (1 << 20):                      # make the 64 KiB fragments differ a little, like configs[1]
        rng = np.random.default_rng(7)
        idx = rng.integers(0, n, n // 100)
        data[idx] = rng.integers(0, 256, idx.size, dtype=np.uint8)
    out = np.empty(SP.Snappy.GetMaxCompressedLength(n), dtype=np.uint8)
    back = np.empty(n, dtype=np.uint8)
    def best(f, k=3):
        ts = []
        for _ in range(k):
            t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
        return min(ts), r
    t_c, w = best(lambda: SP.Snappy.Compress(data, out, ctx))
    t_d, w2 = best(lambda: SP.Snappy.Decompress(out[:w], back, ctx))
    assert w2 == n and back.tobytes() == data.tobytes()
    t_fe, framed = best(lambda: SP.frame_encode(data, ctx), 2)
    t_fd, plain = best(lambda: SP.frame_decode(framed, ctx), 2)
    assert plain == data.tobytes()
    print(json.dumps({"bytes": n, "ratio": round(w / n, 4),
                      "compress_ms": round(t_c * 1e3, 3), "compress_GBps": round(n / t_c / 1e9, 3),
                      "decompress_ms": round(t_d * 1e3, 3), "decompress_GBps": round(n / t_d / 1e9, 3),
                      "frame_encode_GBps": round(n / t_fe / 1e9, 3), "frame_decode_GBps": round(n / t_fd / 1e9, 3)}), flush=True)
