#!/usr/bin/env python3
"""Throughput of the HOST-pointer boundary (snp_try_compress / snp_try_decompress / snp_frame_encode / snp_frame_decode),
called through ctypes on preallocated pageable numpy buffers exactly as a P/Invoke caller would: the PCIe transfers are
inside the timed call.  These are the numbers DESIGN.md quotes as "PCIe-inclusive"; they are never bench.py's `value`.
One JSON line per size."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import snappier_amd as S

html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
ctx = S.Context(0, S.HASH_CRC32C)
L = S.lib()
pinned = os.environ.get("PINNED", "0") == "1"
sizes = [int(a) for a in sys.argv[1:]] or [65536, 1 << 20, 16 << 20, 256 << 20, 1 << 30]
vp = lambda a: C.c_void_p(a.ctypes.data)   # noqa: E731
for n in sizes:
    reps = (n + len(html) - 1) // len(html)
    data = np.frombuffer((html * reps)[:n], dtype=np.uint8).copy()
    if n > (1 << 20):                      # make the 64 KiB fragments differ a little, like configs[1]
        rng = np.random.default_rng(7)
        idx = rng.integers(0, n, n // 100)
        data[idx] = rng.integers(0, 256, idx.size, dtype=np.uint8)
    if pinned:                             # page-locked buffers (hipHostMalloc through torch): what snp_host_alloc hands a caller
        hold = [torch.empty(k, dtype=torch.uint8).pin_memory() for k in (n, L.snp_max_compressed_length(n), L.snp_frame_max_encoded_length(n), n)]
        hold[0].numpy()[:] = data
        data, comp, framed, back = (t.numpy() for t in hold)
    else:
        comp = np.empty(L.snp_max_compressed_length(n), dtype=np.uint8)
        framed = np.empty(L.snp_frame_max_encoded_length(n), dtype=np.uint8)
        back = np.empty(n, dtype=np.uint8)
    w = C.c_size_t(0)

    def call(fn, src, sn, dst):
        st = fn(ctx.handle, vp(src), sn, vp(dst), dst.size, C.byref(w))
        assert st == 0, st
        return w.value

    def best(f, k=3):
        ts, r = [], None
        for _ in range(k):
            t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
        return min(ts), r
    t_c, wc = best(lambda: call(L.snp_try_compress, data, n, comp))
    t_d, wd = best(lambda: call(L.snp_try_decompress, comp, wc, back))
    assert wd == n and np.array_equal(back, data)
    back[:] = 0
    t_fe, wf = best(lambda: call(L.snp_frame_encode, data, n, framed), 2)
    t_fd, wb = best(lambda: call(L.snp_frame_decode, framed, wf, back), 2)
    assert wb == n and np.array_equal(back, data)
    print(json.dumps({"bytes": n, "host_buffers": "pinned" if pinned else "pageable", "ratio": round(wc / n, 4), 
                      "compress_ms": round(t_c * 1e3, 3), "compress_GBps": round(n / t_c / 1e9, 3),
                      "decompress_ms": round(t_d * 1e3, 3), "decompress_GBps": round(n / t_d / 1e9, 3),
                      "frame_encode_GBps": round(n / t_fe / 1e9, 3), "frame_decode_GBps": round(n / t_fd / 1e9, 3)}), flush=True)
