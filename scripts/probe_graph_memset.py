import ctypes as C, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int
for nbytes in (4, 272, 4096, 1 << 20):
    t = torch.ones(max(nbytes // 4, 1) + 16, dtype=torch.int32, device="cuda")
    u = torch.zeros(8, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream()); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        u.add_(1.0)
        rc = hip.hipMemsetAsync(t.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        u.add_(1.0)
    res = []
    for i in range(3):
        t.fill_(7); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        res.append((int((t[: nbytes // 4] != 0).sum()), int((t[nbytes // 4:] != 7).sum())))
    print(nbytes, "rc", rc, "nonzero-after-replay / clobbered-beyond:", res, "u", float(u[0]))
