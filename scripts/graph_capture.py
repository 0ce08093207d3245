#!/usr/bin/env python3
"""Are the batch entry points capturable into a hipGraph, and what does replay save for small batches?  For each batch size: compress + decompress
called directly (events around the pair) and replayed from one captured graph; results of the replay compared with the direct call's.
python scripts/graph_capture.py [blocks...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
counts = [int(a) for a in sys.argv[1:]] or [16, 256, 4096]
html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
for nb in counts:
    cd = SB.BlockCodec(0, S.HASH_CRC32C)
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
    in_off, in_len = cd.uniform_layout(nb)
    comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
    comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
    back = torch.zeros_like(raw)
    row = {"blocks": nb}

    what = os.environ.get("WHAT", "both")
    state = {}

    def pair():
        if what != "decompress" or "len" not in state:
            _, _, out_len, st = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
            state["len"], state["st"] = out_len, st
        out_len, st = state["len"], state["st"]
        if what != "compress":
            dlen, dst = cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
        else:
            dlen, dst = cd.decompress(comp, comp_off, out_len, back, in_off, in_len) if "d" not in state else state["d"]
            state["d"] = (dlen, dst)
        return out_len, st, dlen, dst

    def timed(f, reps=20):
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best
    import time
    cold = os.environ.get("COLD", "0") == "1"                             # COLD=1: the capture follows the context's very first call
    if not cold:
        for _ in range(3):
            out_len, st, dlen, dst = pair()                              # warm: workspaces allocated, layout hints settled
        torch.cuda.synchronize()
        assert torch.equal(back, raw)
        row["direct_ms"] = round(timed(pair), 4)
        t0 = time.perf_counter()
        for _ in range(50): pair()
        torch.cuda.synchronize()
        row["direct_wall_ms"] = round((time.perf_counter() - t0) * 1e3 / 50, 4)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            out_len, st, dlen, dst = pair()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ref_len = out_len.clone(); ref_comp = comp.clone()
        with torch.cuda.graph(g, stream=s):
            g_len, g_st, g_dlen, g_dst = pair()
        if what == "both": comp.zero_()
        if what != "compress": back.zero_()
        g.replay(); torch.cuda.synchronize()
        row["replay_same_results"] = bool(torch.equal(back, raw) and torch.equal(g_len, ref_len) and torch.equal(comp, ref_comp) and int((g_st != 0).sum()) == 0 and int((g_dst != 0).sum()) == 0)
        row["replay_ms"] = round(timed(g.replay), 4)
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        row["replay_wall_ms"] = round((time.perf_counter() - t0) * 1e3 / 50, 4)
        # a replay reads the buffers as they are at replay time: other contents, same shape
        other = SD.html_like_blocks(html, 5 * nb, nb, "cuda")
        keep = raw.clone()
        raw.copy_(other); comp.zero_(); back.zero_()
        g.replay(); torch.cuda.synchronize()
        row["replay_other_contents_ok"] = bool(torch.equal(back, other))
        raw.copy_(keep)
        # and the context still works outside the graph afterwards
        back.zero_(); pair(); torch.cuda.synchronize()
        row["direct_after_capture_ok"] = bool(torch.equal(back, raw))
    except Exception as e:                                                # noqa: BLE001
        row["capture_error"] = repr(e)[:300]
        row["last_error"] = cd.ctx.last_error() if hasattr(cd.ctx, "last_error") else None
    print(json.dumps(row), flush=True)
