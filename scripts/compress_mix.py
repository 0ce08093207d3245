#!/usr/bin/env python3
"""Experiment: a mid-size compress batch split over SEVERAL layouts that run concurrently on their own streams (LDS-table window kernel, global-table
window kernel, lane kernel) -- each is bound by something else (LDS capacity: 5 fragments per CU; the texture path; memory latency), so do they add up?
python scripts/compress_mix.py [blocks...]      SPLITS="1:0:0 0.6:0.4:0 ..." (win : wing : lanes shares)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
counts = [int(a) for a in sys.argv[1:]] or [4096, 16383]
splits = [tuple(float(x) for x in s.split(":")) for s in os.environ.get("SPLITS", "1:0:0 0:1:0 0:0:1 0.5:0.5:0 0.6:0.4:0 0.7:0.3:0 0.5:0.3:0.2 0.4:0.3:0.3 0.6:0:0.4").split()]
html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
cds, streams = {}, {}
for lay in ("win", "wing", "lanes"):
    os.environ["SNAPPIER_HIP_COMPRESS"] = lay
    os.environ["SNAPPIER_HIP_TABLE_TRIES"] = "1"
    streams[lay] = torch.cuda.Stream()
    with torch.cuda.stream(streams[lay]):
        cds[lay] = SB.BlockCodec(0, S.HASH_CRC32C)
for nb in counts:
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
    stride = cds["win"].comp_stride
    comp = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
    in_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
    in_len = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
    comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * stride
    out_len = torch.zeros(nb, dtype=torch.int32, device="cuda")
    ref = None
    for sp in splits:
        n_win = int(round(nb * sp[0])); n_wing = int(round(nb * sp[1])); n_lanes = nb - n_win - n_wing
        parts = [("win", 0, n_win), ("wing", n_win, n_wing), ("lanes", n_win + n_wing, n_lanes)]
        best = 1e9
        for it in range(4):
            out_len.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = []
            for lay, first, cnt in parts:
                if cnt <= 0: continue
                with torch.cuda.stream(streams[lay]):
                    r = cds[lay].compress(raw, in_off[first:first + cnt], in_len[first:first + cnt], out=comp, out_off=comp_off[first:first + cnt])
                    res.append((first, cnt, r))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            if it: best = min(best, dt)
        for first, cnt, r in res:
            out_len[first:first + cnt] = r[2]
            assert int((r[3] != 0).sum()) == 0
        crcs = cds["win"].crc32c(comp, comp_off, out_len)
        sig = (int(out_len.to(torch.int64).sum().item()), int(crcs.to(torch.int64).sum().item()))
        ref = ref or sig
        print(json.dumps({"blocks": nb, "win:wing:lanes": sp, "ms": round(best, 3), "GBps": round(nb * 65536 / best / 1e6, 2), "same_bytes": sig == ref}), flush=True)
