#!/usr/bin/env python3
"""Round 6 experiment: the two table forms of the per-wavefront compressor SIDE BY SIDE on two streams, with the global-slot form limited to so few
wavefronts that its 32 KiB slots stay in L2 (4 MiB per XCD): does it add to the LDS form's 4 wavefronts per CU instead of sharing its bound?
    python scripts/compress_mix2.py [blocks...]     SLOTS="512 768 1024 2048 8192"  SHARES="0 0.2 0.3 0.4 0.5 1"   (share = fraction given to the global-slot form)
One JSON line per (blocks, slots, share); bytes verified against the LDS form's."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD, _native as N
counts = [int(a) for a in sys.argv[1:]] or [8192]
slots_list = [int(x) for x in os.environ.get("SLOTS", "512 768 1024 2048 8192").split()]
shares = [float(x) for x in os.environ.get("SHARES", "0 0.2 0.3 0.4 0.5 1").split()]
kind = os.environ.get("DATA", "html")
html = open(os.path.join(ROOT, "tests", "golden", "testdata", "html"), "rb").read()
st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(st_a):
    win = SB.BlockCodec(0, S.HASH_CRC32C); win.ctx.set_option(N.OPT_COMPRESS_LAYOUT, N.COMPRESS_WINDOW_LDS)
with torch.cuda.stream(st_b):
    wing = SB.BlockCodec(0, S.HASH_CRC32C); wing.ctx.set_option(N.OPT_COMPRESS_LAYOUT, N.COMPRESS_WINDOW_GLOBAL)
for nb in counts:
    raw = SD.html_like_blocks(html, 0, nb, "cuda") if kind == "html" else SD.low_entropy_blocks(0, nb, "cuda")
    stride = win.comp_stride
    in_off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
    in_len = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
    comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * stride
    ref, _, ref_len, _ = win.compress(raw, in_off, in_len)
    torch.cuda.synchronize()
    comp = torch.empty_like(ref)
    for slots in slots_list:
        wing.ctx.set_option(N.OPT_COMPRESS_WINDOW_GLOBAL_SLOTS, slots)
        for share in shares:
            n_g = int(round(nb * share)); n_w = nb - n_g
            best = 1e9
            for it in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lens = []
                if n_w:
                    with torch.cuda.stream(st_a):
                        lens.append(win.compress(raw, in_off[:n_w], in_len[:n_w], out=comp, out_off=comp_off[:n_w])[2])
                if n_g:
                    with torch.cuda.stream(st_b):
                        lens.append(wing.compress(raw, in_off[n_w:], in_len[n_w:], out=comp, out_off=comp_off[n_w:])[2])
                torch.cuda.synchronize()
                if it: best = min(best, time.perf_counter() - t0)
            ol = torch.cat(lens)
            same = bool(torch.equal(ol, ref_len)) and bool(torch.equal(win.compact(comp, comp_off, ol)[0], win.compact(ref, comp_off, ref_len)[0]))
            print(json.dumps({"data": kind, "blocks": nb, "global_slots": slots, "share_global": share, "ms": round(best * 1e3, 3), "GBps": round(nb * 65536 / best / 1e9, 2), "same_bytes": same}), flush=True)
            if share in (0.0,) and slots != slots_list[0]:
                pass
