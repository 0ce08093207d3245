#!/usr/bin/env python3
"""Same-process, interleaved A/B of kernel VARIANT LIBRARIES (scripts/build_variant.sh) against the product library: one process, one set of buffers,
launches of the libraries alternating, so that box-to-box and process-to-process drift (5-6 %) cancels.
    python scripts/ab_libs.py product snappier_amd/variants/libsnappier_hip_x.so ...      MODE=decode|compress  DATA=html,low,mixed  NB=163840  REPS=6
    LAYOUT=win|wing|lanes (compress: SNP_OPT_COMPRESS_LAYOUT through tests/layouts.py)
Every library's output is verified (round trip; compress: bytes equal to the product library's).  One JSON line per (data, library)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD, _native as N
import layouts

libs = sys.argv[1:] or ["product"]
mode = os.environ.get("MODE", "decode")
nb = int(os.environ.get("NB", "163840"))
reps = int(os.environ.get("REPS", "6"))
layout = os.environ.get("LAYOUT", "")
td = os.path.join(ROOT, "tests", "golden", "testdata")
html = open(os.path.join(td, "html"), "rb").read()

def codec_of(path):
    if path != "product":
        N._lib = N._load(os.path.abspath(path))      # the next Context binds to this library and keeps it
    else:
        N._lib = N._load(N.LIB_PATH)
    cd = SB.BlockCodec(0, S.HASH_CRC32C)
    if layout:
        layouts.set_compress_layout(cd.ctx, layout)
    if os.environ.get("SLOTS"):
        cd.ctx.set_option(N.OPT_COMPRESS_WINDOW_GLOBAL_SLOTS, int(os.environ["SLOTS"]))
    return cd

cds = [(p, codec_of(p)) for p in libs]
for kind in os.environ.get("DATA", "html").split(","):
    if kind == "html":
        raw = SD.html_like_blocks(html, 0, nb, "cuda")
    elif kind == "low":
        raw = SD.low_entropy_blocks(0, nb, "cuda")
    else:
        names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
        raw = SD.corpus_blocks([(html * 4 if n == "html_x_4" else open(os.path.join(td, n), "rb").read()) for n in names], 0, nb, SD.MIXED_SEED, "cuda")
    ref = cds[0][1]
    in_off, in_len = ref.uniform_layout(nb)
    out, out_off, out_len, st = ref.compress(raw, in_off, in_len)
    torch.cuda.synchronize()
    ms = {p: [] for p, _ in cds}
    ok = {p: True for p, _ in cds}
    if mode == "decode":
        back = torch.zeros_like(raw)
        for r in range(reps + 1):
            for p, cd in cds:
                back.zero_() if r == 0 else None
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len); e1.record()
                torch.cuda.synchronize()
                if r: ms[p].append(e0.elapsed_time(e1))
                else: ok[p] = bool(torch.equal(back, raw)) and int((dst != 0).sum()) == 0
    else:
        out2 = torch.empty_like(out)
        for r in range(reps + 1):
            for p, cd in cds:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); _o, _oo, ol2, st2 = cd.compress(raw, in_off, in_len, out=out2, out_off=out_off); e1.record()
                torch.cuda.synchronize()
                if r: ms[p].append(e0.elapsed_time(e1))
                else:
                    same_len = bool(torch.equal(ol2, out_len)) and int((st2 != 0).sum()) == 0
                    # bytes: compare the compacted streams
                    a, _ = ref.compact(out, out_off, out_len); b, _ = ref.compact(out2, out_off, ol2)
                    ok[p] = same_len and bool(torch.equal(a, b))
    for p, _ in cds:
        v = sorted(ms[p])
        print(json.dumps({"mode": mode, "layout": layout or "auto", "data": kind, "blocks": nb, "lib": os.path.basename(p), "verified": ok[p], "min_ms": round(v[0], 3),
                          "median_ms": round(v[len(v) // 2], 3), "all_ms": [round(x, 2) for x in ms[p]], "GBps_min": round(nb * 65536 / v[0] / 1e6, 1)}), flush=True)
    del raw, out
