#!/usr/bin/env python3
"""Compress rate by batch size and layout on ONE box: for each fragment count, each SNP_OPT_COMPRESS_LAYOUT (auto, win, wing, wind = dual, lanes) in turn (product library);
every layout's bytes are checked against the first one's (length sum + CRC sum).  python scripts/compress_by_batch.py [counts...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import layouts as LAY
from snappier_amd import _native as N
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
counts = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16383, 16384, 32768, 65536]
layouts = os.environ.get("LAYOUTS", "auto win wing wind lanes").split()
html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
for nb in counts:
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
    row = {"blocks": nb}
    ref = None
    for lay in layouts:
        cd = SB.BlockCodec(0, S.HASH_CRC32C)
        LAY.set_compress_layout(cd.ctx, lay)
        if lay == "lanes":
            cd.ctx.set_option(N.OPT_TABLE_PROBE_TRIES, 1)
        in_off, in_len = cd.uniform_layout(nb)
        comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
        comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
        ms = []
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _, _, out_len, st = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off); e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        crcs = cd.crc32c(comp, comp_off, out_len)
        sig = (int(out_len.to(torch.int64).sum().item()), int(crcs.to(torch.int64).sum().item()), int((st != 0).sum()))
        ref = ref or sig
        row[lay] = {"ms": round(min(ms[1:]), 3), "GBps": round(nb * 65536 / min(ms[1:]) / 1e6, 2), "same_bytes": sig == ref}
        del cd
    print(json.dumps(row), flush=True)
