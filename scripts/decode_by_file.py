#!/usr/bin/env python3
"""Decoder front ends per corpus file: NB blocks of 64 KiB windows of ONE file (tiled + mutated like the mixed-corpus workload), decoded by each
SNAPPIER_HIP_DECODE mode in turn; prints ms, GB/s and output bytes per tag.   python scripts/decode_by_file.py [blocks]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
modes = os.environ.get("MODES", "chains ring").split()
for n in names:
    f = open(os.path.join(td, n), "rb").read()
    raw = SD.corpus_blocks([f], 0, nb, SD.MIXED_SEED, "cuda")
    row = {"file": n, "blocks": nb}
    comp = None
    for m in modes:
        if m == "default":                  # the library's own default, no knob set (the product library when SNAPPIER_HIP_LIB is not set either)
            os.environ.pop("SNAPPIER_HIP_DECODE", None)
        else:
            os.environ["SNAPPIER_HIP_DECODE"] = m
        cd = SB.BlockCodec(0, S.HASH_CRC32C)
        in_off, in_len = cd.uniform_layout(nb)
        if comp is None:
            comp, comp_off, comp_len, st = cd.compress(raw, in_off, in_len)
            row["ratio"] = round(float(comp_len.to(torch.int64).sum().item()) / (nb * 65536), 3)
        back = torch.zeros_like(raw)
        ms = []
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dlen, dst = cd.decompress(comp, comp_off, comp_len, back, in_off, in_len); e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        assert torch.equal(back, raw) and int((dst != 0).sum()) == 0
        row[m] = {"ms": round(min(ms[1:]), 3), "GBps": round(nb * 65536 / min(ms[1:]) / 1e6, 1)}
        del cd
    print(json.dumps(row), flush=True)
