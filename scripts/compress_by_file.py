#!/usr/bin/env python3
"""Why does the mixed corpus compress slower than html?  The lane compressor on 163 840 blocks cut from ONE corpus file at a time
(same offset / mutation scheme as config 5), and on the mix: ms, GB/s, compressed ratio per file.  One JSON line per file.
   python scripts/compress_by_file.py [blocks]          SNAPPIER_HIP_CL_OPTS etc. apply per launch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
def rd(n):
    return open(os.path.join(td, "html"), "rb").read() * 4 if n == "html_x_4" else open(os.path.join(td, n), "rb").read()
files = {n: rd(n) for n in names}
cd = SB.BlockCodec(0, S.HASH_CRC32C)
in_off, in_len = cd.uniform_layout(nb)
comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
back = None
for label, fl in [(n, [files[n]]) for n in names] + [("MIXED", [files[n] for n in names])]:
    raw = SD.corpus_blocks(fl, 0, nb, SD.MIXED_SEED, "cuda")
    ms = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _, _, out_len, st = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    dms = []
    back = torch.empty_like(raw) if back is None else back
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dlen, dst = cd.decompress(comp, comp_off, out_len, back, in_off, in_len); e1.record(); torch.cuda.synchronize()
        dms.append(e0.elapsed_time(e1))
    ok = bool(torch.equal(back, raw)) and int((st != 0).sum()) == 0 and int((dst != 0).sum()) == 0
    ol = out_len.to(torch.int64)
    print(json.dumps({"file": label, "blocks": nb, "compress_ms": round(min(ms), 2), "compress_GBps": round(nb * 65536 / min(ms) / 1e6, 1),
                      "decompress_ms": round(min(dms), 2), "decompress_GBps": round(nb * 65536 / min(dms) / 1e6, 1),
                      "ratio": round(float(ol.sum().item()) / (nb * 65536), 3), "max_block": int(ol.max().item()), "min_block": int(ol.min().item()), "roundtrip_ok": ok}), flush=True)
    del raw
