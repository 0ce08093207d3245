#!/usr/bin/env python3
"""snp_crc32c_batch alone: 163 840 x 64 KiB chunks (configs[3]'s CRC leg), HIP-event time, fraction of the HBM peak.
Algorithmic bytes = U (every byte read once)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
html = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata", "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
raw = SD.html_like_blocks(html, 0, nb, "cuda")
in_off, in_len = cd.uniform_layout(nb)
table_free = int(os.environ.get("TABLE_FREE", "0"))      # SNP_OPT_CRC_KERNEL: 0 default (three tables), 1 table-free, 2 four 8-bit tables
cd.ctx.set_option(S._native.OPT_CRC_KERNEL, table_free)
ms = []
for i in range(int(os.environ.get('REPS', '12'))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    crc = cd.crc32c(raw, in_off, in_len, masked=True)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
t = sorted(ms[1:])[len(ms[1:]) // 2]                     # median
print(json.dumps({"kernel": {0: "k_crc32c<11+11+10-bit tables>", 1: "k_crc32c<table_free>", 2: "k_crc32c<8-bit tables>"}[table_free], "chunks": nb, "ms": [round(m, 3) for m in ms], "GBps": round(nb * 65536 / t / 1e6, 1),
                  "frac_of_8TBps": round(nb * 65536 / t / 1e6 / 8000, 4), "checksum_of_checksums": int(crc.to(torch.int64).sum().item())}))
