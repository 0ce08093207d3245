#!/usr/bin/env python3
"""Soak: compress batches of RANDOM size (1 .. 40 000 fragments: every layout boundary of layout 0 gets crossed) and ragged fragment lengths through ONE
context, layout 0, every block compared with the oracle; then decompressed back.   python scripts/soak_compress_sizes.py [iterations=24] [seed=1]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle as O
import snappier_amd as S
from snappier_amd import batch as SB
import test_gpu_fuzz as F
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
td = os.path.join(ROOT, "tests", "golden", "testdata")
text = np.frombuffer(open(os.path.join(td, "html"), "rb").read() + open(os.path.join(td, "alice29.txt"), "rb").read() + open(os.path.join(td, "geo.protodata"), "rb").read(), dtype=np.uint8)
cd = SB.BlockCodec(0, S.HASH_CRC32C)
total = 0
sizes = [1, 2, 3, 1023, 1024, 1025, 1535, 1536, 1537, 4095, 4096, 32767, 32768]
for it in range(iters):
    nb = sizes[it] if it < len(sizes) else int(rng.integers(1, 40001))
    base = [F.make_block(rng, text) for _ in range(min(nb, 512))]
    blocks = [base[i % len(base)] if i < len(base) else np.roll(base[i % len(base)], i) for i in range(nb)]
    data, off, lens = F.batch_of(blocks)
    total += F._compare_batch(cd, data, off, lens, O.HASH_CRC32C, f"iteration {it}: {nb} fragments")
    print(json.dumps({"iteration": it, "fragments": nb, "bytes": int(lens.sum()), "result": "all equal, and back"}), flush=True)
print(json.dumps({"iterations": iters, "seed": seed, "fragments_compared": total, "result": "all equal to the oracle, and back"}))
