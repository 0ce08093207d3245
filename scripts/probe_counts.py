#!/usr/bin/env python3
"""How many hash-table accesses does the reference parse make per 64 KiB fragment of each corpus file?  (CPU, no GPU.)
The lane compressor's time is its table accesses, so GB/s x accesses per fragment should be one constant across
files -- this script prints the accesses (probes = read + insert, post-copy inserts = write only) so that profiles/r03b_by_file_*.jsonl
can be read against them.  The parse below is the oracle's with counters; its output length is checked against
oracle.compress on every block.     python scripts/probe_counts.py [blocks per file]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as O
import datagen as D
from oracle import pymodel as P

def count(f: bytes):
    n = len(f); ld32 = lambda p: int.from_bytes(f[p:p + 4], "little")
    ts = P.tsize(n); mask = 2 * (ts - 1); table = [0] * ts
    probes = inserts = hits = tokens = 0
    ip = 0; limit = n - 15
    while True:
        next_emit = ip; ip += 1; skip = 32
        while True:
            d = ld32(ip); bb = skip >> 5; skip += bb; nxt = ip + bb      # (the unrolled 16-probe section follows the same sequence)
            if nxt > limit: return probes, inserts, hits
            h = P.h_crc(d, mask); c = table[h]; table[h] = ip; probes += 1
            if ld32(c) == d: cand = c; break
            ip = nxt
        while True:
            hits += 1; m = 4
            while ip + m < n and f[cand + m] == f[ip + m]: m += 1
            ip += m
            if ip >= limit: return probes, inserts, hits
            table[P.h_crc(ld32(ip - 1), mask)] = ip - 1; inserts += 1
            d = ld32(ip); h = P.h_crc(d, mask); cand = table[h]; table[h] = ip; probes += 1
            if ld32(cand) != d: break

nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 2
td = os.path.join(ROOT, "tests", "golden", "testdata")
names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
for nme in names:
    data = open(os.path.join(td, nme), "rb").read()
    blocks = D.corpus_blocks([data], 0, nblk, D.MIXED_SEED)
    pr = ins = hit = clen = 0
    for b in range(nblk):
        f = blocks[b * 65536:(b + 1) * 65536].tobytes()
        p, i, h = count(f)
        pr += p; ins += i; hit += h; clen += len(O.compress(f)) - 3
    print(json.dumps({"file": nme, "blocks": nblk, "probes_per_fragment": pr // nblk, "post_copy_inserts_per_fragment": ins // nblk,
                      "table_accesses_per_fragment": (pr + ins) // nblk, "copies_per_fragment": hit // nblk, "ratio": round(clen / nblk / 65536, 3)}), flush=True)
