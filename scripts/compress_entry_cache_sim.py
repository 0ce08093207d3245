"""CPU simulation (round 6): would a small per-lane WRITE-BACK CACHE of hash-table entries in LDS (the lane compressor has ~120 B of LDS per lane to spare at
10 wavefronts per CU) absorb a useful share of the table accesses?  Trace of (bucket) accesses of the reference parse (oracle/pymodel.py semantics), hit rate of a
direct-mapped cache of N entries indexed by bucket mod N, and of a fully associative LRU one.  Answer: no (html-like: 6.7 % at 32 entries, 10.5 % at 64).
    python scripts/compress_entry_cache_sim.py   -> one JSON line per data set"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import pymodel as M
import datagen
from collections import OrderedDict

def trace(f, variant=M.HASH_CRC32C):
    """sequence of (bucket, is_probe) table accesses of the reference parse"""
    n = len(f); H = M.h_crc if variant == M.HASH_CRC32C else M.h_mul
    ld32 = lambda p: int.from_bytes(f[p:p+4], 'little')
    ts = M.tsize(n); mask = 2*(ts-1); table=[0]*ts; acc=[]
    def probe(p, d):
        h = H(d, mask) >> 1; cand = table[h]; table[h] = p; acc.append((h, 1)); return cand, ld32(cand) == d
    ip = 0
    if n >= 15:
        limit = n - 15
        while True:
            next_emit = ip; ip += 1; skip = 32; found = False; cand = 0
            if limit - ip >= 16:
                for j in range(16):
                    p = ip + j; cand, hit = probe(p, ld32(p))
                    if hit: ip, found = p, True; break
                if not found: ip += 16; skip += 16
            rem = False
            if not found:
                while True:
                    d = ld32(ip); bb = skip >> 5; skip += bb; nxt = ip + bb
                    if nxt > limit: rem = True; break
                    cand, hit = probe(ip, d)
                    if hit: break
                    ip = nxt
            if rem: break
            while True:
                m = 4
                while ip + m < n and f[cand+m] == f[ip+m]: m += 1
                ip += m
                if ip >= limit: rem = True; break
                dm1 = ld32(ip-1); hm1 = H(dm1, mask) >> 1; table[hm1] = ip-1; acc.append((hm1, 0))
                cand, hit = probe(ip, ld32(ip))
                if not hit: break
            if rem: break
    return acc

def sim(acc, size, assoc_lru=False):
    hits = 0
    if assoc_lru:
        c = OrderedDict()
        for h, _ in acc:
            if h in c: hits += 1; c.move_to_end(h)
            else:
                c[h] = 1
                if len(c) > size: c.popitem(last=False)
    else:
        c = [-1]*size
        for h, _ in acc:
            i = h % size
            if c[i] == h: hits += 1
            else: c[i] = h
    return hits / len(acc)

td = os.path.join(ROOT, 'tests', 'golden', 'testdata')
html = open(os.path.join(td, 'html'), 'rb').read()
sets = {'html-like': [bytes(datagen.html_like_blocks(html, b, 1).tobytes()) for b in range(3)]}
for name in ['alice29.txt', 'geo.protodata', 'kppkn.gtb', 'urls.10K']:
    d = open(os.path.join(td, name), 'rb').read(); sets[name] = [d[:65536]]
for name, frs in sets.items():
    acc = []
    for f in frs: acc += trace(f)
    row = {'data': name, 'accesses': len(acc)//len(frs)}
    for size in (16, 32, 64, 128, 256, 1024):
        row['dm%d' % size] = round(sim(acc, size), 3)
    for size in (32, 128, 1024):
        row['lru%d' % size] = round(sim(acc, size, True), 3)
    print(json.dumps(row))
