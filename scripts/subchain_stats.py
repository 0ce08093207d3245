#!/usr/bin/env python3
"""CPU statistics model of the sub-chain tag parse of k_decompress<.., FRONT = 3> (HISTORY.md §4.1c): how often the guessed
chains of a super-window merge with the true chain, how many loop trips the phases take, how many tokens a super-window
yields.  Pure Python over oracle-compressed corpus blocks; prints one JSON line per (file, R, run-in).  Design aid only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as O

TD = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")


def adv(buf, p):
    c = buf[p]
    t = c & 3
    hi6 = c >> 2
    if t == 0:
        if hi6 >= 60:
            ex = hi6 - 59
            ln = int.from_bytes(buf[p + 1:p + 1 + ex], "little") + 1
        else:
            ex, ln = 0, hi6 + 1
        return 1 + ex + min(ln, 0x40000000)
    return 1 + (4 if t == 3 else t)


def simulate(comp, R, runin, cap):
    n = len(comp)
    # skip varint
    ip = 0
    while comp[ip] & 0x80:
        ip += 1
    ip += 1
    W = 64 * R
    st = dict(windows=0, tokens=0, a_trips=0, o_trips=0, unsynced=0, slow_tags=0, active=0, lanes=0, bytes=0)
    buf = comp + bytes(W + 80)
    while ip + 72 <= n:
        L = min(W, n - 8 - ip)
        b = buf[ip:ip + W + 80]
        V = [set() for _ in range(64)]
        x = [0] * 64
        a_max = 0
        for k in range(64):
            s = R * k
            if s >= L:
                x[k] = s
                continue
            p = max(0, s - runin) if k else 0
            trips = 0
            while p < s + R and p < L:
                if p >= s:
                    V[k].add(p)
                p += adv(b, p)
                trips += 1
            x[k] = p
            a_max = max(a_max, trips)
        # overrun
        m = [0] * 64
        nx = [64] * 64
        Oset = [[] for _ in range(64)]
        o_max = 0
        for k in range(64):
            p = x[k]
            if p >= L:
                m[k], nx[k] = p, 64
                continue
            obase = (p // R) * R
            trips = 0
            while True:
                if p in V[p // R]:
                    m[k], nx[k] = p, p // R
                    break
                if p - obase >= cap:
                    m[k], nx[k] = p, 65
                    break
                Oset[k].append(p)
                p += adv(b, p)
                trips += 1
                if p >= L:
                    m[k], nx[k] = p, 64
                    break
            o_max = max(o_max, trips)
        # resolve
        k, e = 0, 0
        toks = 0
        act = 0
        while True:
            act += 1
            toks += len([q for q in V[k] if q >= e]) + len(Oset[k])
            mk, nk = m[k], nx[k]
            if nk == 65:                     # slow uniform walk until it merges
                st["unsynced"] += 1
                p = mk
                while p < L and p not in V[p // R]:
                    toks += 1
                    st["slow_tags"] += 1
                    p += adv(b, p)
                if p >= L:
                    consumed = p
                    break
                e, k = p, p // R
                continue
            if nk >= 64:
                consumed = mk
                break
            e, k = mk, nk
        st["windows"] += 1
        st["tokens"] += toks
        st["a_trips"] += a_max
        st["o_trips"] += o_max
        st["active"] += act
        st["lanes"] += min(64, (L + R - 1) // R)
        st["bytes"] += consumed
        ip += consumed
    return st


def main():
    files = sys.argv[1:] or ["html", "alice29.txt", "urls.10K", "geo.protodata", "kppkn.gtb", "fireworks.jpeg"]
    for f in files:
        raw = open(os.path.join(TD, f), "rb").read()[:65536]
        comp = O.compress(raw)
        # self-check of the walk: the true chain visits exactly the tag starts
        for R, runin, cap in ((32, 0, 64), (32, 16, 64), (32, 32, 64), (16, 16, 64), (64, 0, 64)):
            s = simulate(comp, R, runin, cap)
            w = s["windows"]
            print(json.dumps({"file": f, "R": R, "runin": runin, "cap": cap, "comp": len(comp), "windows": w,
                              "tokens/window": round(s["tokens"] / w, 1), "A trips": round(s["a_trips"] / w, 1),
                              "overrun trips": round(s["o_trips"] / w, 1), "unsynced/window": round(s["unsynced"] / w, 2),
                              "slow tags/window": round(s["slow_tags"] / w, 2),
                              "active lanes %": round(100 * s["active"] / max(1, s["lanes"]), 1)}))


if __name__ == "__main__":
    main()
