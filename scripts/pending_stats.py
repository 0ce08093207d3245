"""CPU statistics of what the tag-per-lane batches of k_decompress_chains leave for the in-order finish: the tags of a compressed
html-like block are cut into batches of <= 64 tags / <= 2 KiB of output as the kernel does (super-window cuts ignored), and every tag is classified:
first pass (literal, or a copy whose source lies below the batch), second pass (source inside the batch, not written by a pending tag, no pattern,
not straddling the batch start), the rest by reason -- and how many further lane-parallel passes would drain them (dependency depth)."""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
import numpy as np
from oracle import pyoracle as O
from ring_model import parse_tags

def blocks(kind, nb):
    td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
    if kind == "html":
        import datagen
        html = open(os.path.join(td, "html"), "rb").read()
        raw = datagen.html_like_blocks(html, 0, nb).tobytes()
        return [raw[i * 65536:(i + 1) * 65536] for i in range(nb)]
    data = open(os.path.join(td, kind), "rb").read()
    return [data[i * 65536:(i + 1) * 65536] for i in range(min(nb, len(data) // 65536))]

def stats(z):
    _, tags = parse_tags(z)
    c = collections.Counter()
    depth_hist = collections.Counter()
    i, op = 0, 0
    while i < len(tags):
        if tags[i][0] and tags[i][1] > 64:
            op += tags[i][1]; i += 1; c["long_literals"] += 1
            continue
        j, span = i, 0
        while j < len(tags) and j - i < 64 and not (tags[j][0] and tags[j][1] > 64) and span + tags[j][1] <= 2048:
            span += tags[j][1]; j += 1
        mark = op
        done = np.zeros(span, dtype=bool)          # bytes of the batch that exist
        pend = []
        o = op
        for (lit, ln, off, _b) in tags[i:j]:
            if lit or (off >= ln and o - off + ln <= mark):
                done[o - mark:o - mark + ln] = True
                c["pass1"] += 1
            else:
                pend.append((o, ln, off))
            o += ln
        c["batches"] += 1
        c["tags"] += j - i
        c["pend1"] += len(pend)
        # second pass
        rest = []
        newly = []
        for (o, ln, off) in pend:
            s = o - off
            if off < ln: rest.append((o, ln, off, "pattern"))
            elif s < mark: rest.append((o, ln, off, "straddle"))
            elif done[s - mark:s - mark + ln].all(): newly.append((o, ln)); c["pass2"] += 1
            else: rest.append((o, ln, off, "blocked"))
        for (o, ln) in newly: done[o - mark:o - mark + ln] = True
        c["rest"] += len(rest)
        c["rest_bytes"] += sum(r[1] for r in rest)
        c["rest_batches"] += 1 if rest else 0
        c["rest_chunks64"] += (sum(r[1] for r in rest) + 63) // 64
        for r in rest: c["rest_" + r[3]] += 1
        # how many more lane-parallel passes (pattern and straddle allowed from now on, byte-exact availability)
        level = 0
        cur = rest
        while cur:
            level += 1
            nxt, newly = [], []
            for (o, ln, off, why) in cur:
                s = o - off
                need = min(ln, off)                 # a pattern copy needs its first `off` bytes
                lo = max(s, mark)
                if done[lo - mark:s + need - mark].all(): newly.append((o, ln))
                else: nxt.append((o, ln, off, why))
            for (o, ln) in newly: done[o - mark:o - mark + ln] = True
            depth_hist[level] += len(newly)
            cur = nxt
        c["max_level_sum"] += level
        op += span
        i = j
    return c, depth_hist

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "html"
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tot, dh = collections.Counter(), collections.Counter()
    bl = blocks(kind, nb)
    for b in bl:
        z = O.compress(b)
        c, d = stats(z)
        tot += c; dh += d
    n = len(bl)
    out = {"data": kind, "blocks": n, **{k: round(v / n, 1) for k, v in sorted(tot.items())}, "levels_per_batch": round(tot["max_level_sum"] / tot["batches"], 2),
           "drained_at_level": {str(k): round(v / n, 1) for k, v in sorted(dh.items())}}
    print(json.dumps(out))
