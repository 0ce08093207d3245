#!/usr/bin/env python3
"""Copy-offset and copy-length histograms of the bench workloads (configs 2, 3, 5), from the oracle's compressed blocks
(CPU only).  VERDICT r1 asked for them before sizing an LDS ring of recent output for the decompressor: the histogram says
how much of the back-reference traffic a ring of R bytes would serve.  One JSON object on stdout."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle as O
import datagen
from conftest import CORPUS, read_testdata

EDGES = [64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536]


def tags(z: bytes):
    """(offset, length) of every copy tag and (0, length) of every literal of one compressed block."""
    i, n = 0, len(z)
    while z[i] & 0x80:
        i += 1
    i += 1
    out = []
    while i < n:
        c = z[i]
        t = c & 3
        if t == 0:
            l = (c >> 2) + 1
            i += 1
            if l > 60:
                k = l - 60
                l = int.from_bytes(z[i:i + k], "little") + 1
                i += k
            out.append((0, l))
            i += l
        elif t == 1:
            out.append((((c >> 5) << 8) | z[i + 1], ((c >> 2) & 7) + 4))
            i += 2
        elif t == 2:
            out.append((z[i + 1] | (z[i + 2] << 8), (c >> 2) + 1))
            i += 3
        else:
            out.append((int.from_bytes(z[i + 1:i + 5], "little"), (c >> 2) + 1))
            i += 5
    return out


def summarize(name, blocks):
    off_tags = np.zeros(len(EDGES), dtype=np.int64)
    off_bytes = np.zeros(len(EDGES), dtype=np.int64)
    ntags = ncopy = nlit = lit_bytes = copy_bytes = comp = 0
    for blk in blocks:
        z = O.compress(blk, O.HASH_CRC32C)
        comp += len(z)
        for off, l in tags(z):
            ntags += 1
            if off == 0:
                nlit += 1
                lit_bytes += l
            else:
                ncopy += 1
                copy_bytes += l
                k = int(np.searchsorted(EDGES, off, side="left"))
                off_tags[k] += 1
                off_bytes[k] += l
    nb = len(blocks)
    cum_t = np.cumsum(off_tags) / max(1, ncopy)
    cum_b = np.cumsum(off_bytes) / max(1, copy_bytes)
    return {"workload": name, "blocks": nb, "ratio": round(comp / (nb * 65536), 4), "tags_per_block": round(ntags / nb, 1),
            "copies_per_block": round(ncopy / nb, 1), "literals_per_block": round(nlit / nb, 1),
            "bytes_per_tag": round(65536 * nb / ntags, 2), "copy_byte_share": round(copy_bytes / (copy_bytes + lit_bytes), 4),
            "copies_with_offset_at_most": {str(e): round(float(c), 4) for e, c in zip(EDGES, cum_t)},
            "copy_bytes_with_offset_at_most": {str(e): round(float(c), 4) for e, c in zip(EDGES, cum_b)}}


nb = int(sys.argv[1]) if len(sys.argv) > 1 else 96
html = read_testdata("html")
res = [summarize("configs[1] html-like", [bytes(datagen.html_like_blocks(html, b, 1)) for b in range(0, nb * 1700, 1700)][:nb]),
       summarize("configs[2] low-entropy", [datagen.low_entropy_block(b).tobytes() for b in range(nb)]),
       summarize("configs[4] mixed corpus", [bytes(datagen.corpus_blocks([read_testdata(n) for n in CORPUS], b, 1, datagen.MIXED_SEED)) for b in range(nb)])]
print(json.dumps({"note": "copy offsets / lengths of the oracle's compressed blocks (hash crc32c); a ring of the last R output bytes in LDS "
                          "serves the copies with offset + length <= R", "results": res}, indent=1))
