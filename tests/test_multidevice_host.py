"""Host logic of snappier_amd/multidevice.py (no GPU): the range split and the header walk that cuts a framed stream at chunk boundaries."""
import numpy as np

import oracle as O
from snappier_amd.multidevice import chunk_table, split_ranges


def test_ranges_cover_every_chunk_once_in_order():
    for n in (1, 15, 16, 17, 100, 163840):
        for k in (1, 2, 3, 8):
            r = split_ranges(n, k, 16)
            assert r[0][0] == 0 and sum(c for _f, c in r) == n and len(r) <= k
            assert all(r[i][0] + r[i][1] == r[i + 1][0] for i in range(len(r) - 1))
            assert all(c >= min(16, n) for _f, c in r)


def test_chunk_table_cuts_an_oracle_framed_stream_at_its_chunk_headers():
    rng = np.random.default_rng(5)
    data = bytes(rng.integers(0, 4, 5 * 65536 + 123, dtype=np.uint8))        # compressible chunks and a partial last one
    framed = np.frombuffer(O.frame_encode(data), dtype=np.uint8)
    table, end = chunk_table(framed)
    assert end == framed.size and len(table) == 1 + 6                         # stream identifier + six data chunks
    assert table[0] == (0, 10)
    for (off, size) in table[1:]:
        assert int(framed[off]) in (0, 1) and off + size <= framed.size
    # every sub-range of whole chunks is itself a decodable stream, and the pieces concatenate to the input
    parts = [O.frame_decode(framed[table[a][0]:table[b - 1][0] + table[b - 1][1]].tobytes()) for a, b in ((0, 3), (3, 5), (5, 7))]
    assert b"".join(parts) == data
    # a truncated stream: the table stops before the broken chunk
    t2, e2 = chunk_table(framed[:-5])
    assert len(t2) == len(table) - 1 and e2 == table[-1][0]
