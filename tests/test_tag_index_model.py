"""The tag index of one large block (snappier_amd/csrc/tag_index.hip) as a CPU model (tests/tag_index_model.py): candidate entries per chunk,
rows, a scan that carries rows or positions, rows added for landings that were no candidates -- against the serial tag walk of the
reference (SnappyDecompressor.cs:184-347).  Small chunks (256 / 64 bytes) so that literals longer than a chunk, streams that end inside a
sub-chunk and many incompressible regions all occur in a few KiB.  The -m gpu tests (tests/test_gpu_big_blocks.py) run the kernels."""
import numpy as np
import pytest

import oracle as O
from tag_index_model import TagIndexModel, reference_entries


def _stream(data: bytes):
    z = O.compress(data, O.HASH_MUL)
    hb = 1
    while z[hb - 1] & 0x80:
        hb += 1
    return z, hb


def _low_entropy(rng, n):
    out = bytearray()
    while len(out) < n:
        p = int(rng.integers(1, 9))
        pat = bytes(rng.integers(0, 256, p, dtype=np.uint8))
        out += (pat * (int(rng.integers(8, 200)) // p + 1))
    return bytes(out[:n])


def _mixed(rng, n, regions):
    a = bytearray(_low_entropy(rng, n))
    for i in range(regions):
        at = (i + 1) * n // (regions + 1)
        ln = int(rng.integers(300, 1500))                                   # longer than a 256-byte chunk: the literal jumps chunks
        a[at: at + ln] = bytes(rng.integers(0, 256, ln, dtype=np.uint8))
    return bytes(a)


@pytest.mark.parametrize("kind,regions", [("low", 0), ("text", 0), ("mixed", 1), ("mixed", 5), ("mixed", 20), ("random", 0)])
def test_model_gives_the_serial_walks_entries(kind, regions):
    rng = np.random.default_rng(11 + regions)
    passes = 0
    for trial in range(6):
        n = int(rng.integers(3000, 40000))
        if kind == "low":
            data = _low_entropy(rng, n)
        elif kind == "text":
            words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(60)]
            data = b" ".join(words[int(i)] for i in rng.integers(0, 60, n // 5))[:n]
        elif kind == "mixed":
            data = _mixed(rng, n, regions)
        else:
            data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        z, hb = _stream(data)
        m = TagIndexModel(z, hb)
        r = m.run()
        assert r[0] == "done", (kind, regions, trial, r[0])
        want, final = reference_entries(z, hb, m.chunk, m.sub)
        assert r[2] == final == (len(z), len(data))
        assert r[1] == want, (kind, regions, trial)
        if kind in ("low", "text"):
            assert m.passes == 0                                             # every entry was a candidate: one scan
        if kind == "mixed":
            assert m.passes <= 4 * regions + 2                                # about a pass per incompressible region
            passes += m.passes
    if kind == "mixed":
        assert passes >= 1                                                    # (the path this test is for did run)


def test_a_truncated_stream_fails_and_never_claims_done():
    rng = np.random.default_rng(3)
    data = _mixed(rng, 20000, 3)
    z, hb = _stream(data)
    for cut in (len(z) - 1, len(z) // 2, hb + 5):
        m = TagIndexModel(z[:cut], hb)
        r = m.run()
        want = reference_entries(z[:cut], hb, m.chunk, m.sub)
        if want is None:
            assert r[0] == "fail"                                           # (the kernels then take the look-back pass, which marks the table irregular)
        else:
            assert r[0] == "done" and r[1] == want[0]


def test_more_landings_than_rows_give_up_cleanly():
    """A chunk can hold 8 rows; a stream made of literals that each leave their chunk lands everywhere: whatever happens, the model either
    equals the serial walk or fails -- it never returns other entries."""
    rng = np.random.default_rng(5)
    data = bytes(rng.integers(0, 256, 60000, dtype=np.uint8))
    z, hb = _stream(data)
    m = TagIndexModel(z, hb, chunk=128, sub=32, probe=8)
    r = m.run(max_passes=400)
    want, final = reference_entries(z, hb, 128, 32)
    assert r[0] == "fail" or (r[1] == want and r[2] == final)
