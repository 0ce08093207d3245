"""The CPU model of the output-granular batch executor (tests/ring_model.py = k_decompress_ring, FRONT = 4 of decompress.hip) must decode
exactly what a sequential decoder does -- on text, on binary data with long literals, on low-entropy blocks whose pattern copies make
the pointer doubling run its full depth -- for the shipped ring geometry and for smaller rings / other misalignments of the output."""
import numpy as np
import pytest

import oracle as O
from conftest import read_testdata
import datagen
import ring_model as RM


def _blocks():
    html = read_testdata("html")
    yield "html-like", datagen.html_like_blocks(html, 3, 1).tobytes()
    yield "low-entropy", datagen.low_entropy_block(1).tobytes()
    yield "alice", read_testdata("alice29.txt")[:65536]
    yield "jpeg (long literals)", read_testdata("fireworks.jpeg")[:65536]
    yield "geo", read_testdata("geo.protodata")[:40000]
    yield "zeros", bytes(20000)
    yield "short", b"abcabcabcabcabcabcabc" * 3 + b"xyz"


@pytest.mark.parametrize("ring,g0,span", [(2048, 0, 1024), (2048, 15, 1024), (4096, 5, 1984), (1024, 9, 512)])
def test_ring_model_equals_sequential_decode(ring, g0, span):
    for name, blk in _blocks():
        z = O.compress(blk)
        got, st = RM.decode_with_ring_model(z, ring, g0, span)
        assert got == blk, f"{name}: ring {ring} g0 {g0}"
        assert st["batches"] >= 1


def test_ring_model_statistics_match_the_design_numbers():
    """What DESIGN.md quotes for the html-like workload: ~1 000 sub-steps per block, about a fifth of them with a source inside the
    sub-step, ~1.5 doubling rounds each, and about 40 % of the copies older than the 2 KiB ring."""
    html = read_testdata("html")
    blk = datagen.html_like_blocks(html, 0, 1).tobytes()
    got, st = RM.decode_with_ring_model(O.compress(blk))
    assert got == blk
    assert 1000 <= st["substeps"] <= 1200
    assert 0.1 < st["dep_substeps"] / st["substeps"] < 0.35
    assert 1.0 <= st["rounds"] / st["dep_substeps"] <= 3.0
    assert 1200 <= st["far_tags"] <= 2400


def test_ring_model_pattern_copies_need_log_rounds():
    blk = bytes([7]) * 5000                                                     # offset-1 copies of 64: every lane's source is its left neighbour
    got, st = RM.decode_with_ring_model(O.compress(blk))
    assert got == blk
    assert st["rounds"] / max(st["dep_substeps"], 1) >= 6.0                      # 64 lanes: six doublings + the round that proves it
