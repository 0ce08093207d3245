"""The sub-chain tag parse of the decompressor (HISTORY.md §4.1c), as an executable CPU model (tests/subchain_model.py), against the
sequential tag walk it must reproduce: on the corpus, on streams built against it, on corrupted streams and on plain garbage."""
import importlib.util
import os

import numpy as np
import pytest

import oracle as O
import subchain_model as M
from conftest import CORPUS, ROOT, read_testdata


def check_window(buf: bytes, avail: int):
    L = min(M.W, avail) - 8
    want_pos, want_end = M.sequential(buf, L)
    pos, consumed = M.window(buf, avail)
    assert pos == want_pos
    assert consumed == want_end
    assert all(b - a >= 2 for a, b in zip(pos, pos[1:]))          # the u16 tag list of the kernel holds at most W / 2 entries
    return len(pos)


@pytest.mark.parametrize("name", CORPUS)
def test_windows_of_corpus_blocks_equal_the_sequential_walk(name):
    data = read_testdata(name)
    for start in (0, 65536):
        raw = data[start:start + 65536]
        if not raw:
            continue
        comp = O.compress(raw)
        stats = {}
        for ip, pos, consumed in M.stream_windows(comp, stats):
            want_pos, want_end = M.sequential(comp[ip:] + bytes(M.W + 16), min(M.W, len(comp) - ip) - 8)
            assert pos == want_pos and consumed == want_end
        if name == "html" and start == 0:                           # the numbers DESIGN.md §4.1, HISTORY.md §4.1c quotes for the sizing
            assert 550 < stats["tokens"] / stats["windows"] < 700
            assert stats["active"] / stats["lanes"] > 0.6


def test_streams_built_against_the_parse():
    spec = importlib.util.spec_from_file_location("adversarial_streams", os.path.join(ROOT, "scripts", "adversarial_streams.py"))
    A = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(A)
    for kind in A.KINDS:
        comp = A.build(kind)
        stats = {}
        wins = M.stream_windows(comp, stats)
        assert wins
        for ip, pos, consumed in wins:
            want_pos, want_end = M.sequential(comp[ip:] + bytes(M.W + 16), min(M.W, len(comp) - ip) - 8)
            assert pos == want_pos and consumed == want_end
    # chains inside 60-byte literals of 0xFF step by 5 and meet a tag start (every 61 bytes) only after ~300 bytes: the true
    # chain does not merge within the cap and the fallback must have run
    stats = {}
    M.stream_windows(A.build("literals_of_ff_period61"), stats)
    assert stats["unmerged"] > 0 and stats["slow_tags"] > 0
    # the purest case: every byte a 7-byte tag (0x14: literal of 6), so each of the seven phases is a self-consistent chain
    # and only every seventh lane starts on the true one: 224 bytes apart, beyond the cap
    buf = bytes([0x14]) * 5000
    stats = {}
    pos, consumed = M.window(buf, len(buf), stats)
    assert pos == list(range(0, M.W - 8, 7)) and consumed == pos[-1] + 7 and stats["unmerged"] > 0


def test_garbage_and_corrupted_streams():
    rng = np.random.default_rng(20260928)
    total = 0
    for trial in range(300):
        avail = int(rng.integers(72, 5000))
        kind = trial % 4
        if kind == 0:
            buf = rng.integers(0, 256, avail, dtype=np.uint8).tobytes()
        elif kind == 1:                                             # few distinct bytes: long runs of the same tag shape
            buf = rng.choice(np.array([0x00, 0x01, 0x02, 0x03, 0xF0, 0xF4, 0xFC, 0x7F], dtype=np.uint8), avail).tobytes()
        else:                                                       # a real stream with a few bytes flipped
            raw = read_testdata("html")[int(rng.integers(0, 30000)):][:8192]
            z = bytearray(O.compress(raw)[3:])
            for _ in range(int(rng.integers(0, 6))):
                z[int(rng.integers(0, len(z)))] = int(rng.integers(0, 256))
            buf = bytes(z[:avail])
            avail = len(buf)
            if avail < 72:
                continue
        total += check_window(buf + bytes(M.W + 16), avail)
    assert total > 10000


def test_long_literals_leave_the_window():
    # a literal longer than the window: the chain leaves it at once, `consumed` is the byte after the literal's body
    body = bytes(range(256)) * 20
    comp = bytes([61 << 2, (len(body) - 1) & 255, (len(body) - 1) >> 8]) + body + bytes([0x05, 0x10]) * 50
    pos, consumed = M.window(comp + bytes(M.W + 16), len(comp))
    assert pos == [0] and consumed == 3 + len(body)
