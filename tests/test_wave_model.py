"""The 64-lane model of the HIP compress kernel's algorithm (tests/wave_model.py) must reproduce the serial greedy
parse of the oracle byte-for-byte -- including when LDS write winners are arbitrary.  CPU only."""
import numpy as np
import pytest

import oracle as O
from conftest import read_testdata
import datagen
import kats
import wave_model as WM


def check(data: bytes, variant: int, rng=None, stats=None):
    got = WM.compress_wave(data, variant, stats, rng)
    ref = O.compress(data, variant)
    assert got == ref, (len(got), len(ref), next(i for i, (a, b) in enumerate(zip(got, ref)) if a != b) if got[:len(ref)] != ref[:len(got)] else "prefix")


def test_probe_sequence_matches_skip_heuristic():
    # D[k] from the closed recurrence == positions produced by the reference's skip bookkeeping
    skip, ip, seq = 32, 0, []
    for _ in range(300):
        seq.append(ip)
        bb = skip >> 5
        skip += bb
        ip += bb
    assert WM.D[:300].tolist() == seq


def test_crc_rows_reproduce_hash():
    rng = np.random.default_rng(3)
    d = rng.integers(0, 2**32, 64, dtype=np.uint64)
    for mask in (2 * 255, 2 * 1023, 2 * 16383):
        for variant in (0, 1):
            got = WM.hash_lanes(d, mask, variant)
            ref = [O.hash_bytes(int(x), mask, variant) >> 1 for x in d]
            assert got.tolist() == ref


@pytest.mark.parametrize("variant", [O.HASH_CRC32C, O.HASH_MUL])
@pytest.mark.parametrize("name", ["html", "alice29.txt", "kppkn.gtb", "fireworks.jpeg", "paper-100k.pdf", "geo.protodata"])
def test_corpus_first_block(name, variant):
    stats = {}
    check(read_testdata(name)[:65536], variant, np.random.default_rng(7), stats)
    assert stats["rounds"] > 0


@pytest.mark.parametrize("s", kats.STRING_CASES)
def test_string_cases(s):
    for variant in (0, 1):
        check(s, variant, np.random.default_rng(1))


def test_small_and_edge_lengths():
    html = read_testdata("html")
    for n in [0, 1, 14, 15, 16, 17, 30, 31, 32, 33, 255, 256, 257, 1000, 4096, 16383, 16384, 16385]:
        for variant in (0, 1):
            check(html[:n], variant, np.random.default_rng(n))
    check(bytes(65536), 0)
    check(bytes(range(256)) * 256, 1)


def test_low_entropy_and_random():
    for b in range(3):
        check(datagen.low_entropy_block(b).tobytes(), 0, np.random.default_rng(b))
    rng = np.random.default_rng(301)
    for i in list(range(2)) + list(range(100, 160)):
        check(datagen.random_data_case(i, rng), i & 1, np.random.default_rng(i))
    # tiny alphabets maximise same-bucket conflicts inside a round
    r = np.random.default_rng(5)
    for _ in range(20):
        check(r.integers(0, 3, int(r.integers(20, 3000)), dtype=np.uint8).tobytes(), int(r.integers(0, 2)), r)
