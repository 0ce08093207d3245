"""The CPU model of the window compressor's algorithm (tests/window_model.c: multi-token speculative rounds, cut at
the first repeated bucket, sparse rounds for long scans and the fragment tail) must reproduce the serial greedy parse
of the oracle byte for byte.  CPU only; the HIP kernel itself is compared with the oracle in the -m gpu tests."""
import numpy as np
import pytest

import oracle as O
from conftest import read_testdata
import datagen
import kats
import window_model as WM

CORPUS = ["html", "alice29.txt", "asyoulik.txt", "kppkn.gtb", "fireworks.jpeg", "paper-100k.pdf", "geo.protodata",
          "lcet10.txt", "plrabn12.txt", "urls.10K"]


def check(data: bytes, variant: int, np_: int = 2, cap: int = 32, stats=None):
    got = WM.compress(data, variant, np_, cap, stats)
    ref = O.compress(data, variant)
    assert got == ref, (len(data), variant, np_, cap, len(got), len(ref))


@pytest.mark.parametrize("variant", [O.HASH_CRC32C, O.HASH_MUL])
@pytest.mark.parametrize("name", CORPUS)
def test_corpus_whole_files(name, variant):
    data = read_testdata(name)[:4 * 65536 + 1234]
    for np_ in (1, 2, 4):
        st = WM.Stats()
        check(data, variant, np_, 32, st)
        assert st.dense_rounds + st.sparse_rounds > 0
    check(data[:65536], variant, 2, 16)


@pytest.mark.parametrize("s", kats.STRING_CASES)
def test_string_cases(s):
    for variant in (0, 1):
        check(s, variant)


def test_small_and_edge_lengths():
    html = read_testdata("html")
    lens = list(range(0, 70)) + [127, 128, 129, 255, 256, 257, 1000, 4096, 16383, 16384, 16385, 65535, 65536, 65537]
    for n in lens:
        for variant in (0, 1):
            for np_ in (1, 2):
                check(html[7:7 + n], variant, np_)
    check(bytes(65536), 0)
    check(bytes(range(256)) * 256, 1)


def test_generated_configs_and_fuzz():
    html = read_testdata("html")
    blocks = datagen.html_like_blocks(html, 5, 6).tobytes()
    for variant in (0, 1):
        check(blocks, variant)
    for b in range(4):
        check(datagen.low_entropy_block(b).tobytes(), b & 1, 1 + (b & 1))
    rng = np.random.default_rng(301)
    for i in list(range(4)) + list(range(100, 180)):
        check(datagen.random_data_case(i, rng), i & 1, 1 + (i % 3 == 0))
    # tiny alphabets maximise repeated buckets inside a window; short periods make in-window candidates
    r = np.random.default_rng(5)
    for _ in range(60):
        n = int(r.integers(20, 5000))
        check(r.integers(0, int(r.integers(2, 5)), n, dtype=np.uint8).tobytes(), int(r.integers(0, 2)), int(r.integers(1, 3)))
    for period in (1, 2, 3, 5, 8, 31, 33, 63, 64, 65, 100, 127, 129, 300):
        unit = r.integers(0, 256, period, dtype=np.uint8).tobytes()
        data = (unit * (70000 // period + 1))[:70000]
        check(data, period & 1)
        noisy = bytearray(data[:20000])
        for k in r.integers(0, 20000, 40):
            noisy[int(k)] ^= 0x55
        check(bytes(noisy), 0, 2)
