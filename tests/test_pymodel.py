"""oracle/pymodel.py (pure Python, transcribed from the language-neutral spec of SURVEY.md Appendix A; H_crc from the
bit-serial CRC definition) and oracle/snappy_oracle.c (C, x86 crc32 instruction) are two independent statements of the
same semantics: they must agree byte for byte, for both hashes.  This is what pins the crc32c-hash compressor bytes from
a second lineage (no reference fixture covers them, DESIGN.md "Oracle").  CPU only."""
import numpy as np
import pytest

import oracle as O
from oracle import pymodel as PM
from conftest import CORPUS, read_testdata
import datagen
import kats


def test_hash_definitions_agree():
    rng = np.random.default_rng(11)
    for mask in (2 * 255, 2 * 1023, 2 * 8191, 2 * 16383):
        for b in [0, 1, 0xFFFFFFFF, 0x80000000] + [int(x) for x in rng.integers(0, 2**32, 300, dtype=np.uint64)]:
            assert PM.h_crc(b, mask) == O.hash_bytes(b, mask, O.HASH_CRC32C) >> 1
            assert PM.h_mul(b, mask) == O.hash_bytes(b, mask, O.HASH_MUL) >> 1
            assert PM.crc32c_step32(b ^ mask) == PM._step32_fast(b ^ mask)      # linearity shortcut == bit-serial definition


@pytest.mark.parametrize("variant", [O.HASH_CRC32C, O.HASH_MUL])
@pytest.mark.parametrize("name", CORPUS)
def test_corpus_first_fragment_and_a_ragged_tail(name, variant):
    data = read_testdata(name)
    assert PM.compress(data[:65536], variant) == O.compress(data[:65536], variant)
    tail = data[-(len(data) % 65536 or 4097):]
    assert PM.compress(tail, variant) == O.compress(tail, variant)


def test_survey_known_answers():
    """compress(html[0:65536]) = 16 446 B (crc) / 16 533 B (mul, = golden chunk 0 of html_x_4.snappy)."""
    import hashlib
    html = read_testdata("html")
    z = PM.compress(html[:65536], PM.HASH_CRC32C)
    assert len(z) == 16446 and hashlib.sha256(z).hexdigest().startswith("822945612f80e8d4")
    z = PM.compress(html[:65536], PM.HASH_MUL)
    assert len(z) == 16533 and hashlib.sha256(z).hexdigest().startswith("2f8a1e2979f6b2cb")
    z = PM.compress(html[:102400], PM.HASH_CRC32C)
    assert len(z) == 22774 == len(O.compress(html[:102400], O.HASH_CRC32C))


def test_edge_lengths_strings_and_generated_blocks():
    html = read_testdata("html")
    for n in list(range(0, 40)) + [255, 256, 257, 1000, 4095, 4096, 16383, 16384, 16385]:
        for variant in (0, 1):
            assert PM.compress(html[3:3 + n], variant) == O.compress(html[3:3 + n], variant), (n, variant)
    for s in kats.STRING_CASES:
        for variant in (0, 1):
            assert PM.compress(s, variant) == O.compress(s, variant)
    for variant in (0, 1):
        blk = datagen.html_like_blocks(html, 77, 1).tobytes()
        assert PM.compress(blk, variant) == O.compress(blk, variant)
        low = datagen.low_entropy_block(5).tobytes()
        assert PM.compress(low, variant) == O.compress(low, variant)


def test_fuzz_small_inputs_and_decoder():
    rng = np.random.default_rng(2024)
    for i in range(150):
        n = int(rng.integers(0, 6000))
        kind = i % 4
        if kind == 0:
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif kind == 1:
            data = rng.integers(0, int(rng.integers(1, 5)), n, dtype=np.uint8).tobytes()
        elif kind == 2:
            unit = rng.integers(0, 256, int(rng.integers(1, 70)), dtype=np.uint8).tobytes()
            data = (unit * (n // len(unit) + 1))[:n]
        else:
            data = datagen.random_data_case(i, rng)[:6000]
        v = i & 1
        z = PM.compress(data, v)
        assert z == O.compress(data, v), (i, n, v)
        assert PM.decompress(z) == data == O.decompress(z)


def test_decoder_on_reference_fixtures_and_bad_data():
    for name in ("baddata1.snappy", "baddata2.snappy", "baddata3.snappy"):
        with pytest.raises(PM.Invalid):
            PM.decompress(read_testdata(name))
        assert O.decompress_status(read_testdata(name)) == O.ERR_BAD_OFFSET
    # framing: the Python statement reproduces the reference's golden stream (mul hash) byte for byte
    alice = O.frame_decode(read_testdata("alice29.snappy"))
    assert PM.frame(alice, PM.HASH_MUL) == read_testdata("alice29.snappy")
