"""Pins the CPU oracle (oracle/snappy_oracle.c) against every golden vector / KAT the reference's tests hold for the
hot path.  CPU only.  The HIP path is then compared against this oracle in test_gpu_*.py."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import oracle as O
from conftest import CORPUS, GOLDEN, read_testdata
import kats
import datagen

STREAM_HEADER = bytes([0xFF, 0x06, 0x00, 0x00, 0x73, 0x4E, 0x61, 0x50, 0x70, 0x59])


def parse_frames(stream: bytes):
    """-> list of (type, masked_crc, body) for data chunks of a framed stream."""
    assert stream[:10] == STREAM_HEADER
    ip, chunks = 10, []
    while ip < len(stream):
        t = stream[ip]
        size = int.from_bytes(stream[ip + 1:ip + 4], "little")
        body = stream[ip + 4:ip + 4 + size]
        ip += 4 + size
        if t in (0, 1):
            chunks.append((t, struct.unpack("<I", body[:4])[0], body[4:]))
    assert ip == len(stream)
    return chunks


# ---------------------------------------------------------------- primitives

@pytest.mark.parametrize("data,expected", kats.CRC32C)
def test_crc32c_kats(data, expected):
    assert O.crc32c(data) == expected
    assert O.crc32c_bitwise(data) == expected


def test_crc32c_fast_equals_bitwise_random():
    rng = np.random.default_rng(1)
    for n in [0, 1, 3, 4, 7, 8, 9, 63, 64, 65, 1000, 65536]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert O.crc32c(d) == O.crc32c_bitwise(d)


def test_crc32c_mask_formula():
    for x in [0, 1, 0xE3069283, 0xFFFFFFFF, 0x12345678]:
        assert O.crc32c_mask(x) == ((((x >> 15) | (x << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


@pytest.mark.parametrize("value,enc", kats.VARINT)
def test_varint_kats(value, enc):
    assert O.varint_write(value) == enc
    assert O.varint_write(value, cap=len(enc) - 1) == b""            # Test_TryWriteInsufficientBufferLength
    for pad in (b"", b"\x00" * (16 - len(enc)), b"\xff" * (16 - len(enc))):   # ZeroPadding / OnePadding
        st, v, br = O.varint_read(enc + pad)
        assert (st, v, br) == (O.OK, value, len(enc))


@pytest.mark.parametrize("enc", kats.VARINT_INCOMPLETE)
def test_varint_incomplete(enc):
    st, _v, br = O.varint_read(enc)
    assert st == O.ERR_INCOMPLETE and br == 0


def test_varint_bad():
    st, _v, br = O.varint_read(kats.VARINT_BAD)
    assert st == O.ERR_BAD_LENGTH and br == 0
    with pytest.raises(O.OracleError):
        O.get_uncompressed_length(kats.VARINT_BAD)


@pytest.mark.parametrize("expected,s1,s2,length", kats.FIND_MATCH_LENGTH)
def test_find_match_length(expected, s1, s2, length):
    assert O.find_match_length(s1.encode(), s2.encode(), length) == expected


def test_helpers():
    L = O.lib()
    for v, s in kats.LEFT_SHIFT_OVERFLOWS_TRUE:
        assert L.orc_left_shift_overflows(v, s)
    for v, s in kats.LEFT_SHIFT_OVERFLOWS_FALSE:
        assert not L.orc_left_shift_overflows(v, s)
    for v in range(1, 32):
        assert L.orc_log2_floor(v) == int(np.floor(np.log2(v)))
    assert L.orc_log2_floor(0) == 0
    assert O.max_compressed_length(65536) == 76496 and L.orc_max_fragment_compressed_length(65536) == 76491
    assert [O.table_size(n) for n in (0, 255, 256, 257, 512, 513, 16384, 16385, 65536)] == \
           [256, 256, 256, 512, 512, 1024, 16384, 16384, 16384]


def test_hash_variants():
    # crc variant = 32 reflected shift/xor steps of (bytes ^ mask), no inversion (HashTable.cs:109-112)
    rng = np.random.default_rng(2)
    for b in rng.integers(0, 2**32, 200, dtype=np.uint64):
        for mask in (2 * 255, 2 * 16383):
            b = int(b)
            assert O.hash_bytes(b, mask, O.HASH_CRC32C) == O.lib().orc_crc32c_u32_step_bitwise(b, mask) & mask
            assert O.hash_bytes(b, mask, O.HASH_MUL) == (((0x1E35A7BD * b) & 0xFFFFFFFF) >> 17) & mask


# ---------------------------------------------------------------- golden framed files (mul hash, bit-exact)

@pytest.mark.parametrize("name,nchunks", [("html_x_4.snappy", 7), ("alice29.snappy", 3)])
def test_golden_framed_files(name, nchunks):
    stream = read_testdata(name)
    chunks = parse_frames(stream)
    assert len(chunks) == nchunks
    raw = b""
    for t, crc, body in chunks:
        assert t == 0
        dec = O.decompress(body)
        assert O.crc32c(dec, masked=True) == crc                     # masked CRC matches
        assert O.compress(dec, O.HASH_MUL) == body                   # mul-hash compressor reproduces every chunk body
        raw += dec
    if name == "html_x_4.snappy":
        assert raw == read_testdata("html_x_4")
    else:
        assert len(raw) == 152089                                    # CRLF variant of alice29
    assert O.frame_decode(stream) == raw
    assert O.frame_encode(raw, O.HASH_MUL) == stream                 # whole framed stream is reproduced byte-for-byte
    assert O.frame_decoded_length(stream) == len(raw)


def test_html64k_kats():
    html = read_testdata("html")[:65536]
    for variant, (n, sha) in ((O.HASH_MUL, kats.HTML64K_MUL), (O.HASH_CRC32C, kats.HTML64K_CRC)):
        c = O.compress(html, variant)
        assert len(c) == n and hashlib.sha256(c).hexdigest() == sha
        assert c[:3] == bytes([0x80, 0x80, 0x04])
        assert O.decompress(c) == html


def test_libsnappy_goldens():
    """mul-hash output == libsnappy 1.1.8 for every >=16 KiB window of the corpus (vectors made by make_golden.py)."""
    g = json.load(open(os.path.join(GOLDEN, "libsnappy_mul_goldens.json")))
    for name, rows in g["files"].items():
        data = read_testdata(name)
        for r in rows:
            c = O.compress(data[r["start"]:r["start"] + r["len"]], O.HASH_MUL)
            assert len(c) == r["clen"] and hashlib.sha256(c).hexdigest() == r["sha256"], (name, r["start"])
    for name, r in g["whole_files"].items():
        c = O.compress(read_testdata(name), O.HASH_MUL)
        assert len(c) == r["clen"] and hashlib.sha256(c).hexdigest() == r["sha256"], name


# ---------------------------------------------------------------- bad data

@pytest.mark.parametrize("name,declared", [("baddata1.snappy", 128082), ("baddata2.snappy", 128059), ("baddata3.snappy", 130378)])
def test_baddata_files(name, declared):
    d = read_testdata(name)
    assert O.get_uncompressed_length(d) == declared
    assert O.decompress_status(d) == O.ERR_BAD_OFFSET                # SnappyTests.cs:287-331 -> InvalidDataException


def test_bad_simple_corruption():                                     # SnappyTests.cs:247-267
    c = bytearray(O.compress(b"making sure we don't crash with corrupted input"))
    c[1] = (c[1] - 1) & 0xFF
    c[3] = (c[3] + 1) & 0xFF
    assert O.decompress_status(bytes(c)) in (O.ERR_BAD_OFFSET, O.ERR_TOO_LONG, O.ERR_INCOMPLETE)


def test_bad_long_length():                                           # SnappyTests.cs:269-285
    c = bytearray(O.compress(b"A" * 1000))
    c[0], c[1] = 255, 127
    assert O.decompress_status(bytes(c), cap=1000) == O.ERR_OUTPUT_TOO_SMALL
    assert O.decompress_status(bytes(c), cap=16383) == O.ERR_INCOMPLETE


def test_insufficient_output():                                       # SnappyTests.cs:218-245
    c = O.compress(b"A" * 100000)
    assert O.decompress_status(c, cap=100) == O.ERR_OUTPUT_TOO_SMALL
    with pytest.raises(O.OracleError) as e:
        O.compress(b"A" * 100000, cap=10)
    assert e.value.status == O.ERR_OUTPUT_TOO_SMALL


def test_decoder_error_taxonomy():
    assert O.decompress_status(b"") == O.ERR_INCOMPLETE                       # no preamble at all
    assert O.decompress_status(bytes([5, 0x00])) == O.ERR_INCOMPLETE          # literal tag, body missing
    assert O.decompress_status(bytes([4, 0x0C, 97, 98, 99, 100])) == O.OK     # 4-byte literal
    assert O.decompress_status(bytes([4, 0x10, 97, 98, 99, 100, 101])) == O.ERR_TOO_LONG
    assert O.decompress_status(bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x00])) == O.ERR_BAD_OFFSET   # offset 0
    assert O.decompress_status(bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x05])) == O.ERR_BAD_OFFSET   # offset > produced
    assert O.decompress(bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x04])) == b"abcdabcd"
    assert O.decompress(bytes([8, 0x00, 97, 0x1A, 0x01, 0x00])) == b"a" * 8   # copy-2, offset 1, len 7: pattern copy
    assert O.decompress(bytes([6, 0x00, 97, 0x13, 0x01, 0x00, 0x00, 0x00])) == b"a" * 6   # copy-4
    assert O.decompress_status(bytes([8, 0x0C, 97, 98, 99, 100, 0x02])) == O.ERR_INCOMPLETE          # truncated copy tag


# ---------------------------------------------------------------- round trips (property tests)

@pytest.mark.parametrize("variant", [O.HASH_CRC32C, O.HASH_MUL])
@pytest.mark.parametrize("name", CORPUS)
def test_corpus_roundtrip(name, variant):                             # SnappyTests.cs:8-39
    data = read_testdata(name)
    c = O.compress(data, variant)
    assert len(c) <= O.max_compressed_length(len(data))
    assert O.get_uncompressed_length(c) == len(data)
    assert O.decompress(c) == data


@pytest.mark.parametrize("s", kats.STRING_CASES)
def test_string_cases(s):                                             # SnappyTests.cs:178-202
    for variant in (O.HASH_CRC32C, O.HASH_MUL):
        assert O.decompress(O.compress(s, variant)) == s
    assert O.frame_decode(O.frame_encode(s)) == s


def test_zeros_block():
    c = O.compress(bytes(65536))
    assert len(c) == 3077                                             # 1 literal + 1024 copies
    assert c[3:6] == bytes([0x00, 0x00, 0xFE]) and c[6:8] == bytes([0x01, 0x00])


def test_random_data_property():                                      # SnappyTests.cs:401-446 (property, own PRNG)
    rng = np.random.default_rng(301)
    for i in list(range(12)) + list(range(100, 700)):
        d = datagen.random_data_case(i, rng)
        for variant in (O.HASH_CRC32C, O.HASH_MUL):
            assert O.decompress(O.compress(d, variant)) == d


def test_framing_uncompressed_block():                                # SnappyStreamTests.cs:241-262
    raw = bytes(range(256))
    s = O.frame_encode(raw)
    assert len(s) == 10 + 8 + 256
    assert s[10] == 0x01 and int.from_bytes(s[11:14], "little") == 260
    assert struct.unpack("<I", s[14:18])[0] == O.crc32c(raw, masked=True)
    assert O.frame_decode(s) == raw


def test_framing_rules():
    raw = read_testdata("html") + read_testdata("fireworks.jpeg")
    s = O.frame_encode(raw)
    assert O.frame_decode(s) == raw
    # skippable (0x80..0xfd), padding (0xfe) and a repeated stream identifier (0xff) are skipped unvalidated
    extra = bytes([0x80, 3, 0, 0, 1, 2, 3]) + bytes([0xFE, 2, 0, 0, 0, 0]) + STREAM_HEADER
    assert O.frame_decode(s[:10] + extra + s[10:]) == raw
    with pytest.raises(O.OracleError) as e:
        O.frame_decode(s[:10] + bytes([0x02, 1, 0, 0, 0]) + s[10:])    # reserved unskippable 0x02..0x7f
    assert e.value.status == O.ERR_CHUNK_TYPE
    bad = bytearray(s)
    bad[14] ^= 1                                                       # first chunk's CRC
    with pytest.raises(O.OracleError) as e:
        O.frame_decode(bytes(bad))
    assert e.value.status == O.ERR_CRC_MISMATCH
    with pytest.raises(O.OracleError) as e:
        O.frame_decode(s[:-5])
    assert e.value.status == O.ERR_TRUNCATED_STREAM


def test_synthetic_generators_are_deterministic():
    html = read_testdata("html")
    a = datagen.html_like_blocks(html, 3, 2)
    b = datagen.html_like_blocks(html, 0, 5)
    assert a.tobytes() == b[3 * 65536:].tobytes()
    assert 0.15 < len(O.compress(a[:65536].tobytes())) / 65536 < 0.45
    le = datagen.low_entropy_block(7)
    assert len(O.compress(le.tobytes())) / 65536 < 0.2


def test_batch_helpers_match_single_calls():
    html = read_testdata("html")
    x = datagen.html_like_blocks(html, 0, 6)
    in_off = np.arange(6, dtype=np.uint64) * np.uint64(65536)
    in_len = np.full(6, 65536, dtype=np.uint32)
    in_len[5] = 1000
    out, out_off, out_len, status = O.compress_batch(x, in_off, in_len, threads=3)
    assert (status == 0).all()
    for b in range(6):
        ref = O.compress(x[int(in_off[b]):int(in_off[b]) + int(in_len[b])].tobytes())
        assert out[int(out_off[b]):int(out_off[b]) + int(out_len[b])].tobytes() == ref
    dec, dlen, dst = O.decompress_batch(out, out_off, out_len, in_off, in_len, x.size, threads=2)
    assert (dst == 0).all() and (dlen == in_len).all()
    for b in range(6):
        s, e = int(in_off[b]), int(in_off[b]) + int(in_len[b])
        assert dec[s:e].tobytes() == x[s:e].tobytes()
