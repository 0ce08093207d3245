"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same inputs, bit-exact.
Mirrors Snappier.Tests/SnappyTests.cs, SnappyStreamTests.cs, Internal/*Tests.cs for the hot path.  Needs an MI355X."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest
import torch

import oracle as O
from conftest import CORPUS, GOLDEN, ROOT, read_testdata
import datagen
import kats
import layouts

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import snappier_amd as S
    from snappier_amd import batch as SB, datagen as SD, _native as N
    Snappy = S.Snappy

VARIANTS = [O.HASH_CRC32C, O.HASH_MUL]
OUTDIR = os.path.join(ROOT, "gpurun_out")


def ctx_for(variant):
    return S.default_context(variant)


def dump_mismatch(tag, got: bytes, ref: bytes):
    os.makedirs(OUTDIR, exist_ok=True)
    i = next((k for k, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
    msg = f"{tag}: len got {len(got)} ref {len(ref)} first diff at {i}: got {got[i:i+16].hex()} ref {ref[i:i+16].hex()}"
    with open(os.path.join(OUTDIR, "mismatch.log"), "a") as f:
        f.write(msg + "\n")
    return msg


def assert_same(tag, got: bytes, ref: bytes):
    if got != ref:
        pytest.fail(dump_mismatch(tag, got, ref))


@pytest.fixture(scope="module")
def codec():
    cds = {v: SB.BlockCodec(0, v) for v in VARIANTS}
    yield cds
    for cd in cds.values():                 # (a device's table pool lives as long as its last context: later modules test its lifetime)
        cd.ctx.close()


def to_dev(a: np.ndarray):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def blocks_of(data: bytes):
    """-> (concatenated u8 array, in_off, in_len) for consecutive 64 KiB windows of data."""
    n = len(data)
    nb = max(1, (n + 65535) // 65536)
    off = np.arange(nb, dtype=np.int64) * 65536
    ln = np.minimum(65536, n - off).astype(np.int32)
    return np.frombuffer(data, dtype=np.uint8), off, ln


# ------------------------------------------------------------------ decompress

@pytest.mark.parametrize("name", ["html_x_4.snappy", "alice29.snappy"])
def test_decompress_golden_chunks(name):
    from test_oracle_golden import parse_frames
    for _t, crc, body in parse_frames(read_testdata(name)):
        ref = O.decompress(body)
        assert_same(name, Snappy.DecompressToArray(body), ref)
        assert S.crc32c(ref, masked=True) == crc


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", CORPUS)
def test_decompress_corpus_whole_files(name, variant):            # SnappyTests.cs:8-39 (decode side; multi-fragment blocks)
    data = read_testdata(name)
    comp = O.compress(data, variant)
    assert Snappy.GetUncompressedLength(comp) == len(data)
    assert_same(name, Snappy.DecompressToArray(comp), data)


@pytest.mark.parametrize("name", ["baddata1.snappy", "baddata2.snappy", "baddata3.snappy"])
def test_baddata_files(name):                                      # SnappyTests.cs:287-331
    d = read_testdata(name)
    with pytest.raises(S.InvalidDataException) as e:
        Snappy.DecompressToArray(d)
    assert e.value.status == O.decompress_status(d) == O.ERR_BAD_OFFSET
    out = np.empty(Snappy.GetUncompressedLength(d), dtype=np.uint8)
    with pytest.raises(S.InvalidDataException):
        Snappy.TryDecompress(d, out)


def test_decoder_error_taxonomy_matches_oracle():
    vectors = [
        b"", bytes([5, 0x00]), bytes([4, 0x0C, 97, 98, 99, 100]), bytes([4, 0x10, 97, 98, 99, 100, 101]),
        bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x00]), bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x05]),
        bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x04]), bytes([8, 0x00, 97, 0x1A, 0x01, 0x00]),
        bytes([6, 0x00, 97, 0x13, 0x01, 0x00, 0x00, 0x00]), bytes([8, 0x0C, 97, 98, 99, 100, 0x02]),
        bytes([0x80]), bytes([0xFF] * 6), bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x7F, 0]), bytes([0]),
        bytes([3, 0xFC, 0xFF, 0xFF, 0xFF, 0xFF, 1, 2, 3]),            # 4-byte literal length 2^32: partial literal then stop
        bytes([70, 0x00, 97]) + bytes([0xFE, 0x01, 0x00]) + bytes([0x12, 0x01, 0x00]),   # 64-byte + 5-byte pattern copies
    ]
    N = S._native
    ctx = S.default_context()
    for v in vectors:
        src = np.frombuffer(v, dtype=np.uint8)
        cap = 128
        out = np.zeros(cap, dtype=np.uint8)
        import ctypes as C
        w = C.c_size_t(0)
        st = N.lib().snp_try_decompress(ctx.handle, C.c_void_p(src.ctypes.data if src.size else None), src.size,
                                        C.c_void_p(out.ctypes.data), cap, C.byref(w))
        ref_out = np.zeros(cap, dtype=np.uint8)
        rw = C.c_size_t(0)
        ref = O.lib().orc_decompress(src.ctypes.data if src.size else None, src.size, ref_out.ctypes.data, cap, C.byref(rw))
        assert st == ref, (v.hex(), st, ref)
        if ref == 0:
            assert w.value == rw.value and out[:w.value].tobytes() == ref_out[:rw.value].tobytes(), v.hex()


def test_bad_long_length_and_small_buffers():                     # SnappyTests.cs:218-285
    comp = bytearray(O.compress(b"A" * 1000))
    comp[0], comp[1] = 255, 127
    with pytest.raises(S.InvalidDataException):
        Snappy.DecompressToArray(bytes(comp))
    comp = O.compress(b"A" * 100000)
    out = np.empty(100, dtype=np.uint8)
    with pytest.raises(S.InsufficientBufferException):
        Snappy.Decompress(comp, out)
    assert Snappy.TryDecompress(comp, out) == (False, 0)
    ok, n = Snappy.TryCompress(b"A" * 100000, np.empty(10, dtype=np.uint8))
    assert (ok, n) == (False, 0)
    assert Snappy.TryCompress(b"", np.empty(0, dtype=np.uint8)) == (False, 0)          # Snappy.cs:57-62
    buf = np.zeros(2048, dtype=np.uint8)
    with pytest.raises(S.InvalidOperationException):                                    # SnappyTests.cs:204-210
        Snappy.Compress(buf[:1024], buf[1023:])


def test_compress_and_decompress_limited_output_buffer():        # SnappyTests.cs:41-63
    """Output buffer smaller than GetMaxCompressedLength but larger than the actual compressed length
    (SnappyCompressor.cs:56-74: compress to scratch, then copy), down to exactly the compressed length; one byte less fails."""
    data = read_testdata("alice29.txt")[:65536]
    ref = O.compress(data)
    for cap in (Snappy.GetMaxCompressedLength(len(data)) - 5, len(ref) + 1, len(ref)):
        buf = np.zeros(cap, dtype=np.uint8)
        n = Snappy.Compress(data, buf)
        assert n == len(ref) and buf[:n].tobytes() == ref
        out = np.zeros(Snappy.GetUncompressedLength(buf[:n].tobytes()), dtype=np.uint8)
        assert Snappy.Decompress(buf[:n].tobytes(), out) == len(data) and out.tobytes() == data
    assert Snappy.TryCompress(data, np.zeros(len(ref) - 1, dtype=np.uint8)) == (False, 0)
    with pytest.raises(S.InsufficientBufferException):
        Snappy.Compress(data, np.zeros(len(ref) - 1, dtype=np.uint8))
    big = read_testdata("alice29.txt")                                                  # multi-fragment, same rule
    ref = O.compress(big)
    buf = np.zeros(len(ref), dtype=np.uint8)
    assert Snappy.Compress(big, buf) == len(ref) and buf.tobytes() == ref


@pytest.mark.parametrize("expected,s1,s2,length", kats.FIND_MATCH_LENGTH)
def test_find_match_length_kats_on_the_device(expected, s1, s2, length):   # SnappyCompressorTests.cs:10-96
    """The reference's FindMatchLength KATs through the device function the compressor uses (wave_match_extend):
    buffer = s1 || s2, candidate at 0, position at len(s1), fragment end = the KAT's s2Limit."""
    import ctypes as C
    L = S.lib()
    buf = to_dev(np.frombuffer((s1 + s2).encode() + bytes(16), dtype=np.uint8))
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = L.snp_debug_match_length(C.c_void_p(buf.data_ptr()), C.c_uint32(len(s1) + length), C.c_uint32(len(s1)), C.c_uint32(0),
                                  C.c_uint32(0), C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and int(out.item()) == expected == O.find_match_length(s1.encode(), s2.encode(), length)


@pytest.mark.parametrize("expected,s1,s2,length", kats.FIND_MATCH_LENGTH)
def test_find_match_length_kats_on_the_lane_kernels_function(expected, s1, s2, length):   # SnappyCompressorTests.cs:10-96
    """The same KATs through the HEADLINE kernel's form of FindMatchLength (compress_lanes.hip, lane_find_match_length; hook declared
    in include/snappier_hip_debug.h): s1 at 0, s2 at len(s1), fragment end = the KAT's s2Limit."""
    import ctypes as C
    L = S.lib()
    buf = to_dev(np.frombuffer((s1 + s2).encode() + bytes(16), dtype=np.uint8))
    out = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = L.snp_debug_lane_match_length(C.c_void_p(buf.data_ptr()), C.c_uint32(len(s1) + length), C.c_uint32(0), C.c_uint32(len(s1)),
                                       C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == 0 and int(out.item()) == expected == O.find_match_length(s1.encode(), s2.encode(), length)


# ------------------------------------------------------------------ compress (bit-exact against the oracle)

@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", CORPUS)
def test_compress_corpus_whole_files(name, variant):               # multi-fragment TryCompress  SnappyCompressor.cs:40-80
    data = read_testdata(name)
    got = Snappy.CompressToArray(data, ctx_for(variant))
    assert_same(f"{name}/v{variant}", got, O.compress(data, variant))
    assert len(got) <= Snappy.GetMaxCompressedLength(len(data))


def test_compress_matches_golden_fixtures():
    html = read_testdata("html")[:65536]
    for variant, (n, sha) in ((O.HASH_MUL, kats.HTML64K_MUL), (O.HASH_CRC32C, kats.HTML64K_CRC)):
        c = Snappy.CompressToArray(html, ctx_for(variant))
        assert len(c) == n and hashlib.sha256(c).hexdigest() == sha
    g = json.load(open(os.path.join(GOLDEN, "libsnappy_mul_goldens.json")))
    for name, r in g["whole_files"].items():
        c = Snappy.CompressToArray(read_testdata(name), ctx_for(O.HASH_MUL))
        assert len(c) == r["clen"] and hashlib.sha256(c).hexdigest() == r["sha256"], name


@pytest.mark.parametrize("variant", VARIANTS)
def test_compress_batch_all_corpus_windows(codec, variant):
    """Every 64 KiB window of every corpus file in ONE batch launch, ragged last windows included."""
    parts, offs, lens, pos = [], [], [], 0
    for name in CORPUS:
        a, off, ln = blocks_of(read_testdata(name))
        parts.append(a)
        offs.append(off + pos)
        lens.append(ln)
        pos += a.size
    data, in_off, in_len = np.concatenate(parts), np.concatenate(offs), np.concatenate(lens)
    cd = codec[variant]
    out, out_off, out_len, status = cd.compress(to_dev(data), to_dev(in_off), to_dev(in_len))
    torch.cuda.synchronize()
    out, out_off, out_len, status = out.cpu().numpy(), out_off.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy()
    assert (status == 0).all()
    for b in range(len(in_len)):
        ref = O.compress(data[in_off[b]:in_off[b] + in_len[b]].tobytes(), variant)
        got = out[out_off[b]:out_off[b] + out_len[b]].tobytes()
        assert_same(f"window {b} v{variant}", got, ref)
    if variant == O.HASH_MUL:                                       # and against libsnappy 1.1.8's bytes directly
        g = json.load(open(os.path.join(GOLDEN, "libsnappy_mul_goldens.json")))
        b = 0
        for name in CORPUS:
            nwin = len(blocks_of(read_testdata(name))[1])
            rows = {r["start"]: r for r in g["files"].get(name, [])}
            for wdx in range(nwin):
                r = rows.get(wdx * 65536)
                if r:
                    got = out[out_off[b + wdx]:out_off[b + wdx] + out_len[b + wdx]].tobytes()
                    assert len(got) == r["clen"] and hashlib.sha256(got).hexdigest() == r["sha256"], (name, wdx)
            b += nwin


@pytest.mark.parametrize("variant", VARIANTS)
def test_compress_small_and_edge_lengths(codec, variant):
    html = read_testdata("html")
    lens = [0, 1, 2, 3, 14, 15, 16, 17, 30, 31, 32, 33, 48, 63, 64, 65, 127, 128, 255, 256, 257, 511, 512, 513, 1000,
            4095, 4096, 16383, 16384, 16385, 32768, 65535, 65536]
    data = np.frombuffer(html, dtype=np.uint8)
    in_off = np.array([(7 * i) % 1000 for i in range(len(lens))], dtype=np.int64)   # unaligned starts on purpose
    in_len = np.array(lens, dtype=np.int32)
    cd = codec[variant]
    out, out_off, out_len, status = cd.compress(to_dev(data), to_dev(in_off), to_dev(in_len))
    torch.cuda.synchronize()
    out, out_off, out_len = out.cpu().numpy(), out_off.cpu().numpy(), out_len.cpu().numpy()
    assert (status.cpu().numpy() == 0).all()
    for b, n in enumerate(lens):
        ref = O.compress(html[in_off[b]:in_off[b] + n], variant)
        assert_same(f"len {n} v{variant}", out[out_off[b]:out_off[b] + out_len[b]].tobytes(), ref)


@pytest.mark.parametrize("s", kats.STRING_CASES)
def test_string_cases(s):                                          # SnappyTests.cs:178-202
    for variant in VARIANTS:
        c = Snappy.CompressToArray(s, ctx_for(variant))
        assert_same("string", c, O.compress(s, variant))
        assert_same("string rt", Snappy.DecompressToArray(c), s)


def test_random_data(codec):                                       # SnappyTests.cs:401-446 (own PRNG; property + parity)
    rng = np.random.default_rng(301)
    cases = [datagen.random_data_case(i, rng) for i in list(range(24)) + list(range(100, 1100))]
    small = [c for c in cases if len(c) <= 65536]
    big = [c for c in cases if len(c) > 65536]
    for variant in VARIANTS:
        cd = codec[variant]
        data = np.frombuffer(b"".join(small) + b"\0", dtype=np.uint8)
        in_len = np.array([len(c) for c in small], dtype=np.int32)
        in_off = np.concatenate([[0], np.cumsum(in_len[:-1], dtype=np.int64)]).astype(np.int64)
        d_data, d_off, d_len = to_dev(data), to_dev(in_off), to_dev(in_len)
        out, out_off, out_len, status = cd.compress(d_data, d_off, d_len)
        back = torch.zeros_like(d_data)
        dlen, dst = cd.decompress(out, out_off, out_len, back, d_off, d_len)
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all() and (dst.cpu().numpy() == 0).all()
        assert torch.equal(dlen, d_len) and torch.equal(back[:-1], d_data[:-1])
        o, oo, ol = out.cpu().numpy(), out_off.cpu().numpy(), out_len.cpu().numpy()
        for b in range(0, len(small), 7):
            assert_same(f"random {b}", o[oo[b]:oo[b] + ol[b]].tobytes(), O.compress(small[b], variant))
        for c in big:
            comp = Snappy.CompressToArray(c, ctx_for(variant))
            assert_same("random big", comp, O.compress(c, variant))
            assert_same("random big rt", Snappy.DecompressToArray(comp), c)


# ------------------------------------------------------------------ synthetic configs (BASELINE.json configs 2, 3, 5)

def test_device_generators_match_cpu_statement():
    html = read_testdata("html")
    a = SD.html_like_blocks(html, 5, 3, "cuda").cpu().numpy()
    assert a.tobytes() == datagen.html_like_blocks(html, 5, 3).tobytes()
    le = SD.low_entropy_blocks(7, 2, "cuda").cpu().numpy()
    assert le.tobytes() == np.concatenate([datagen.low_entropy_block(7), datagen.low_entropy_block(8)]).tobytes()
    files = [read_testdata(n) for n in CORPUS]
    m = SD.corpus_blocks(files, 9, 13, SD.MIXED_SEED, "cuda").cpu().numpy()
    assert m.tobytes() == datagen.corpus_blocks(files, 9, 13, datagen.MIXED_SEED).tobytes()


def _roundtrip_blocks(cd, raw: torch.Tensor, nb: int, variant: int, check_oracle_every: int):
    in_off, in_len = cd.uniform_layout(nb)
    out, out_off, out_len, status = cd.compress(raw, in_off, in_len)
    back = torch.empty_like(raw)
    dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len)
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0 and int((dst != 0).sum()) == 0
    assert torch.equal(dlen, in_len)
    assert torch.equal(back, raw)                                      # encode -> decode is the identity
    o_len = out_len.cpu().numpy()
    for b in range(0, nb, check_oracle_every):                         # sampled bit-exact parity with the oracle
        blk = raw[b * 65536:(b + 1) * 65536].cpu().numpy().tobytes()
        got = out[b * cd.comp_stride: b * cd.comp_stride + int(o_len[b])].cpu().numpy().tobytes()
        assert_same(f"block {b} v{variant}", got, O.compress(blk, variant))
    return o_len


@pytest.mark.parametrize("variant", VARIANTS)
def test_config2_html_like_blocks(codec, variant):
    nb = 2048
    raw = SD.html_like_blocks(read_testdata("html"), 0, nb, "cuda")
    o_len = _roundtrip_blocks(codec[variant], raw, nb, variant, 64)
    assert 0.2 < o_len.sum() / (nb * 65536) < 0.4


@pytest.mark.parametrize("variant", VARIANTS)
def test_config3_low_entropy_blocks(codec, variant):               # stresses overlapping / pattern copies
    nb = 2048
    raw = SD.low_entropy_blocks(0, nb, "cuda")
    o_len = _roundtrip_blocks(codec[variant], raw, nb, variant, 64)
    assert o_len.sum() / (nb * 65536) < 0.2


def test_config5_mixed_corpus_blocks(codec):
    nb = 2048
    raw = SD.corpus_blocks([read_testdata(n) for n in CORPUS], 0, nb, SD.MIXED_SEED, "cuda")
    _roundtrip_blocks(codec[O.HASH_CRC32C], raw, nb, O.HASH_CRC32C, 37)


def test_zero_and_incompressible_blocks(codec):
    cd = codec[O.HASH_CRC32C]
    zeros = torch.zeros(4 * 65536, dtype=torch.uint8, device="cuda")
    o_len = _roundtrip_blocks(cd, zeros, 4, O.HASH_CRC32C, 1)
    assert (o_len == 3077).all()                                       # 1 literal + 1024 copies of 64
    g = torch.Generator(device="cuda").manual_seed(5)
    noise = torch.randint(0, 256, (8 * 65536,), dtype=torch.uint8, device="cuda", generator=g)
    _roundtrip_blocks(cd, noise, 8, O.HASH_CRC32C, 1)


def test_decompress_batch_per_block_status(codec):
    """Good and bad blocks in one launch: each block reports its own status (SURVEY 8b 'per-block status array')."""
    cd = codec[O.HASH_CRC32C]
    good = O.compress(read_testdata("html")[:65536])
    blobs = [good, read_testdata("baddata1.snappy"), good[:1000], bytes([4, 0x10, 97, 98, 99, 100, 101]), good]
    caps = [65536, 128082, 65536, 64, 100]
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    in_len = np.array([len(b) for b in blobs], dtype=np.int32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1])]).astype(np.int64)
    out_cap = np.array(caps, dtype=np.int32)
    out_off = np.concatenate([[0], np.cumsum(out_cap[:-1])]).astype(np.int64)
    out = torch.zeros(int(out_cap.sum()), dtype=torch.uint8, device="cuda")
    dlen, dst = cd.decompress(to_dev(data), to_dev(in_off), to_dev(in_len), out, to_dev(out_off), to_dev(out_cap))
    torch.cuda.synchronize()
    expect = [O.decompress_status(b, c) for b, c in zip(blobs, caps)]
    assert dst.cpu().tolist() == expect == [0, O.ERR_BAD_OFFSET, O.ERR_INCOMPLETE, O.ERR_TOO_LONG, O.ERR_OUTPUT_TOO_SMALL]
    assert out[:65536].cpu().numpy().tobytes() == read_testdata("html")[:65536]


@pytest.mark.parametrize("layout", layouts.COMPRESS_LAYOUTS)
def test_compress_layouts_are_bit_identical(layout):
    """Both compressor layouts (one fragment per wavefront with the table in LDS or in a global slot -- the window kernel; one fragment per lane
    with the table in an HBM workspace, under every store / probe option) must give the oracle's bytes on every kind of input, ragged lengths
    included.  Selected through snp_ctx_set_option on the product library."""
    html = read_testdata("html")
    for variant in VARIANTS:
        cd = SB.BlockCodec(0, variant)
        layouts.set_compress_layout(cd.ctx, layout)
        nb = 300
        for raw in (SD.html_like_blocks(html, 11, nb, "cuda"), SD.low_entropy_blocks(3, nb, "cuda"),
                    SD.corpus_blocks([read_testdata(n) for n in CORPUS], 2, nb, SD.MIXED_SEED, "cuda")):
            _roundtrip_blocks(cd, raw, nb, variant, 23)
        lens = [0, 1, 14, 15, 16, 17, 31, 32, 33, 255, 256, 257, 1000, 4096, 16383, 16384, 16385, 65535, 65536]
        data = np.frombuffer(html, dtype=np.uint8)
        in_off = np.array([(7 * i) % 1000 for i in range(len(lens))], dtype=np.int64)
        in_len = np.array(lens, dtype=np.int32)
        out, out_off, out_len, status = cd.compress(to_dev(data), to_dev(in_off), to_dev(in_len))
        torch.cuda.synchronize()
        out, out_off, out_len = out.cpu().numpy(), out_off.cpu().numpy(), out_len.cpu().numpy()
        assert (status.cpu().numpy() == 0).all()
        for b, n in enumerate(lens):
            assert_same(f"{layout} len {n} v{variant}", out[out_off[b]:out_off[b] + out_len[b]].tobytes(),
                        O.compress(html[in_off[b]:in_off[b] + n], variant))
        # whole files through the host API (multi-fragment)
        for name in ("html_x_4", "kppkn.gtb", "fireworks.jpeg"):
            assert_same(f"{layout} {name}", Snappy.CompressToArray(read_testdata(name), layouts.set_compress_layout(S.Context(0, variant), layout)),
                        O.compress(read_testdata(name), variant))


@pytest.mark.parametrize("decode", ["chains", "wave-only", "serial", "small", "small-grid"])
@pytest.mark.parametrize("fenced", ["0", "1"])
def test_decode_kernel_variants_agree(fenced, decode):
    """Same-wave store->load ordering: the default kernel relies on in-order vector memory; the fenced variant drains
    vmcnt before touching young output.  The token-parallel front end, the small-block pre-pass with its two leftover forms and the serial
    loop must also agree.  Every variant must be exact on the overlap-heavy config, the html-like config and the mixed corpus."""
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    layouts.set_decode_layout(cd.ctx, decode, fenced)
    nb = 512
    _roundtrip_blocks(cd, SD.low_entropy_blocks(1000, nb, "cuda"), nb, O.HASH_CRC32C, 128)
    _roundtrip_blocks(cd, SD.html_like_blocks(read_testdata("html"), 77, nb, "cuda"), nb, O.HASH_CRC32C, 128)
    _roundtrip_blocks(cd, SD.corpus_blocks([read_testdata(n) for n in CORPUS], 5, nb, SD.MIXED_SEED, "cuda"), nb,
                      O.HASH_CRC32C, 128)
    # statuses of broken blocks are identical too
    blobs = [read_testdata("baddata1.snappy"), O.compress(read_testdata("html")[:65536])[:1000],
             bytes([4, 0x10, 97, 98, 99, 100, 101]) + bytes(100), bytes([0x80]), bytes([0xFF] * 6), b"",
             bytes([3, 0xFC, 0xFF, 0xFF, 0xFF, 0xFF, 1, 2, 3]), bytes([8, 0x0C, 97, 98, 99, 100, 0x02]),
             bytes([70, 0x00, 97]) + bytes([0xFE, 0x01, 0x00]) + bytes([0x12, 0x01, 0x00]), read_testdata("baddata3.snappy")]
    caps = [128082, 65536, 64, 16, 16, 16, 16, 16, 128, 130378]
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    in_len = np.array([len(b) for b in blobs], dtype=np.int32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1])]).astype(np.int64)
    out_cap = np.array(caps, dtype=np.int32)
    out_off = np.concatenate([[0], np.cumsum(out_cap[:-1])]).astype(np.int64)
    out = torch.zeros(int(out_cap.sum()), dtype=torch.uint8, device="cuda")
    _dlen, dst = cd.decompress(to_dev(data), to_dev(in_off), to_dev(in_len), out, to_dev(out_off), to_dev(out_cap))
    torch.cuda.synchronize()
    assert dst.cpu().tolist() == [O.decompress_status(b, c) for b, c in zip(blobs, caps)]


@pytest.mark.parametrize("decode", ["chains", "wave-only", "serial", "small"])
def test_streams_built_against_the_sub_chain_decoder(decode):
    """Legal Snappy that no 64 KiB-fragment compressor emits, chosen so that the guessed chains of the sub-chain front end
    (decompress.hip, FRONT = 3) rarely or never land on a tag start: 5- and 7-byte tag periods, literal bodies made of
    long-literal tag bytes, copy-2 offsets that read as long literals.  Output and status must equal the oracle's through
    every front end, also when such streams are truncated or their capacity is one byte short."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("adversarial_streams", os.path.join(ROOT, "scripts", "adversarial_streams.py"))
    A = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(A)
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    layouts.set_decode_layout(cd.ctx, decode)
    blobs, caps = [], []
    for kind in A.KINDS:
        s = A.build(kind)
        for cut, cap in ((len(s), 65536), (len(s), 65535), (len(s) - 1, 65536), (len(s) // 2, 65536), (2100, 65536)):
            blobs.append(s[:cut])
            caps.append(cap)
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    in_len = np.array([len(b) for b in blobs], dtype=np.int32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1])]).astype(np.int64)
    out_cap = np.array(caps, dtype=np.int32)
    out_off = (np.arange(len(blobs), dtype=np.int64) * 65600)
    out = torch.zeros(len(blobs) * 65600, dtype=torch.uint8, device="cuda")
    dlen, dst = cd.decompress(to_dev(data), to_dev(in_off), to_dev(in_len), out, to_dev(out_off), to_dev(out_cap))
    torch.cuda.synchronize()
    dlen, dst, h_out = dlen.cpu().numpy(), dst.cpu().numpy(), out.cpu().numpy()
    want_st = [O.decompress_status(b, c) for b, c in zip(blobs, caps)]
    assert dst.tolist() == want_st
    for i, (b, c) in enumerate(zip(blobs, caps)):
        if want_st[i] == 0:
            ref = O.decompress(b, c)
            assert dlen[i] == len(ref)
            assert_same(f"{decode} stream {i}", h_out[out_off[i]: out_off[i] + dlen[i]].tobytes(), ref)


def test_one_context_through_batches_of_changing_block_sizes():
    """The decode policy of a context follows its previous batch (small-block pre-pass layout by mean block size; no pre-pass after
    a batch of large blocks; capi_batch.hip launch_decompress).  One context, batches of 40 / 200 / 65536 / 300 / 65536 / 40 / 1000-byte
    blocks, each decoded twice (the second call sees the first one's read-back): every block round-trips, a sample equals the
    oracle, whatever the context remembered."""
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    raw_all = SD.corpus_blocks([read_testdata(n) for n in CORPUS], 11, 4096, SD.MIXED_SEED, "cuda")     # 256 MiB
    for bs in (40, 200, 65536, 300, 65536, 40, 1000):
        nb = min(raw_all.numel() // bs, 65536)
        assert nb >= 4096                                                 # (the small-block pre-pass is for batches of >= 4096 blocks)
        raw = raw_all[: nb * bs]
        in_off, in_len = cd.uniform_layout(nb, bs)
        out, out_off, out_len, status = cd.compress(raw, in_off, in_len)
        assert int((status != 0).sum()) == 0
        h_out, h_off, h_len, h_raw = out.cpu().numpy(), out_off.cpu().numpy(), out_len.cpu().numpy(), raw.cpu().numpy()
        for b in range(0, nb, max(1, nb // 64)):
            assert_same(f"bs {bs} block {b}", h_out[h_off[b]: h_off[b] + h_len[b]].tobytes(), O.compress(h_raw[b * bs: (b + 1) * bs].tobytes()))
        for rep in range(2):
            back = torch.zeros_like(raw)
            dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len)
            torch.cuda.synchronize()
            assert int((dst != 0).sum()) == 0 and bool((dlen == bs).all()), f"bs {bs} call {rep}"
            assert torch.equal(back, raw), f"bs {bs} call {rep}"


@pytest.mark.parametrize("bs", [64, 256, 1000, 4096])
def test_many_small_blocks_roundtrip_and_parity(bs):               # SURVEY 8(f4); SnappyStreamTests.cs:145-192 pattern
    """Batches of small blocks take the block-per-lane decoder and the lane compressor with small tables: every block
    round-trips, a sample equals the oracle byte for byte, and corrupted small blocks report the oracle's status."""
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    total = 8 << 20
    nb = total // bs
    raw = SD.corpus_blocks([read_testdata(n) for n in CORPUS], 3, total // 65536, SD.MIXED_SEED, "cuda")[: nb * bs]
    in_off, in_len = cd.uniform_layout(nb, bs)
    out, out_off, out_len, status = cd.compress(raw, in_off, in_len)
    back = torch.zeros_like(raw)
    dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len)
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0 and int((dst != 0).sum()) == 0 and bool((dlen == bs).all())
    assert torch.equal(back, raw)
    lens = out_len.cpu().numpy()
    h_raw, h_out, h_off = raw.cpu().numpy(), out.cpu().numpy(), out_off.cpu().numpy()
    for b in range(0, nb, max(1, nb // 200)):
        assert h_out[h_off[b]: h_off[b] + lens[b]].tobytes() == O.compress(h_raw[b * bs:(b + 1) * bs].tobytes()), b
    # corrupt one byte in every 7th block: status (and bytes when it still decodes) as the oracle
    rng = np.random.default_rng(bs)
    h_out = h_out.copy()
    victims = list(range(1, nb, 7))[:4000]
    for b in victims:
        h_out[h_off[b] + int(rng.integers(0, lens[b]))] ^= int(rng.integers(1, 256))
    back.zero_()
    dlen, dst = cd.decompress(to_dev(h_out), out_off, out_len, back, in_off, in_len)
    torch.cuda.synchronize()
    dst, dlen, h_back = dst.cpu().numpy(), dlen.cpu().numpy(), back.cpu().numpy()
    for b in victims[:600]:
        blob = h_out[h_off[b]: h_off[b] + lens[b]].tobytes()
        want = O.decompress_status(blob, bs)
        assert dst[b] == want, (b, dst[b], want)
        if want == 0:
            assert h_back[b * bs: b * bs + dlen[b]].tobytes() == O.decompress(blob)
    clean = np.ones(nb, dtype=bool)
    clean[victims] = False
    assert (dst[clean] == 0).all()


def test_concat_batch_compacts_the_strided_blocks(codec):
    """snp_concat_batch: the compressed blocks sit at a fixed stride after snp_compress_batch; compaction must give exactly
    their concatenation (what SnappyCompressor.TryCompress produces by advancing its output span, :40-80; what a rank sends in
    the payload gather)."""
    cd = codec[O.HASH_CRC32C]
    nb = 777
    raw = SD.corpus_blocks([read_testdata(n) for n in CORPUS], 9, nb, SD.MIXED_SEED, "cuda")
    in_off, in_len = cd.uniform_layout(nb)
    out, out_off, out_len, status = cd.compress(raw, in_off, in_len)
    stream, dst_off = cd.compact(out, out_off, out_len)
    torch.cuda.synchronize()
    h_out, h_off, h_len = out.cpu().numpy(), out_off.cpu().numpy(), out_len.cpu().numpy()
    want = b"".join(h_out[h_off[b]: h_off[b] + h_len[b]].tobytes() for b in range(nb))
    assert stream.cpu().numpy().tobytes() == want
    assert dst_off.cpu().tolist() == np.concatenate([[0], np.cumsum(h_len[:-1].astype(np.int64))]).tolist()
    empty, _ = cd.compact(out, out_off[:0], out_len[:0])
    assert empty.numel() == 0


# ------------------------------------------------------------------ CRC-32C

@pytest.mark.parametrize("data,expected", kats.CRC32C)
def test_crc32c_kats(data, expected):                              # Crc32CAlgorithmTests.cs:7-24
    assert S.crc32c(data) == expected
    assert S.crc32c(data, masked=True) == O.crc32c(data, masked=True)


def test_crc32c_lengths_and_alignment(codec):
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, 300000, dtype=np.uint8)
    lens = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 256, 257, 259, 260, 511, 512, 1023, 1024, 1025,
            4096, 65535, 65536, 65537, 100000]
    in_off = np.array([(13 * i) % 97 for i in range(len(lens))], dtype=np.int64)
    in_len = np.array(lens, dtype=np.int32)
    for masked in (False, True):
        crc = codec[O.HASH_CRC32C].crc32c(to_dev(data), to_dev(in_off), to_dev(in_len), masked=masked)
        torch.cuda.synchronize()
        got = crc.cpu().numpy().view(np.uint32)
        ref = O.crc32c_batch(data, in_off.astype(np.uint64), in_len.astype(np.uint32), masked)
        assert got.tolist() == ref.tolist()


# ------------------------------------------------------------------ framing (SnappyStream)

def test_frame_encode_reproduces_golden_streams():
    """hash=MUL: the whole framed stream equals the reference's own html_x_4.snappy / alice29.snappy byte-for-byte."""
    for name in ("html_x_4.snappy", "alice29.snappy"):
        stream = read_testdata(name)
        raw = O.frame_decode(stream)
        assert_same(name, S.frame_decode(stream), raw)
        assert_same(name, S.frame_encode(raw, ctx_for(O.HASH_MUL)), stream)
        assert_same(name, S.frame_encode(raw, ctx_for(O.HASH_CRC32C)), O.frame_encode(raw, O.HASH_CRC32C))


@pytest.mark.parametrize("name", CORPUS)
def test_stream_roundtrip(name):                                   # SnappyStreamTests.cs:8-143
    import io
    data = read_testdata(name)
    sink = io.BytesIO()
    with S.SnappyStream(sink, S.CompressionMode.Compress, leaveOpen=True) as z:
        for s in range(0, len(data), 50000):                       # writes that straddle chunk boundaries
            z.write(data[s:s + 50000])
    framed = sink.getvalue()
    assert_same(name, framed, O.frame_encode(data))
    with S.SnappyStream(io.BytesIO(framed), S.CompressionMode.Decompress) as z:
        assert_same(name, z.read(), data)


def test_known_8192_byte_chunk_stress():                           # SnappyStreamTests.cs:196-216
    """The reference's streamerrorsequence.txt fixture (hex text, 508 KB of binary): frame it, read it back in the 8192-byte
    reads that once broke the reference's decoder, compare with the input and with the oracle's stream."""
    import io
    data = bytes.fromhex(read_testdata("streamerrorsequence.txt").decode("ascii").strip())
    sink = io.BytesIO()
    with S.SnappyStream(sink, S.CompressionMode.Compress, leaveOpen=True) as z:
        z.write(data)
        z.flush()
    framed = sink.getvalue()
    assert_same("streamerrorsequence framed", framed, O.frame_encode(data))
    out = bytearray()
    with S.SnappyStream(io.BytesIO(framed), S.CompressionMode.Decompress) as z:
        while True:
            b = z.read(8192)
            if not b:
                break
            out += b
    assert_same("streamerrorsequence rt", bytes(out), data)
    assert_same("streamerrorsequence block", Snappy.DecompressToArray(Snappy.CompressToArray(data)), data)


def test_stream_chunk_stress():                                    # SnappyStreamTests.cs:145-192
    """Random 1..100-byte writes with a Flush after each: hundreds of tiny chunks; every flush boundary becomes a chunk
    boundary, exactly as SnappyStreamCompressor.Flush (:82-97) does."""
    import io
    rng = np.random.default_rng(123)
    data = rng.integers(0, 256, 9000, dtype=np.uint8).tobytes() + bytes(3000) + read_testdata("html")[:5000]
    sink = io.BytesIO()
    pieces, pos = [], 0
    with S.SnappyStream(sink, S.CompressionMode.Compress, leaveOpen=True) as z:
        while pos < len(data):
            k = int(rng.integers(1, 101))
            z.write(data[pos:pos + k])
            z.flush()
            pieces.append(data[pos:pos + k])
            pos += k
    framed = sink.getvalue()
    ref = b"".join(O.frame_encode(p)[10 if i else 0:] for i, p in enumerate(pieces))   # stream identifier only once
    assert_same("chunk stress", framed, ref)
    with S.SnappyStream(io.BytesIO(framed), S.CompressionMode.Decompress) as z:
        out = bytearray()
        while True:                                                # mid-chunk reads of odd sizes
            b = z.read(int(rng.integers(1, 777)))
            if not b:
                break
            out += b
    assert_same("chunk stress rt", bytes(out), data)
    assert_same("chunk stress oracle", O.frame_decode(framed), data)


def test_framing_uncompressed_block_and_rules():                   # SnappyStreamTests.cs:241-262 + format rules
    raw = bytes(range(256))
    s = S.frame_encode(raw)
    assert len(s) == 10 + 8 + 256 and s[10] == 0x01
    assert struct.unpack("<I", s[14:18])[0] == O.crc32c(raw, masked=True)
    assert S.frame_decode(s) == raw
    assert S.frame_encode(b"") == s[:10] and S.frame_decode(s[:10]) == b""
    raw = read_testdata("html") + read_testdata("fireworks.jpeg")
    s = S.frame_encode(raw)
    assert_same("mixed", s, O.frame_encode(raw))
    header = s[:10]
    extra = bytes([0x80, 3, 0, 0, 1, 2, 3]) + bytes([0xFE, 2, 0, 0, 0, 0]) + header
    assert S.frame_decode(header + extra + s[10:]) == raw
    for mutate, status in ((lambda b: b[:10] + bytes([0x02, 1, 0, 0, 0]) + b[10:], O.ERR_CHUNK_TYPE),
                           (lambda b: b[:14] + bytes([b[14] ^ 1]) + b[15:], O.ERR_CRC_MISMATCH),
                           (lambda b: b[:-5], O.ERR_TRUNCATED_STREAM)):
        with pytest.raises(S.InvalidDataException) as e:
            S.frame_decode(mutate(s))
        assert e.value.status == status
    # an error in an earlier chunk wins over a later header error, as in the sequential reference
    bad = bytearray(s + bytes([0x05, 0, 0, 0]))
    bad[14] ^= 1
    with pytest.raises(S.InvalidDataException) as e:
        S.frame_decode(bytes(bad))
    assert e.value.status == O.ERR_CRC_MISMATCH


def test_frame_encode_device_resident(codec):
    cd = codec[O.HASH_CRC32C]
    nb = 300
    raw = SD.html_like_blocks(read_testdata("html"), 0, nb, "cuda")[: nb * 65536 - 12345]   # ragged last chunk
    framed, written = cd.frame_encode(raw)
    torch.cuda.synchronize()
    w = int(written.item())
    got = framed[:w].cpu().numpy().tobytes()
    assert_same("device framing", got, O.frame_encode(raw.cpu().numpy().tobytes()))
    assert_same("device framing rt", S.frame_decode(got), raw.cpu().numpy().tobytes())


def test_frame_decode_chunks_device_resident(codec):
    """Device-resident decode of a framed stream from its chunk table: raw and compressed chunks in one launch, every
    chunk CRC-verified on the device; a flipped CRC or a corrupt body is reported for exactly that chunk."""
    cd = codec[O.HASH_CRC32C]
    raw = read_testdata("html") + read_testdata("fireworks.jpeg") + read_testdata("alice29.txt")   # jpeg -> raw (type 1) chunks
    framed = np.frombuffer(O.frame_encode(raw), dtype=np.uint8).copy()
    pos, types, boff, blen, crcs = 10, [], [], [], []
    while pos < framed.size:
        t = int(framed[pos]); size = int(framed[pos + 1]) | (int(framed[pos + 2]) << 8) | (int(framed[pos + 3]) << 16)
        types.append(t); crcs.append(int.from_bytes(framed[pos + 4:pos + 8].tobytes(), "little"))
        boff.append(pos + 8); blen.append(size - 4)
        pos += 4 + size
    nc = len(types)
    assert 1 in types and 0 in types
    caps = np.minimum(65536, len(raw) - 65536 * np.arange(nc)).astype(np.int32)
    out_off = (np.arange(nc, dtype=np.int64) * 65536)
    crc_arr = np.array(crcs, dtype=np.uint32)

    def run(fr, crc):
        out = torch.zeros(len(raw), dtype=torch.uint8, device="cuda")
        dlen, dst = cd.frame_decode_chunks(to_dev(fr), to_dev(np.array(types, dtype=np.uint8)), to_dev(np.array(boff, dtype=np.int64)),
                                           to_dev(np.array(blen, dtype=np.int32)), to_dev(crc.view(np.int32)), out, to_dev(out_off), to_dev(caps))
        torch.cuda.synchronize()
        return out.cpu().numpy().tobytes(), dlen.cpu().numpy(), dst.cpu().numpy()

    got, dlen, dst = run(framed, crc_arr)
    assert (dst == 0).all() and (dlen == caps).all()
    assert_same("frame chunks", got, raw)
    bad_crc = crc_arr.copy(); bad_crc[2] ^= 1
    _g, _l, dst = run(framed, bad_crc)
    assert dst.tolist() == [0, 0, O.ERR_CRC_MISMATCH] + [0] * (nc - 3)
    corrupt = framed.copy(); corrupt[boff[0] + 100] ^= 0xFF
    _g, _l, dst = run(corrupt, crc_arr)
    assert dst[0] != 0 and (dst[1:] == 0).all()


def test_frame_decode_device_walks_the_headers_itself(codec):
    """snp_frame_decode_device: the framed stream arrives without a chunk table; a device kernel walks the headers
    (skippable / padding / repeated stream-identifier chunks, raw and compressed chunks), decodes and CRC-checks every
    chunk.  Bytes and the stream's status must equal the oracle's on good and broken streams."""
    cd = codec[O.HASH_CRC32C]
    raw = read_testdata("html") + read_testdata("fireworks.jpeg") + read_testdata("alice29.txt")
    good = O.frame_encode(raw)

    def chunk(t, body):
        return bytes([t]) + len(body).to_bytes(3, "little") + body

    # rebuild the stream chunk by chunk with extras in between
    pos, pieces = 10, [good[:10]]
    k = 0
    while pos < len(good):
        size = int.from_bytes(good[pos + 1:pos + 4], "little")
        pieces.append(good[pos:pos + 4 + size])
        if k % 3 == 0:
            pieces.append(chunk(0xfe, b"\0" * (k + 1)))               # padding
        if k % 4 == 1:
            pieces.append(chunk(0x80 + k % 0x7e, b"skip me" * k))      # reserved skippable
        if k == 2:
            pieces.append(good[:10])                                   # a second stream identifier
        pos += 4 + size
        k += 1
    spiced = b"".join(pieces)
    empty_stream = good[:10]

    def run(stream, cap=None, max_chunks=64):
        fr = to_dev(np.frombuffer(stream, dtype=np.uint8)) if len(stream) else torch.empty(0, dtype=torch.uint8, device="cuda")
        out = torch.zeros(len(raw) + 64 if cap is None else cap, dtype=torch.uint8, device="cuda")
        res = cd.frame_decode(fr, len(stream), out, max_chunks)
        torch.cuda.synchronize()
        written, status = (int(v) for v in res.cpu().tolist())
        return out[:written].cpu().numpy().tobytes(), status

    for name, stream in (("plain", good), ("with skippable chunks", spiced), ("header only", empty_stream), ("empty", b"")):
        got, st = run(stream)
        assert st == 0, name
        assert_same(name, got, O.frame_decode(stream))

    def oracle_status(stream):
        try:
            O.frame_decode(stream)
            return 0
        except O.OracleError as e:
            return e.status

    broken = []
    b = bytearray(good); b[10 + 8 + 500] ^= 0x55; broken.append(("corrupt body", bytes(b)))
    b = bytearray(good); b[10 + 4] ^= 1; broken.append(("crc flipped", bytes(b)))
    broken.append(("truncated in a chunk", good[: len(good) - 100]))
    broken.append(("truncated in a header", good[:10 + 2]))
    broken.append(("unknown unskippable chunk", good[:10] + chunk(0x05, b"abc") + good[10:]))
    b = bytearray(good); b[10 + 8] = 0xff; b[10 + 9] = 0xff; b[10 + 10] = 0xff; b[10 + 11] = 0xff; b[10 + 12] = 0x7f
    broken.append(("bad block preamble", bytes(b)))
    for name, stream in broken:
        want = oracle_status(stream)
        got, st = run(stream)
        assert want != 0 and st == want and got == b"", (name, st, want)
    # capacity / chunk-table limits
    assert run(good, cap=len(raw) - 1)[1] == O.ERR_OUTPUT_TOO_SMALL
    assert run(good, max_chunks=3)[1] == O.ERR_OUTPUT_TOO_SMALL
    got, st = run(good, cap=len(raw), max_chunks=(len(raw) + 65535) // 65536)
    assert st == 0 and got == raw


@pytest.mark.parametrize("scan", ["spans", "serial"])
def test_frame_decode_device_span_walk(scan):
    """The concurrent header walk (frame_scan.hip: 1 MiB spans, candidate entry points, resolver, emitter) against the
    oracle and against the one-lane walk on streams built to hit its corners: chunks straddling span boundaries, a
    skippable chunk larger than several spans, raw payloads full of fake chunk headers, hundreds of tiny chunks (more
    candidates than a span keeps), errors deep in the stream, a full chunk table, a short output buffer."""
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    cd.ctx.set_option(N.OPT_FRAME_SCAN, 1 if scan == "serial" else 0)
    rng = np.random.default_rng(2)

    def chunk(t, body):
        return bytes([t]) + len(body).to_bytes(3, "little") + body

    def data_chunks(raw):                      # the oracle's chunks for raw, without the stream identifier
        return O.frame_encode(raw)[10:]

    html, jpeg = read_testdata("html"), read_testdata("fireworks.jpeg")
    fake = b"".join(chunk(1, bytes(4) + bytes(60)) + chunk(0, bytes(4) + bytes([40]) + bytes(30)) for _ in range(300))
    pieces = [O.frame_encode(b"")]
    raws = []
    def add(raw):
        raws.append(raw)
        pieces.append(data_chunks(raw))
    add(html * 9)                                                   # ~0.9 MiB of compressed chunks: crosses the first span boundary
    add(jpeg * 9)                                                   # incompressible: raw (type 1) chunks
    pieces.append(chunk(0x99, rng.integers(0, 256, 2_600_000, dtype=np.uint8).tobytes()))   # skippable, larger than two spans
    add(fake[:65536] * 20)                                          # raw-looking payload that is full of plausible headers
    add(bytes(rng.integers(0, 256, 65536 * 18, dtype=np.uint8)))    # raw chunks again, random bytes
    for k in range(400):                                            # hundreds of tiny chunks: > 4 candidates per window
        add(bytes([k & 255]) * (1 + k % 7))
        if k % 50 == 0:
            pieces.append(chunk(0xfe, bytes(k)))                    # padding
    pieces.append(O.frame_encode(b""))                              # a repeated stream identifier
    add(html[:70000])
    stream = b"".join(pieces)
    raw = b"".join(raws)
    assert len(stream) > 4 * (1 << 20)
    nchunks = sum((len(r) + 65535) // 65536 for r in raws)

    def run(st, cap=None, max_chunks=None):
        fr = to_dev(np.frombuffer(st, dtype=np.uint8))
        out = torch.zeros(len(raw) + 64 if cap is None else cap, dtype=torch.uint8, device="cuda")
        res = cd.frame_decode(fr, len(st), out, nchunks + 8 if max_chunks is None else max_chunks)
        torch.cuda.synchronize()
        written, status = (int(v) for v in res.cpu().tolist())
        return out[:written].cpu().numpy().tobytes(), status

    got, st = run(stream)
    assert st == 0
    assert_same(f"span walk {scan}", got, raw)
    assert_same(f"span walk {scan} oracle", O.frame_decode(stream), raw)
    assert run(stream, max_chunks=nchunks)[1] == 0                   # exactly as many rows as chunks
    assert run(stream, max_chunks=nchunks - 1)[1] == O.ERR_OUTPUT_TOO_SMALL
    assert run(stream, max_chunks=7)[1] == O.ERR_OUTPUT_TOO_SMALL
    assert run(stream, cap=len(raw) - 1)[1] == O.ERR_OUTPUT_TOO_SMALL
    # errors far into the stream: status as the oracle, nothing returned
    cut = len(stream) - 33333
    for name, bad in (("truncated", stream[:cut]), ("truncated in header", stream[:len(stream) - len(data_chunks(html[:70000])) + 2]),
                      ("reserved chunk type", stream[:-len(data_chunks(html[:70000]))] + chunk(0x33, b"x") + data_chunks(html[:70000]))):
        try:
            O.frame_decode(bad)
            want = 0
        except O.OracleError as e:
            want = e.status
        got, st = run(bad)
        assert want != 0 and st == want and got == b"", (scan, name, st, want)
    flipped = bytearray(stream)
    flipped[len(stream) // 2] ^= 0x10                                # lands in a chunk body or header somewhere in the middle
    try:
        O.frame_decode(bytes(flipped))
        want = 0
    except O.OracleError as e:
        want = e.status
    assert run(bytes(flipped))[1] == want


# ------------------------------------------------------------------ full BASELINE size, size-independent properties

@pytest.mark.timeout(1200)
def test_full_size_config2_roundtrip(codec):
    """10 GiB of 64 KiB html-like blocks (BASELINE.json configs[1]): decode(encode(x)) == x for every block, every
    status OK, every decoded length 65536, and EVERY one of the 163 840 compressed blocks equals the oracle's (length + CRC-32C of
    the bytes; byte compare on any mismatch) -- through the default context (16-piece workspace, input register window, non-temporal
    stores: the launch bench.py times) and through a plain one-allocation workspace."""
    nb = _full_size_blocks()
    cd = codec[O.HASH_CRC32C]
    raw = SD.html_like_blocks(read_testdata("html"), 0, nb, "cuda")
    in_off, in_len = cd.uniform_layout(nb)
    out, out_off, out_len, status = cd.compress(raw, in_off, in_len)
    back = torch.empty_like(raw)
    dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len)
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0 and int((dst != 0).sum()) == 0
    assert bool((dlen == 65536).all())
    assert torch.equal(back, raw)
    assert _oracle_all(cd, raw, out, out_off, out_len, nb, O.HASH_CRC32C, "default context") == nb
    # the same 10 GiB with the pre-pass pinned (every block is offered to decompress_small.hip first, all of them come back through the list kernel)
    listc = SB.BlockCodec(0, O.HASH_CRC32C)
    layouts.set_decode_layout(listc.ctx, "small", small_max=512)
    back.zero_()
    dlen, dst = listc.decompress(out, out_off, out_len, back, in_off, in_len)
    torch.cuda.synchronize()
    assert int((dst != 0).sum()) == 0 and bool((dlen == 65536).all()) and torch.equal(back, raw)
    listc.ctx.close()
    del listc
    out2, _oo, out_len2, _st = cd.compress(raw, in_off, in_len, out=torch.empty_like(out))
    torch.cuda.synchronize()
    assert torch.equal(out_len, out_len2)
    del out2, back
    # the same batch through a context whose hash-table workspace is ONE plain allocation (no placement search, SNP_OPT_TABLE_PROBE_TRIES = 1)
    plain = SB.BlockCodec(0, O.HASH_CRC32C)
    plain.ctx.set_option(N.OPT_TABLE_PROBE_TRIES, 1)
    out.zero_()
    out, out_off, out_len, status = plain.compress(raw, in_off, in_len, out=out)
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0
    assert _oracle_all(plain, raw, out, out_off, out_len, nb, O.HASH_CRC32C, "plain workspace") == nb


def _full_size_blocks():
    """163 840 blocks (10 GiB), or a SKIP: a full-size test that quietly ran smaller would still read "full size" in the log."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info()
    nb = 163840
    need = nb * (65536 * 2 + 76512) + (3 << 30)
    if free < need:
        pytest.skip(f"full-size test needs {need >> 30} GiB of free device memory, {free >> 30} GiB are free")
    return nb


def _oracle_all(cd, raw, out, out_off, out_len, nb, variant, what, slice_blocks=16384):
    """EVERY block of the batch against the oracle (not a sample: the launch that produces bench.py's value runs only at this size):
    the device computes the CRC-32C of each block's compressed bytes (snp_crc32c_batch over out_off / out_len), the host compresses
    all nb blocks with the oracle on min(nproc, 64) threads, a slice at a time, and CRCs ITS bytes; lengths and CRCs must agree for
    every block, and any block that disagrees is compared byte for byte for the message."""
    threads = min(os.cpu_count() or 1, 64)
    d_crc = cd.crc32c(out, out_off, out_len)
    torch.cuda.synchronize()
    h_crc, h_len = d_crc.cpu().numpy().view(np.uint32), out_len.cpu().numpy().astype(np.uint32)
    compared = 0
    for s in range(0, nb, slice_blocks):
        k = min(slice_blocks, nb - s)
        blocks = raw[s * 65536:(s + k) * 65536].cpu().numpy()
        off = np.arange(k, dtype=np.uint64) * np.uint64(65536)
        ref, ref_off, ref_len, ref_st = O.compress_batch(blocks, off, np.full(k, 65536, dtype=np.uint32), variant, threads)
        ref_crc = O.crc32c_batch(ref, ref_off, ref_len, False, threads)
        bad = np.nonzero((ref_st != 0) | (ref_len != h_len[s:s + k]) | (ref_crc != h_crc[s:s + k]))[0]
        for j in bad[:4]:
            b = s + int(j)
            got = out[b * cd.comp_stride: b * cd.comp_stride + int(h_len[b])].cpu().numpy()
            want = ref[int(ref_off[j]): int(ref_off[j]) + int(ref_len[j])]
            m = min(len(got), len(want))
            d = np.nonzero(got[:m] != want[:m])[0]
            raise AssertionError(f"{what}: block {b} differs from the oracle: {len(got)} vs {len(want)} bytes, first difference at {int(d[0]) if d.size else m}"
                                 f" (device CRC {int(h_crc[b]):08x}, oracle CRC {int(ref_crc[j]):08x}); {bad.size} such blocks in this slice of {k}")
        compared += k
    assert compared == nb
    return compared


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("config", [3, 5])
def test_full_size_configs_3_and_5_roundtrip(codec, config):
    """BASELINE.json configs[2] (10 GiB low-entropy blocks) and configs[4] (one GPU's 10 GiB share of the mixed corpus) at
    full size: decode(encode(x)) == x for every block, all status OK, and every compressed block equals the oracle's (_oracle_all)."""
    nb = _full_size_blocks()
    cd = codec[O.HASH_CRC32C]
    if config == 3:
        raw = SD.low_entropy_blocks(0, nb, "cuda")
    else:
        raw = SD.corpus_blocks([read_testdata(n) for n in CORPUS], 0, nb, SD.MIXED_SEED, "cuda")
    in_off, in_len = cd.uniform_layout(nb)
    out, out_off, out_len, status = cd.compress(raw, in_off, in_len)
    back = torch.empty_like(raw)
    dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len)
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0 and int((dst != 0).sum()) == 0 and bool((dlen == 65536).all())
    assert torch.equal(back, raw)
    assert _oracle_all(cd, raw, out, out_off, out_len, nb, O.HASH_CRC32C, f"config {config}") == nb


@pytest.mark.timeout(1200)
def test_full_size_config4_framing_roundtrip(codec):
    """BASELINE.json configs[3]: a 10 GiB stream through the framing format on the device -- encode (compress + masked
    CRC-32C + chunk assembly), then decode with the device walking the headers and verifying every chunk CRC.  The
    decoded stream equals the input; the first and last chunks equal the oracle's framing byte for byte."""
    nb = _full_size_blocks()
    cd = codec[O.HASH_CRC32C]
    raw = SD.html_like_blocks(read_testdata("html"), 0, nb, "cuda")[: nb * 65536 - 4321]          # ragged last chunk
    framed, written = cd.frame_encode(raw)
    torch.cuda.synchronize()
    w = int(written.item())
    assert framed[:10].cpu().numpy().tobytes() == O.frame_encode(b"")[:10]
    first = O.frame_encode(raw[:65536].cpu().numpy().tobytes())
    assert framed[:len(first)].cpu().numpy().tobytes() == first
    last = O.frame_encode(raw[(nb - 1) * 65536:].cpu().numpy().tobytes())[10:]
    assert framed[w - len(last):w].cpu().numpy().tobytes() == last
    back = torch.empty(raw.numel(), dtype=torch.uint8, device="cuda")
    result = cd.frame_decode(framed, w, back, nb)
    torch.cuda.synchronize()
    assert result.cpu().tolist() == [raw.numel(), 0]
    assert torch.equal(back, raw)
    framed[10 + 8 + 7 + (w // 2)] ^= 0x40                                # one flipped bit somewhere in the middle of the stream
    result = cd.frame_decode(framed, w, back, nb)
    torch.cuda.synchronize()
    assert result.cpu().tolist()[1] != 0



def test_frame_decode_takes_chunks_beyond_64_kib_like_the_reference():
    """The framing format caps a chunk's uncompressed data at 65 536 bytes, but SnappyStreamDecompressor does not enforce it (it feeds whatever the
    chunk holds to the block decompressor, SnappyStreamDecompressor.cs:90-157): a foreign stream with a 200 000-byte compressed chunk, a 200 000-byte
    raw chunk or a chunk of 65 537 bytes decodes -- same bytes as the oracle through the host API and through the device-side header walk."""
    def chunk(t, body):
        return bytes([t]) + len(body).to_bytes(3, "little") + body

    data = (read_testdata("alice29.txt") + read_testdata("html"))[:200000]
    ident = O.frame_encode(b"x")[:10]
    tail = b"tail" * 10
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    for n, typ in ((200000, 0), (200000, 1), (65537, 0), (70000, 1)):
        raw = data[:n]
        body = struct.pack("<I", O.crc32c(raw, masked=True)) + (O.compress(raw) if typ == 0 else raw)
        stream = ident + chunk(typ, body) + chunk(0, struct.pack("<I", O.crc32c(tail, masked=True)) + O.compress(tail))
        ref = O.frame_decode(stream)
        assert ref == raw + tail
        assert_same(f"host frame decode, chunk of {n} ({'compressed' if typ == 0 else 'raw'})", S.frame_decode(stream), ref)
        out = torch.zeros(len(ref) + 64, dtype=torch.uint8, device="cuda")
        res = cd.frame_decode(to_dev(np.frombuffer(stream, dtype=np.uint8).copy()), len(stream), out, 16)
        torch.cuda.synchronize()
        assert res.tolist() == [len(ref), 0]
        assert_same("device header walk", out[: len(ref)].cpu().numpy().tobytes(), ref)


def test_two_threads_share_the_devices_table_pool():
    """the lane compressor's hash-table workspace belongs to the device.  Two caller threads, a context each, compress
    lane-compressor batches (22 528 fragments each, layout pinned) at the same time: the bytes equal the oracle's, and the library allocates ONE workspace (plus its bounded
    search: at most a second workspace's worth of candidates, transiently), not one per context."""
    import gc
    import threading
    gc.collect()                                                            # (contexts of earlier tests: the device's pool dies with the last of them)
    nb = 22528                                                              # 1.4 GiB of tables per launch: the searched (pieces) form
    html = read_testdata("html")
    raws = [SD.html_like_blocks(html, 1000 * t, nb, "cuda") for t in range(2)]
    cds = [SB.BlockCodec(0, O.HASH_CRC32C) for _ in range(2)]
    for cd in cds:
        cd.ctx.set_option(N.OPT_COMPRESS_LAYOUT, N.COMPRESS_LANES)          # (layout 0 hands a batch of this size to the per-wavefront kernels: < 32 768 fragments)
    outs = [torch.empty(nb * cds[0].comp_stride, dtype=torch.uint8, device="cuda") for _ in range(2)]
    out_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cds[0].comp_stride
    in_off, in_len = cds[0].uniform_layout(nb)
    cds[0].crc32c(raws[0], in_off, in_len)                                  # (first-use allocations that are not the tables)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()                                    # every torch buffer exists: what disappears from here on is the library's
    results, errors = [None, None], []
    low_water = [free0]

    def work(t):
        try:
            cd = cds[t]
            for _ in range(3):
                _o, _oo, out_len, status = cd.compress(raws[t], in_off, in_len, out=outs[t], out_off=out_off)
                low_water[0] = min(low_water[0], torch.cuda.mem_get_info()[0])
            cd.ctx.synchronize()
            results[t] = (out_len.cpu().numpy(), status.cpu().numpy())
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors
    oo = out_off.cpu().numpy()
    for t in range(2):
        out_len, status = results[t]
        assert (status == 0).all()
        raw, out = raws[t].cpu().numpy(), outs[t].cpu().numpy()
        for b in list(range(0, nb, 997)) + [nb - 1]:
            assert_same(f"thread {t} block {b}", out[oo[b]:oo[b] + out_len[b]].tobytes(),
                        O.compress(raw[b * 65536:(b + 1) * 65536].tobytes(), O.HASH_CRC32C))
    tables = nb * 65536
    used = free0 - low_water[0]
    # one workspace (+ 1/16 slack) and, while the search runs, at most a second one's worth of candidates; two private workspaces with a search
    # each would have taken more than twice that
    assert used < 2.4 * tables + (256 << 20), (used / 2**30, tables / 2**30)
    assert cds[0].ctx.counter(5) == cds[1].ctx.counter(5) and cds[0].ctx.counter(3) == cds[1].ctx.counter(3)   # the same pool's search, not one each
    assert cds[0].ctx.counter(3) <= 32, "the default search holds at most two workspaces' worth of candidates"
