"""Large single blocks through the host API (snp_try_decompress): a block of >= 256 KiB is split into 64 KiB output
fragments with the tag index (csrc/tag_index.hip) and decoded one wavefront per fragment; anything a fragment cannot
decode on its own (foreign streams whose copies cross fragments or whose tags straddle them, malformed data) must fall
back to the single-wavefront decoder and give exactly the oracle's bytes / status.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import CORPUS, read_testdata
import datagen

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import snappier_amd as S
    from snappier_amd.errors import InvalidDataException
    from snappier_amd import _native as N
    Snappy = S.Snappy


def varint(v):
    out = bytearray()
    while v >= 128:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def literal(data: bytes) -> bytes:
    k = len(data) - 1
    if k < 60:
        return bytes([k << 2]) + data
    nb = (k.bit_length() + 7) // 8
    return bytes([(59 + nb) << 2]) + k.to_bytes(nb, "little") + data


def copy2(off: int, ln: int) -> bytes:
    return bytes([2 | ((ln - 1) << 2)]) + off.to_bytes(2, "little")


def corpus_bytes(n: int) -> bytes:
    files = [read_testdata(name) for name in CORPUS]
    out = bytearray()
    i = 0
    while len(out) < n:
        out += files[i % len(files)]
        i += 1
    return bytes(out[:n])


def low_entropy_bytes(n: int) -> bytes:
    return b"".join(datagen.low_entropy_block(7 + b, 65536).tobytes() for b in range((n + 65535) // 65536))[:n]


@pytest.fixture(params=["default", "min1", "off"])
def ctx(request):
    """default: parallel path from 256 KiB; min1: every block takes the parallel path; off: never.  (An option of the PRODUCT library: the
    fragment decoder these tests assert counters 0 / 1 / 6 of is decode_chains.hip's k_decode_chains_frag.)"""
    par_min = {"default": 262144, "min1": 1, "off": 0}[request.param]
    c = S.Context(0, O.HASH_CRC32C)
    c.set_option(N.OPT_PARALLEL_DECODE_MIN, par_min)
    c.par_min = par_min
    return c


@pytest.mark.parametrize("size", [1, 14, 65536, 65537, 131072, 262144, 262145, 1000000, 4 << 20, 40 << 20])   # (40 MiB: the stream goes up in slices, indexed as it lands)
@pytest.mark.parametrize("kind", ["corpus", "low_entropy", "random"])
def test_big_block_roundtrip_matches_oracle(ctx, kind, size):
    if kind == "corpus":
        data = corpus_bytes(size)
    elif kind == "low_entropy":
        data = low_entropy_bytes(size)
    else:
        data = np.random.default_rng(size).integers(0, 256, size, dtype=np.uint8).tobytes()
    comp = Snappy.CompressToArray(data, ctx)
    assert comp == O.compress(data, O.HASH_CRC32C)
    assert Snappy.DecompressToArray(comp, ctx) == data
    par_min = ctx.par_min
    took_fragments = par_min != 0 and size >= par_min
    assert (ctx.counter(0), ctx.counter(1)) == ((1, 0) if took_fragments else (0, 0))
    # the tag index (tag_index.hip): candidate tables + one scan -- or, when a chunk's entry is no candidate, the look-back pass.  Low-entropy streams
    # never need it; a literal longer than a 16 KiB chunk of the stream (incompressible fragments) jumps over the next chunk's candidates and does.
    if took_fragments and kind == "low_entropy" and size >= 65536:   # (a stream of >= 85 % of its output's size goes to the look-back pass unasked)
        assert ctx.counter(6) == 0
    if took_fragments and kind == "corpus" and size == 1000000:       # one incompressible region (a jpeg): resolved by a pass of the scan's own
        assert ctx.counter(6) == 0
    if took_fragments and kind == "random" and size >= 131072:       # (two literals of 64 KiB: the first lands four chunks on, short of the end)
        assert ctx.counter(6) == 1


def test_foreign_streams_fall_back_and_match_oracle(ctx):
    rng = np.random.default_rng(5)
    head = rng.integers(0, 256, 65536, dtype=np.uint8).tobytes()
    # (a) copies that reach back into the previous 64 KiB fragment
    body = literal(head) + b"".join(copy2(60000 + (i % 5000), 64) for i in range(3200))
    total = 65536 + 3200 * 64
    s_a = varint(total) + body
    # (b) a literal that straddles the first fragment boundary, then in-fragment copies
    lit = rng.integers(0, 256, 65536 + 100, dtype=np.uint8).tobytes()
    s_b = varint(len(lit) + 3200 * 64) + literal(lit) + b"".join(copy2(1 + (i % 90), 64) for i in range(3200))
    # (c) a copy that straddles a fragment boundary (65536 - 32 bytes of literal, then 64-byte copies)
    lit_c = rng.integers(0, 256, 65536 - 32, dtype=np.uint8).tobytes()
    s_c = varint(len(lit_c) + 4000 * 64) + literal(lit_c) + b"".join(copy2(1000 + i, 64) for i in range(4000))
    for name, stream in (("cross-fragment copies", s_a), ("straddling literal", s_b), ("straddling copy", s_c)):
        ref = O.decompress(stream)
        before = ctx.counter(1)
        got = Snappy.DecompressToArray(stream, ctx)
        assert got == ref, name
        assert ctx.counter(1) == before + (1 if ctx.par_min else 0), name      # decoded by the fallback


def test_malformed_big_blocks_report_the_oracle_status(ctx):
    data = corpus_bytes(600000)
    comp = bytearray(O.compress(data, O.HASH_CRC32C))
    cases = []
    cases.append(("truncated", bytes(comp[: len(comp) - 1000])))
    longer = varint(len(data) + 5) + bytes(comp[len(varint(len(data))):])
    cases.append(("declared too long", longer))
    shorter = varint(len(data) - 5) + bytes(comp[len(varint(len(data))):])
    cases.append(("declared too short", shorter))
    # a copy with offset 0 spliced in after the first fragment's worth of compressed data
    bad = bytes(comp[:40000]) + copy2(0, 8) + bytes(comp[40000:])
    cases.append(("garbage in the middle", bad))
    cases.append(("bad varint", b"\xff\xff\xff\xff\xff\x01" + bytes(comp[3:])))
    for name, stream in cases:
        want = O.decompress_status(stream)
        assert want != 0, name
        with pytest.raises(InvalidDataException) as ei:
            Snappy.DecompressToArray(stream, ctx)
        assert ei.value.status == want, name


def test_big_block_output_too_small(ctx):
    data = corpus_bytes(500000)
    comp = O.compress(data, O.HASH_CRC32C)
    out = np.empty(len(data) - 1, dtype=np.uint8)
    ok, written = Snappy.TryDecompress(comp, out, ctx)
    assert not ok and written == 0
    out = np.empty(len(data) + 4096, dtype=np.uint8)
    ok, written = Snappy.TryDecompress(comp, out, ctx)
    assert ok and written == len(data) and out[:written].tobytes() == data


def test_big_block_all_decoder_variants():
    """One 3 MiB block through the fragment form of the default decoder and of the serial kernel, fenced and not: the same bytes, and the
    fragment path really ran (counter 0) -- on the product library."""
    data = corpus_bytes(3 << 20)
    comp = O.compress(data, O.HASH_CRC32C)
    for decode in ("chains", "serial"):
        for fenced in (0, 1):
            c = S.Context(0, O.HASH_CRC32C)
            c.set_option(N.OPT_DECODE_LAYOUT, N.DECODE_SERIAL if decode == "serial" else N.DECODE_AUTO)
            c.set_option(N.OPT_FENCED, fenced)
            assert Snappy.DecompressToArray(comp, c) == data, (decode, fenced)
            assert c.counter(0) == 1 and c.counter(1) == 0, (decode, fenced)


def test_contexts_are_independent_across_threads():
    """Snappy.* is re-entrant in the reference (a compressor per call, Snappy.cs:64,174,225); here: one context per
    thread (default_context is thread-local).  Eight threads hammer the host API with different data at once."""
    import threading
    files = [read_testdata(name) for name in CORPUS]
    errors = []

    def work(t):
        try:
            rng = np.random.default_rng(t)
            for i in range(12):
                data = files[(t + i) % len(files)]
                cut = int(rng.integers(1, len(data)))
                data = data[:cut] if i % 3 else data
                comp = Snappy.CompressToArray(data)                    # thread-local default context
                assert comp == O.compress(data, O.HASH_CRC32C)
                assert Snappy.DecompressToArray(comp) == data
                framed = S.snappy.frame_encode(data)
                assert S.snappy.frame_decode(framed) == data
        except BaseException as e:                                     # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_work_follows_torch_streams():
    """The batch API enqueues on torch's current stream: work issued on a side stream must be ordered with the tensors
    produced and consumed there, without a device-wide sync."""
    from snappier_amd import batch as SB, datagen as SD
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    html = read_testdata("html")
    side = torch.cuda.Stream()
    nb = 256
    with torch.cuda.stream(side):
        raw = SD.html_like_blocks(html, 5, nb, "cuda")
        in_off, in_len = cd.uniform_layout(nb)
        out, out_off, out_len, status = cd.compress(raw, in_off, in_len)
        back = torch.empty_like(raw)
        dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len)
        same = torch.equal(back, raw)                                  # consumed on the same stream
    side.synchronize()
    assert same and int((status != 0).sum()) == 0 and int((dst != 0).sum()) == 0


@pytest.mark.parametrize("layout", ["lanes", "dual"])
def test_one_context_alternating_between_two_streams_keeps_its_scratch_ordered(layout):
    """The lane compressor's hash tables -- and the dual per-wavefront form's table slots and ticket counter -- live in scratch the context
    (or its device) owns.  A context that is rebound from one torch stream to another (double buffering) must not let the second launch's
    memset + kernels run on them while the first is still using them: snp_ctx_set_stream orders the new stream behind the old one (the dual
    form's side stream is forked from and joined into whichever stream the call runs on).  All results must be the oracle's bytes."""
    from snappier_amd import batch as SB, datagen as SD
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    cd.ctx.set_option(N.OPT_COMPRESS_LAYOUT, N.COMPRESS_LANES if layout == "lanes" else N.COMPRESS_WINDOW_DUAL)
    html = read_testdata("html")
    nb = 4096
    raws = [SD.html_like_blocks(html, 100 + 7 * k, nb, "cuda") for k in range(4)]
    in_off, in_len = cd.uniform_layout(nb)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    results = []
    for k, raw in enumerate(raws):                                     # no sync between the launches
        with torch.cuda.stream(streams[k & 1]):
            results.append(cd.compress(raw, in_off, in_len))
    torch.cuda.synchronize()
    for k, (out, out_off, out_len, status) in enumerate(results):
        assert int((status != 0).sum()) == 0
        lens = out_len.cpu().numpy()
        for b in range(0, nb, 257):
            blk = raws[k][b * 65536:(b + 1) * 65536].cpu().numpy().tobytes()
            got = out[b * cd.comp_stride: b * cd.comp_stride + int(lens[b])].cpu().numpy().tobytes()
            assert got == O.compress(blk, O.HASH_CRC32C), (k, b)
    assert torch.cuda.current_device() == 0                            # entry points leave the current device alone


def test_fuzz_corrupted_big_blocks_match_the_oracle(ctx):
    """Random corruptions of large blocks through the host API: whatever path runs (fragments, fallback, single
    wavefront), status and bytes are the oracle's."""
    import test_gpu_fuzz as F
    rng = np.random.default_rng(99)
    kinds = (corpus_bytes, low_entropy_bytes)
    ok_seen = bad_seen = 0
    for i in range(int(__import__("os").environ.get("FUZZ_BIG", "36"))):
        size = int(rng.integers(200000, 1500000))
        data = kinds[i % 2](size)
        z = np.frombuffer(O.compress(data, O.HASH_CRC32C), dtype=np.uint8)
        stream = F.corrupt(rng, z).tobytes()
        want = O.decompress_status(stream)
        if want == 0:
            assert Snappy.DecompressToArray(stream, ctx) == O.decompress(stream), i
            ok_seen += 1
        else:
            cap = O.get_uncompressed_length(stream) if want not in (O.ERR_BAD_LENGTH,) else 0
            try:
                Snappy.DecompressToArray(stream, ctx)
                got = 0
            except InvalidDataException as e:
                got = e.status
            assert got == want, (i, got, want, cap)
            bad_seen += 1
    assert ok_seen and bad_seen


def test_single_block_of_int_max_bytes_matches_oracle():
    """The reference's largest span: ONE block of int.MaxValue bytes (2 GiB - 1) through snp_try_compress / snp_try_decompress -- 32 768 fragments
    behind one varint, offsets inside the block up to 2^31 - 1: bytes equal to the oracle's, and back.  One byte more is outside the reference's
    domain ((int)length goes negative, SnappyDecompressor.cs:129,160): the oracle and the library both answer "invalid stream length"."""
    import ctypes as C
    import psutil
    if psutil.virtual_memory().available < (20 << 30):
        pytest.skip("needs ~10 GiB of host memory")
    n = 0x7fffffff
    tile = np.frombuffer(corpus_bytes(24 << 20), dtype=np.uint8)
    data = np.tile(tile, n // tile.size + 1)[:n].copy()
    rng = np.random.default_rng(2031)
    idx = rng.integers(0, n, n // 300)
    data[idx] = rng.integers(0, 256, idx.size, dtype=np.uint8)             # the repeats of the tile differ
    L = S.lib()
    ctx = S.Context(0, O.HASH_CRC32C)
    assert L.snp_max_compressed_length(n) == -1                             # Snappy.GetMaxCompressedLength overflows int there (Helpers.cs:17-49) ...
    n_ref = 1840700237                                                      # ... its own largest argument: 32 + n + n / 6 + 1 + 5 == int.MaxValue
    assert L.snp_max_compressed_length(n_ref) == 0x7fffffff and L.snp_max_compressed_length(n_ref + 1) == -1
    cap = 0x7fffffff                                                        # the largest output span there is; enough for this data
    comp = np.empty(cap, dtype=np.uint8)
    w = C.c_size_t(0)
    assert L.snp_try_compress(ctx.handle, data.ctypes.data, n, comp.ctypes.data, cap, C.byref(w)) == 0
    exp = np.empty(cap, dtype=np.uint8)
    we = C.c_size_t(0)
    assert O.lib().orc_compress(data.ctypes.data, n, exp.ctypes.data, cap, O.HASH_CRC32C, C.byref(we)) == 0
    assert w.value == we.value and w.value > (1 << 29)
    assert np.array_equal(comp[: w.value], exp[: we.value])
    del exp
    ln, hdr = C.c_uint32(0), C.c_uint32(0)
    assert L.snp_get_uncompressed_length(comp.ctypes.data, w.value, C.byref(ln), C.byref(hdr)) == 0 and ln.value == n and hdr.value == 5
    back = np.zeros(n, dtype=np.uint8)
    wb = C.c_size_t(0)
    assert L.snp_try_decompress(ctx.handle, comp.ctypes.data, w.value, back.ctypes.data, n, C.byref(wb)) == 0 and wb.value == n
    assert np.array_equal(back, data)
    assert ctx.counter(0) == 1                                               # decoded by fragments (tag index), not by one wavefront
    # one byte short of room: the reference's "output too small"
    assert L.snp_try_decompress(ctx.handle, comp.ctypes.data, w.value, back.ctypes.data, n - 1, C.byref(wb)) == O.ERR_OUTPUT_TOO_SMALL
    # a declared length of 2^31: outside the reference's domain -- same answer as the oracle, whatever the room
    beyond = varint(1 << 31) + literal(b"abcdefgh")
    with pytest.raises(O.OracleError) as e:
        O.decompress(beyond, cap=16)
    assert e.value.status == O.ERR_BAD_LENGTH
    src = np.frombuffer(beyond, dtype=np.uint8).copy()
    assert L.snp_try_decompress(ctx.handle, src.ctypes.data, src.size, back.ctypes.data, n, C.byref(wb)) == O.ERR_BAD_LENGTH


def test_framed_stream_beyond_4_gib_matches_oracle():
    """A SnappyStream of 4 GiB + 197 385 bytes (65 540 chunks, the last one ragged; the corpus' jpeg makes some of them raw chunks) through
    snp_frame_encode / snp_frame_decode: every byte position of the stream and of its framing beyond 2^32 for the tail -- the framed bytes equal
    the oracle's (SnappyStreamCompressor.cs:194-261), and decode + CRC verification give the input back (SnappyStreamDecompressor.cs:38-208)."""
    import ctypes as C
    import psutil
    if psutil.virtual_memory().available < (32 << 30):
        pytest.skip("needs ~18 GiB of host memory")
    n = (4 << 30) + 3 * 65536 + 777
    tile = np.frombuffer(corpus_bytes(24 << 20), dtype=np.uint8)
    data = np.tile(tile, n // tile.size + 1)[:n].copy()
    rng = np.random.default_rng(4097)
    idx = rng.integers(0, n, n // 300)
    data[idx] = rng.integers(0, 256, idx.size, dtype=np.uint8)
    L = S.lib()
    ctx = S.Context(0, O.HASH_CRC32C)
    cap = L.snp_frame_max_encoded_length(n)
    assert cap == O.lib().orc_frame_max_encoded_length(n) and cap > n
    framed = np.empty(cap, dtype=np.uint8)
    w = C.c_size_t(0)
    assert L.snp_frame_encode(ctx.handle, data.ctypes.data, n, framed.ctypes.data, cap, C.byref(w)) == 0
    exp = np.empty(cap, dtype=np.uint8)
    we = C.c_size_t(0)
    assert O.lib().orc_frame_encode(data.ctypes.data, n, exp.ctypes.data, cap, O.HASH_CRC32C, C.byref(we)) == 0
    assert w.value == we.value
    assert np.array_equal(framed[: w.value], exp[: we.value])
    del exp
    back = np.zeros(n, dtype=np.uint8)
    wb = C.c_size_t(0)
    assert L.snp_frame_decode(ctx.handle, framed.ctypes.data, w.value, back.ctypes.data, n, C.byref(wb)) == 0 and wb.value == n
    assert np.array_equal(back, data)
    # a flipped payload byte beyond the 4 GiB mark is caught by that chunk's CRC (or its decoder) and nowhere else
    framed[w.value - 1000] ^= 0x40
    assert L.snp_frame_decode(ctx.handle, framed.ctypes.data, w.value, back.ctypes.data, n, C.byref(wb)) != 0


def test_two_threads_run_the_dual_form_at_once():
    """Two caller threads, a context each (own stream, own side stream, own table slots and ticket), compress batches of 3 000 fragments in the
    dual per-wavefront form at the same time, three times over: every sampled block equals the oracle's bytes, every status is OK."""
    import threading
    from snappier_amd import batch as SB, datagen as SD
    html = read_testdata("html")
    nb = 3000
    raws = [SD.html_like_blocks(html, 31 * t, nb, "cuda") for t in range(2)]
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def work(t):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                cd = SB.BlockCodec(0, O.HASH_CRC32C)
                cd.ctx.set_option(N.OPT_COMPRESS_LAYOUT, N.COMPRESS_WINDOW_DUAL)
                in_off, in_len = cd.uniform_layout(nb)
                for _ in range(3):
                    out, out_off, out_len, status = cd.compress(raws[t], in_off, in_len)
                st.synchronize()
                results[t] = (out.cpu().numpy(), out_len.cpu().numpy(), status.cpu().numpy(), cd.comp_stride)
                cd.ctx.close()
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors
    for t in range(2):
        out, lens, status, stride = results[t]
        assert (status == 0).all()
        h_raw = raws[t].cpu().numpy()
        for b in range(0, nb, 97):
            assert out[b * stride: b * stride + int(lens[b])].tobytes() == O.compress(h_raw[b * 65536:(b + 1) * 65536].tobytes(), O.HASH_CRC32C), (t, b)
