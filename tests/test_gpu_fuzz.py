"""Differential fuzzing of the HIP codec against the oracle, every block compared (not sampled):
  * structured random blocks (ragged lengths, repeats at random distances, runs, tiny alphabets, text) through both
    compressor layouts and both hashes -> compressed bytes identical to the oracle's;
  * the oracle's streams with random corruptions (byte flips, truncation, spliced tags, wrong preambles) through every
    decoder front end -> per-block status identical to the oracle's, and identical bytes whenever it decodes.
FUZZ_ROUNDS / FUZZ_BLOCKS scale it up (scripts/fuzz_parity.sh runs a long session on the GPU box).  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

import oracle as O
from conftest import read_testdata

pytestmark = pytest.mark.gpu

import layouts

if torch.cuda.is_available():
    import snappier_amd as S
    from snappier_amd import batch as SB, _native as N

ROUNDS = int(os.environ.get("FUZZ_ROUNDS", "2"))
BLOCKS = int(os.environ.get("FUZZ_BLOCKS", "768"))
SEED0 = int(os.environ.get("FUZZ_SEED", "0")) * 1000003             # FUZZ_SEED=k: the same tests on other inputs (a long session per k)
THREADS = min(os.cpu_count() or 1, 64)
EDGE_LENGTHS = [0, 1, 3, 4, 14, 15, 16, 17, 18, 19, 31, 32, 60, 61, 64, 65, 255, 256, 257, 4095, 4096, 16383, 16384, 16385,
                32768, 65520, 65521, 65535, 65536]


def log_session(**kw):
    """One JSON line per fuzz session (what ran, with which seeds, how many blocks): gpurun_out/fuzz_log.jsonl; the long
    session of scripts/fuzz_parity.sh is copied to profiles/ as the record of what was compared."""
    import json
    from conftest import ROOT
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fuzz_log.jsonl"), "a") as f:
        f.write(json.dumps(kw) + "\n")


def make_block(rng: np.random.Generator, text: np.ndarray) -> np.ndarray:
    n = int(rng.choice(EDGE_LENGTHS)) if rng.integers(0, 3) == 0 else int(rng.integers(0, 65537))
    kind = int(rng.integers(0, 7))
    if kind == 0:                                             # incompressible
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:                                             # text window with mutations
        s = int(rng.integers(0, len(text)))
        b = np.resize(np.roll(text, -s), n).copy()
        k = int(rng.integers(0, max(1, n // 20) + 1))
        if n and k:
            b[rng.integers(0, n, k)] = rng.integers(0, 256, k, dtype=np.uint8)
        return b
    if kind == 2:                                             # tiny alphabet: long matches, pattern copies
        return rng.integers(0, int(rng.integers(1, 4)), n, dtype=np.uint8)
    if kind == 3:                                             # runs of random length
        out = np.empty(n, dtype=np.uint8)
        pos = 0
        while pos < n:
            L = int(rng.integers(1, 1 << int(rng.integers(1, 11))))
            out[pos:pos + L] = rng.integers(0, 256)
            pos += L
        return out
    if kind == 4:                                             # random bytes with repeats copied from random distances
        out = rng.integers(0, 256, n, dtype=np.uint8)
        pos = 0
        while pos < n:
            pos += int(rng.integers(1, 200))
            if pos >= n:
                break
            dist = int(rng.integers(1, min(pos, 65535) + 1))
            L = min(int(rng.integers(4, 1 << int(rng.integers(3, 9)))), n - pos)
            for i in range(0, L, dist):                       # forward copy semantics (overlap allowed)
                out[pos + i: pos + min(i + dist, L)] = out[pos + i - dist: pos - dist + min(i + dist, L)]
            pos += L
        return out
    if kind == 5:                                             # periodic pattern with a defect now and then
        P = int(rng.integers(1, 70))
        b = np.resize(rng.integers(0, 256, P, dtype=np.uint8), n).copy()
        k = int(rng.integers(0, 6))
        if n and k:
            b[rng.integers(0, n, k)] ^= 0xFF
        return b
    a, c = make_block(rng, text), make_block(rng, text)       # two halves of different kinds
    return np.concatenate([a[: len(a) // 2], c[: len(c) // 2]])[:65536]


def batch_of(blocks):
    lens = np.array([len(b) for b in blocks], dtype=np.int32)
    off = np.zeros(len(blocks), dtype=np.int64)
    off[1:] = np.cumsum(lens[:-1])
    data = np.concatenate(blocks + [np.zeros(64, dtype=np.uint8)])
    return data, off, lens


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("layout", ["win", "win2", "wing", "wind", "lanes"])
@pytest.mark.parametrize("variant", [O.HASH_CRC32C, O.HASH_MUL])
def test_fuzz_compress_bytes_equal_oracle(layout, variant):
    text = np.frombuffer(read_testdata("html") + read_testdata("alice29.txt"), dtype=np.uint8)
    cd = SB.BlockCodec(0, variant)
    layouts.set_compress_layout(cd.ctx, layout)
    for r in range(ROUNDS):
        rng = np.random.default_rng(SEED0 + 1000 * r + 17 * variant + (layout == "lanes"))
        blocks = [make_block(rng, text) for _ in range(BLOCKS)]
        data, off, lens = batch_of(blocks)
        ref, ref_off, ref_len, ref_st = O.compress_batch(data, off.astype(np.uint64), lens.astype(np.uint32), variant, THREADS)
        out, out_off, out_len, status = cd.compress(dev(data), dev(off), dev(lens))
        torch.cuda.synchronize()
        out, out_off, out_len = out.cpu().numpy(), out_off.cpu().numpy(), out_len.cpu().numpy()
        assert (status.cpu().numpy() == 0).all() and (ref_st == 0).all()
        assert (out_len == ref_len).all(), f"round {r}: lengths differ at blocks {np.nonzero(out_len != ref_len)[0][:8]}"
        for b in range(BLOCKS):
            got = out[out_off[b]: out_off[b] + out_len[b]]
            want = ref[int(ref_off[b]): int(ref_off[b]) + int(ref_len[b])]
            assert np.array_equal(got, want), f"round {r} block {b} (len {lens[b]}) {layout} v{variant}"
    log_session(test="compress_bytes_equal_oracle", layout=layout, hash_variant=variant, rounds=ROUNDS, blocks_per_round=BLOCKS,
                blocks_compared=ROUNDS * BLOCKS, seeds=[SEED0 + 1000 * r + 17 * variant + (layout == "lanes") for r in range(ROUNDS)], result="all equal")


def corrupt(rng: np.random.Generator, z: np.ndarray) -> np.ndarray:
    z = z.copy()
    how = int(rng.integers(0, 7))
    n = len(z)
    if how == 0 or n < 8:
        return z                                              # untouched
    if how == 1:                                              # flip a few bytes anywhere
        k = int(rng.integers(1, 4))
        z[rng.integers(0, n, k)] ^= rng.integers(1, 256, k, dtype=np.uint8)
        return z
    if how == 2:                                              # truncate
        return z[: int(rng.integers(0, n))]
    if how == 3:                                              # flip inside the preamble / first tags
        z[int(rng.integers(0, min(n, 6)))] ^= int(rng.integers(1, 256))
        return z
    if how == 4:                                              # splice a random tag somewhere
        p = int(rng.integers(0, n))
        tag = rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8)
        return np.concatenate([z[:p], tag, z[p:]])
    if how == 5:                                              # append garbage
        return np.concatenate([z, rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)])
    p = int(rng.integers(0, n))                               # zero a short range (offset 0 copies, long literals)
    z[p: p + int(rng.integers(1, 9))] = 0
    return z


@pytest.mark.parametrize("decode", layouts.DECODE_LAYOUTS)
def test_fuzz_corrupted_streams_status_and_bytes_equal_oracle(decode):
    text = np.frombuffer(read_testdata("html") + read_testdata("alice29.txt"), dtype=np.uint8)
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    layouts.set_decode_layout(cd.ctx, decode)   # (small*: every block first goes through the small-block pre-pass -- decompress_small.hip: 8 lanes per block, or the named layout)
    for r in range(ROUNDS):
        rng = np.random.default_rng(SEED0 + 777 + r)
        blocks = [make_block(rng, text) for _ in range(BLOCKS)]
        data, off, lens = batch_of(blocks)
        comp, c_off, c_len, _ = O.compress_batch(data, off.astype(np.uint64), lens.astype(np.uint32), O.HASH_CRC32C, THREADS)
        streams = [corrupt(rng, comp[int(c_off[b]): int(c_off[b]) + int(c_len[b])]) for b in range(BLOCKS)]
        sdata, s_off, s_len = batch_of(streams)
        caps = np.array([len(b) if rng.integers(0, 8) else max(0, len(b) - int(rng.integers(0, 3))) for b in blocks], dtype=np.int32)
        caps = np.maximum(caps, 1)                            # a zero capacity is an argument error at the boundary
        out_off = np.zeros(BLOCKS, dtype=np.int64)
        out_off[1:] = np.cumsum(caps[:-1].astype(np.int64) + 64)
        total = int(out_off[-1]) + int(caps[-1]) + 64
        ref, ref_len, ref_st = O.decompress_batch(sdata, s_off.astype(np.uint64), s_len.astype(np.uint32), out_off.astype(np.uint64),
                                                  caps.astype(np.uint32), total, THREADS)
        out = torch.zeros(total, dtype=torch.uint8, device="cuda")
        dlen, dst = cd.decompress(dev(sdata), dev(s_off), dev(s_len), out, dev(out_off), dev(caps))
        torch.cuda.synchronize()
        dlen, dst, out = dlen.cpu().numpy(), dst.cpu().numpy(), out.cpu().numpy()
        bad = np.nonzero(dst != ref_st)[0]
        assert bad.size == 0, f"round {r} {decode}: status differs at blocks {bad[:8]}: got {dst[bad[:8]]} want {ref_st[bad[:8]]}"
        ok = np.nonzero(ref_st == 0)[0]
        assert (dlen[ok] == ref_len[ok]).all()
        for b in ok:
            assert np.array_equal(out[out_off[b]: out_off[b] + dlen[b]], ref[out_off[b]: out_off[b] + ref_len[b]]), f"round {r} block {b}"
        assert ok.size > BLOCKS // 10 and ok.size < BLOCKS    # the corruptions produce both outcomes
    log_session(test="corrupted_streams_status_and_bytes_equal_oracle", decode=decode, rounds=ROUNDS, blocks_per_round=BLOCKS,
                blocks_compared=ROUNDS * BLOCKS, seeds=[SEED0 + 777 + r for r in range(ROUNDS)], result="all statuses and bytes equal")


@pytest.mark.parametrize("layout", ["lanes", "team4", "team8", "team16"])
def test_fuzz_small_blocks_corrupted_streams_equal_oracle(layout):
    """The small-block pre-pass (decompress_small.hip) on what it is for: thousands of blocks of 1 .. 512 bytes, two thirds of them
    corrupted, through every layout (a lane, or a team of 4 / 8 / 16 lanes, per block); leftovers go to the list kernel.  Status,
    length and bytes of every block must equal the oracle's."""
    text = np.frombuffer(read_testdata("html") + read_testdata("alice29.txt") + read_testdata("geo.protodata"), dtype=np.uint8)
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    layouts.set_decode_layout(cd.ctx, "small-" + layout, small_max=512)
    nb = 4 * BLOCKS
    for r in range(ROUNDS):
        rng = np.random.default_rng(SEED0 + 4242 + r)
        blocks = []
        for _ in range(nb):
            n = int(rng.integers(1, 513))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                blocks.append(rng.integers(0, 256, n, dtype=np.uint8))
            elif kind == 1:                                    # a short pattern repeated: overlapping copies with small offsets
                p = rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)
                blocks.append(np.tile(p, n // len(p) + 1)[:n])
            else:
                s = int(rng.integers(0, len(text) - n))
                blocks.append(text[s: s + n].copy())
        data, off, lens = batch_of(blocks)
        comp, c_off, c_len, _ = O.compress_batch(data, off.astype(np.uint64), lens.astype(np.uint32), O.HASH_CRC32C, THREADS)
        streams = []
        for b in range(nb):
            z = comp[int(c_off[b]): int(c_off[b]) + int(c_len[b])]
            streams.append(corrupt(rng, z) if rng.integers(0, 3) else z.copy())
        sdata, s_off, s_len = batch_of(streams)
        caps = np.array([len(b) if rng.integers(0, 8) else max(1, len(b) - int(rng.integers(0, 3))) for b in blocks], dtype=np.int32)
        out_off = np.zeros(nb, dtype=np.int64)
        out_off[1:] = np.cumsum(caps[:-1].astype(np.int64) + 64)
        total = int(out_off[-1]) + int(caps[-1]) + 64
        ref, ref_len, ref_st = O.decompress_batch(sdata, s_off.astype(np.uint64), s_len.astype(np.uint32), out_off.astype(np.uint64),
                                                  caps.astype(np.uint32), total, THREADS)
        out = torch.zeros(total, dtype=torch.uint8, device="cuda")
        dlen, dst = cd.decompress(dev(sdata), dev(s_off), dev(s_len), out, dev(out_off), dev(caps))
        torch.cuda.synchronize()
        dlen, dst, out = dlen.cpu().numpy(), dst.cpu().numpy(), out.cpu().numpy()
        bad = np.nonzero(dst != ref_st)[0]
        assert bad.size == 0, f"round {r} {layout}: status differs at blocks {bad[:8]}: got {dst[bad[:8]]} want {ref_st[bad[:8]]}"
        ok = np.nonzero(ref_st == 0)[0]
        assert (dlen[ok] == ref_len[ok]).all()
        idx = np.concatenate([np.arange(out_off[b], out_off[b] + ref_len[b]) for b in ok])
        assert np.array_equal(out[idx], ref[idx]), f"round {r} {layout}: bytes differ"
        assert ok.size > nb // 10 and ok.size < nb
    log_session(test="small_blocks_corrupted_streams_equal_oracle", layout=layout, rounds=ROUNDS, blocks_per_round=nb,
                blocks_compared=ROUNDS * nb, seeds=[SEED0 + 4242 + r for r in range(ROUNDS)], result="all statuses, lengths and bytes equal")


TAXONOMY = [
    bytes([5, 0x00]), bytes([4, 0x0C, 97, 98, 99, 100]), bytes([4, 0x10, 97, 98, 99, 100, 101]),
    bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x00]), bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x05]),
    bytes([8, 0x0C, 97, 98, 99, 100, 0x01, 0x04]), bytes([8, 0x00, 97, 0x1A, 0x01, 0x00]),
    bytes([6, 0x00, 97, 0x13, 0x01, 0x00, 0x00, 0x00]), bytes([8, 0x0C, 97, 98, 99, 100, 0x02]),
    bytes([0x80]), bytes([0xFF] * 6), bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x7F, 0]), bytes([0]),
    bytes([3, 0xFC, 0xFF, 0xFF, 0xFF, 0xFF, 1, 2, 3]),            # 4-byte literal length 2^32, the input ends inside it
    bytes([3, 0xFC, 0xFF, 0xFF, 0xFF, 0xFF, 0x08, 97, 98, 99]),   # the same length followed by a well-formed literal: skipping the 2^32 tag would "succeed" 
    bytes([3, 0xFC, 0xFF, 0xFF, 0xFF, 0xFF, 0x08, 97, 98, 99] + [0] * 16),
    bytes([70, 0x00, 97]) + bytes([0xFE, 0x01, 0x00]) + bytes([0x12, 0x01, 0x00]),   # 64-byte + 5-byte pattern copies
    bytes([3, 0x08, 97, 98, 99]), bytes([12, 0x08, 97, 98, 99, 0x15, 0x03, 0x01, 0x03]),
]


@pytest.mark.parametrize("layout", ["lanes", "team4", "team8", "team16"])
def test_error_taxonomy_through_every_small_block_layout(layout):
    """The decoder's status vectors (test_decoder_error_taxonomy_matches_oracle) as a BATCH, so that the small-block pre-pass takes
    them, through every layout: status, length and bytes equal the oracle's whichever layout the policy would have picked, and
    nothing is written at or beyond out_len when the capacity is larger than the declared length."""
    small_layout = "small-" + layout
    rng = np.random.default_rng(99)
    text = np.frombuffer(read_testdata("html"), dtype=np.uint8)
    streams, caps = [], []
    for rep in range(40):                                           # 40 x (vectors + clean blocks), shuffled positions inside the wavefronts
        for v in TAXONOMY:
            streams.append(np.frombuffer(v, dtype=np.uint8).copy())
            caps.append(128)
        for _ in range(24):
            n = int(rng.integers(1, 300))
            s = int(rng.integers(0, len(text) - n))
            streams.append(np.frombuffer(O.compress(text[s: s + n].tobytes()), dtype=np.uint8).copy())
            caps.append(n + int(rng.integers(0, 40)))               # capacity >= declared length: the slack must stay untouched
    order = rng.permutation(len(streams))
    streams = [streams[i] for i in order]
    caps = np.array([caps[i] for i in order], dtype=np.int32)
    nb = len(streams)
    sdata, s_off, s_len = batch_of(streams)
    out_off = np.zeros(nb, dtype=np.int64)
    out_off[1:] = np.cumsum(caps[:-1].astype(np.int64) + 64)
    total = int(out_off[-1]) + int(caps[-1]) + 64
    ref, ref_len, ref_st = O.decompress_batch(sdata, s_off.astype(np.uint64), s_len.astype(np.uint32), out_off.astype(np.uint64),
                                              caps.astype(np.uint32), total, THREADS)
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    layouts.set_decode_layout(cd.ctx, small_layout, small_max=512)
    for call in range(2):                                           # the second call runs under the policy the first one taught the context
        out = torch.full((total,), 0xA5, dtype=torch.uint8, device="cuda")
        dlen, dst = cd.decompress(dev(sdata), dev(s_off), dev(s_len), out, dev(out_off), dev(caps))
        torch.cuda.synchronize()
        dlen, dst, outh = dlen.cpu().numpy(), dst.cpu().numpy(), out.cpu().numpy()
        bad = np.nonzero(dst != ref_st)[0]
        assert bad.size == 0, f"{layout} call {call}: status differs at {bad[:8]}: got {dst[bad[:8]]} want {ref_st[bad[:8]]}, streams {[streams[i].tobytes().hex() for i in bad[:3]]}"
        ok = np.nonzero(ref_st == 0)[0]
        assert (dlen[ok] == ref_len[ok]).all()
        for b in ok:
            o, l, c = int(out_off[b]), int(ref_len[b]), int(caps[b])
            assert np.array_equal(outh[o: o + l], ref[o: o + l]), f"{layout}: block {b} bytes"
            assert (outh[o + l: o + c + 64] == 0xA5).all(), f"{layout}: block {b} wrote beyond its {l} bytes (cap {c})"


def test_context_options_pin_layouts_without_changing_results():
    """snp_ctx_set_option (include/snappier_hip.h): decode layout, small-block thresholds, compress layout, probe cap -- set through
    the C-ABI between calls on ONE context, alternating block sizes (the workload whose policy would otherwise come from the previous
    batch): every setting returns the oracle's bytes; bad values are rejected and change nothing."""
    N = S._native
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    ctx = cd.ctx
    assert ctx.get_option(N.OPT_DECODE_LAYOUT) == 0 and ctx.get_option(N.OPT_SMALL_BLOCK_MAX) == 512
    with pytest.raises(ValueError):
        ctx.set_option(N.OPT_DECODE_LAYOUT, 9)
    with pytest.raises(ValueError):
        ctx.set_option(12345, 0)
    assert ctx.get_option(N.OPT_DECODE_LAYOUT) == 0
    ctx.set_option(N.OPT_TABLE_PROBE_MAX_BYTES, 1 << 30)
    ctx.set_option(N.OPT_SMALL_BLOCK_MIN_BATCH, 1)
    html = read_testdata("html")
    rng = np.random.default_rng(5)
    small = [np.frombuffer(html[s: s + n], dtype=np.uint8) for s, n in ((int(rng.integers(0, 90000)), int(rng.integers(1, 400))) for _ in range(3000))]
    large = [np.frombuffer(html[s: s + 65536], dtype=np.uint8) for s in (0, 4099, 30000)] * 8
    for layout, blocks in [(2, small), (1, large), (4, small), (0, large), (3, small), (5, small), (1, small), (0, small)]:
        ctx.set_option(N.OPT_DECODE_LAYOUT, layout)
        assert ctx.get_option(N.OPT_DECODE_LAYOUT) == layout
        data, off, lens = batch_of(blocks)
        for comp_layout in (0, 2, 3):
            ctx.set_option(N.OPT_COMPRESS_LAYOUT, comp_layout)
            out, out_off, out_len, st = cd.compress(dev(data), dev(off), dev(lens))
            torch.cuda.synchronize()
            assert int((st != 0).sum()) == 0
            ol, oo, oh = out_len.cpu().numpy(), out_off.cpu().numpy(), out.cpu().numpy()
            for b in range(0, len(blocks), max(1, len(blocks) // 40)):
                assert oh[oo[b]: oo[b] + ol[b]].tobytes() == O.compress(blocks[b].tobytes()), (layout, comp_layout, b)
        back = torch.zeros(int(lens.sum()) + 64, dtype=torch.uint8, device="cuda")
        dlen, dst = cd.decompress(out, out_off, out_len, back, dev(off), dev(lens))
        torch.cuda.synchronize()
        assert int((dst != 0).sum()) == 0 and np.array_equal(dlen.cpu().numpy(), lens)
        assert np.array_equal(back.cpu().numpy()[: data.size], data), layout
    ctx.set_option(N.OPT_COMPRESS_LAYOUT, 0)
    ctx.set_option(N.OPT_DECODE_LAYOUT, 0)


def test_crc32c_table_free_kernel_equals_the_table_kernel_and_the_oracle():
    """SNP_OPT_CRC_KERNEL = 1 (the kernel BASELINE.json's north star names: no table, the GF(2) shift map bit by bit), 0 (the default: three
    LDS tables of 11 + 11 + 10 bits, four byte ranges per wavefront -- so the ragged list also ends inside a wavefront's group of four) and 2 (round 3's
    four 8-bit tables) against the oracle (Crc32CAlgorithm.cs:41-158): ragged lengths 0 .. 70 000, masked and unmasked, and through the
    framing format (snp_frame_encode computes every chunk's CRC with the selected kernel, snp_frame_decode verifies with it)."""
    N = S._native
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    rng = np.random.default_rng(11)
    lens = [0, 1, 2, 3, 4, 5, 15, 16, 17, 1023, 1024, 1025, 4095, 4096, 65535, 65536, 70000] + [int(x) for x in rng.integers(0, 70000, 201)]   # 218 ranges: not a multiple of 16
    blocks = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    data, off, ln = batch_of(blocks)
    want = {m: np.array([O.crc32c(b.tobytes(), masked=m) for b in blocks], dtype=np.uint32) for m in (False, True)}
    for table_free in (1, 0, 2, 1):
        cd.ctx.set_option(N.OPT_CRC_KERNEL, table_free)
        assert cd.ctx.get_option(N.OPT_CRC_KERNEL) == table_free
        for m in (False, True):
            got = cd.crc32c(dev(data), dev(off), dev(ln), masked=m).cpu().numpy().astype(np.uint32)
            assert np.array_equal(got, want[m]), (table_free, m, np.nonzero(got != want[m])[0][:5])
        payload = read_testdata("html") * 2
        framed = S.frame_encode(payload, cd.ctx)
        assert framed == O.frame_encode(payload) and S.frame_decode(framed, cd.ctx) == payload
    cd.ctx.set_option(N.OPT_CRC_KERNEL, 0)


def test_hash_table_workspace_in_pieces_gives_the_same_bytes():
    """A lane-compressor batch whose hash-table workspace is >= 1 GiB runs on up to 16 separately allocated pieces picked by the placement
    search (capi_pool.hip, TablePool / PieceSearch) -- the DEVICE's workspace, which every context borrows; with SNP_OPT_TABLE_PROBE_TRIES = 1 on a
    plain allocation of the context's own; when the byte cap leaves no room for spare candidates, on unsearched pieces.  All of them must
    return the oracle's bytes: every block compared by length and CRC across the forms, the blocks either side of every piece boundary (and
    the ragged tail) byte for byte against the oracle; a second, larger batch makes the pool grow (pieces freed, bounded search repeated)."""
    N = S._native
    import gc
    from snappier_amd import datagen as SD
    from snappier_amd import context as SC
    SC.close_default_contexts()
    gc.collect()                                                       # (contexts of earlier tests: the device's pool dies with the last of them)
    html = read_testdata("html")
    sizes = (20001, 36000)                                             # 1.3 GB and 2.4 GB of tables: 16 pieces of 1344 / 2432 fragments

    def inputs(nb):
        raw = SD.html_like_blocks(html, 7, nb, "cuda")
        lens = torch.full((nb,), 65536, dtype=torch.int32, device="cuda")
        lens[-1] = 777
        lens[nb // 2] = 0
        off = torch.arange(nb, dtype=torch.int64, device="cuda") * 65536
        return raw, off, lens

    def signature(cd, raw, off, lens):
        out, out_off, out_len, st = cd.compress(raw, off, lens)
        torch.cuda.synchronize()
        assert int((st != 0).sum()) == 0
        crcs = cd.crc32c(out, out_off, out_len)
        return out, out_off, out_len, (out_len.cpu().numpy(), crcs.cpu().numpy())

    # 1. a pool built under a byte cap that leaves no room for spare candidates (no other context alive: the pool is this context's to build)
    capped = {}
    for nb in sizes:
        raw, off, lens = inputs(nb)
        cd = SB.BlockCodec(0, O.HASH_CRC32C)
        cd.ctx.set_option(N.OPT_TABLE_PROBE_MAX_BYTES, 1 << 30)
        cd.ctx.set_option(N.OPT_COMPRESS_LAYOUT, 2)
        *_x, capped[nb] = signature(cd, raw, off, lens)
        assert cd.ctx.counter(3) in (0, 16), "capped: nothing to search"
        cd.ctx.close()                                                  # the device's last context: the pool goes with it
        del raw
    # 2. the searched pool (default budget: two workspaces' worth of candidates), a second context that borrows it, and a private plain workspace
    searched, borrower, plain = SB.BlockCodec(0, O.HASH_CRC32C), SB.BlockCodec(0, O.HASH_CRC32C), SB.BlockCodec(0, O.HASH_CRC32C)
    plain.ctx.set_option(N.OPT_TABLE_PROBE_TRIES, 1)
    for cd in (searched, borrower, plain):
        cd.ctx.set_option(N.OPT_COMPRESS_LAYOUT, 2)
    for nb in sizes:
        raw, off, lens = inputs(nb)
        if nb == sizes[0]:                                              # snp_ctx_reserve_compress: the search runs now, the call below finds the pool built
            searched.ctx.reserve_compress(nb)
            at_reserve = searched.ctx.counter(3)
            assert 16 < at_reserve <= 32 and searched.ctx.counter(2) > 0, f"the default search holds at most two workspaces' worth of candidates: {at_reserve}"
            piece0 = ((nb + nb // 16 + 15) // 16 + 63) // 64 * 64
            assert searched.ctx.counter(4) > 0 and searched.ctx.counter(5) == at_reserve * piece0 * 65536   # search wall time (us), bytes held at once
        out, out_off, out_len, sig = signature(searched, raw, off, lens)
        if nb == sizes[0]:
            assert searched.ctx.counter(3) == at_reserve, "the compress call searched again after snp_ctx_reserve_compress"
        assert 16 < searched.ctx.counter(3) <= 32 and searched.ctx.counter(2) > 0, "the search did not run (or went beyond its default budget)"
        piece = ((nb + nb // 16 + 15) // 16 + 63) // 64 * 64            # capacity = the batch + 1/16 of slack (capi_pool.hip, build_tables)
        check = sorted({b for k in range(1, 16) for b in (k * piece - 1, k * piece) if b < nb} | {0, nb // 2, nb - 2, nb - 1})
        idx = torch.tensor(check, device="cuda")
        sub = torch.cat([raw[int(off[b]): int(off[b]) + int(lens[b])] for b in check]).cpu().numpy()
        sub_len = lens[idx].cpu().numpy().astype(np.uint32)
        sub_off = np.concatenate([[0], np.cumsum(sub_len[:-1], dtype=np.uint64)]).astype(np.uint64)
        ref, ref_off, ref_len, ref_st = O.compress_batch(sub, sub_off, sub_len, O.HASH_CRC32C, THREADS)
        outh_len = out_len[idx].cpu().numpy()
        assert (ref_st == 0).all() and (outh_len == ref_len).all()
        for j, b in enumerate(check):
            got = out[int(out_off[b]): int(out_off[b]) + int(outh_len[j])].cpu().numpy()
            assert np.array_equal(got, ref[int(ref_off[j]): int(ref_off[j]) + int(ref_len[j])]), f"block {b} (piece boundary) of {nb}"
        *_x, sig_b = signature(borrower, raw, off, lens)
        assert borrower.ctx.counter(3) == searched.ctx.counter(3) and borrower.ctx.counter(5) == searched.ctx.counter(5), "the second context built a workspace of its own"
        *_x, sig_p = signature(plain, raw, off, lens)
        assert plain.ctx.counter(3) == 0, "SNP_OPT_TABLE_PROBE_TRIES = 1 searched"
        for name, (l, c) in (("borrower", sig_b), ("one allocation", sig_p), ("capped", capped[nb])):
            assert (l == sig[0]).all() and (c == sig[1]).all(), f"{name} differs from the searched workspace at {nb} blocks"
        del raw
    log_session(test="hash_table_workspace_in_pieces", blocks=list(sizes), contexts=["capped", "searched", "borrower", "one_allocation"], result="all equal")


@pytest.mark.parametrize("variant", [O.HASH_CRC32C, O.HASH_MUL])
def test_small_fragment_launch_with_input_in_lds_equals_oracle(variant):
    """Batches whose longest fragment is small take a second form of the lane compressor (fragment bytes in LDS, compress_lanes.hip
    SMALL) -- chosen from the PREVIOUS launch's longest fragment, verified on the device against this launch's.  Sequences that
    exercise every combination: no hint yet (general launch), hint and batch agree (LDS launch), a batch with a longer fragment after a
    small hint (the LDS launch must step aside), a smaller batch after a larger hint, forced slot sizes; ragged lengths 0..limit incl.
    the 14/15/16-byte boundary of the reference's short-input path.  Every block byte-equal to the oracle."""
    N = S._native
    text = np.frombuffer(read_testdata("html") + read_testdata("alice29.txt"), dtype=np.uint8)
    rng = np.random.default_rng(77 + variant)
    cd = SB.BlockCodec(0, variant)
    cd.ctx.set_option(N.OPT_COMPRESS_LAYOUT, 2)                       # the lane compressor whatever the batch size

    def batch(limit, count, exact=False):
        blocks = []
        for _ in range(count):
            n = limit if exact else int(rng.choice([0, 1, 3, 14, 15, 16, 17, 31, limit - 1, limit])) if rng.integers(0, 4) == 0 else int(rng.integers(0, limit + 1))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                b = rng.integers(0, 256, n, dtype=np.uint8)
            elif kind == 1:
                b = rng.integers(0, int(rng.integers(1, 4)), n, dtype=np.uint8)
            else:
                s0 = int(rng.integers(0, len(text) - n - 1))
                b = text[s0: s0 + n].copy()
                if n and kind == 3:
                    b[n // 2:] = b[: n - n // 2]                       # a long match inside the fragment
            blocks.append(b)
        blocks[int(rng.integers(0, count))] = np.resize(text[100: 100 + limit], limit).copy()   # the longest fragment is `limit` for sure
        return blocks

    def check(blocks, what):
        data, off, lens = batch_of(blocks)
        ref, ref_off, ref_len, ref_st = O.compress_batch(data, off.astype(np.uint64), lens.astype(np.uint32), variant, THREADS)
        out, out_off, out_len, status = cd.compress(dev(data), dev(off), dev(lens))
        torch.cuda.synchronize()
        out, out_off, out_len = out.cpu().numpy(), out_off.cpu().numpy(), out_len.cpu().numpy()
        assert (status.cpu().numpy() == 0).all() and (ref_st == 0).all(), what
        assert (out_len == ref_len).all(), f"{what}: lengths differ at blocks {np.nonzero(out_len != ref_len)[0][:8]}"
        for b in range(len(blocks)):
            assert np.array_equal(out[out_off[b]: out_off[b] + out_len[b]], ref[int(ref_off[b]): int(ref_off[b]) + int(ref_len[b])]), f"{what}: block {b} (len {lens[b]})"
        return len(blocks)

    compared = 0
    for step, limit in enumerate([256, 256, 200, 512, 512, 4096, 96, 96, 768, 768, 1024, 300, 300, 65536, 256]):
        compared += check(batch(limit, 3000 if limit <= 4096 else 64), f"step {step}: fragments up to {limit} bytes")
    for forced in (0, 256, 128, 2000):                                   # SNP_OPT_COMPRESS_SMALL_INPUT_LDS: never / slot sizes that fit, do not fit, exceed what LDS holds at 32 lanes
        cd.ctx.set_option(N.OPT_COMPRESS_SMALL_INPUT_LDS, forced)
        for limit in (256, 130):
            compared += check(batch(limit, 2000), f"SNP_OPT_COMPRESS_SMALL_INPUT_LDS={forced}, fragments up to {limit}")
    cd.ctx.set_option(N.OPT_COMPRESS_SMALL_INPUT_LDS, -1)
    for per in (16, 64):
        cd.ctx.set_option(N.OPT_COMPRESS_SMALL_INPUT_LANES, per)
        for limit in (256, 256):
            compared += check(batch(limit, 2000, exact=(per == 64)), f"{per} lanes per wavefront")
    log_session(test="small_fragment_launch_with_input_in_lds", hash_variant=variant, blocks_compared=compared, result="all equal")


def _compare_batch(cd, data, off, lens, variant, what):
    """Compress (tight output layout: every block gets exactly snp_max_compressed_length of its own length), compare every block with the oracle,
    decompress the device's bytes back and compare with the input.  -> blocks compared."""
    nb = len(lens)
    caps = 32 + lens.astype(np.int64) + lens.astype(np.int64) // 6 + 1 + 5          # Snappy.GetMaxCompressedLength  Snappy.cs:20-24
    c_off = np.zeros(nb, dtype=np.int64)
    c_off[1:] = np.cumsum(caps[:-1])
    out = torch.empty(int(caps.sum()) + 64, dtype=torch.uint8, device="cuda")
    ref, ref_off, ref_len, ref_st = O.compress_batch(data, off.astype(np.uint64), lens.astype(np.uint32), variant, THREADS)
    d_data, d_off, d_lens, d_coff = dev(data), dev(off), dev(lens), dev(c_off)
    _o, _oo, out_len, status = cd.compress(d_data, d_off, d_lens, out=out, out_off=d_coff)
    torch.cuda.synchronize()
    assert int((status != 0).sum()) == 0 and (ref_st == 0).all(), what
    h_len = out_len.cpu().numpy()
    assert (h_len == ref_len).all(), f"{what}: lengths differ at blocks {np.nonzero(h_len != ref_len)[0][:8]}"
    h_out = out.cpu().numpy()
    ro = ref_off.astype(np.int64)
    for b in range(nb):
        if not np.array_equal(h_out[c_off[b]: c_off[b] + h_len[b]], ref[ro[b]: ro[b] + h_len[b]]):
            raise AssertionError(f"{what}: block {b} (len {lens[b]}) differs from the oracle")
    back = torch.zeros(data.size, dtype=torch.uint8, device="cuda")
    dlen, dst = cd.decompress(out, d_coff, out_len, back, d_off, d_lens)
    torch.cuda.synchronize()
    assert int((dst != 0).sum()) == 0 and bool((dlen == d_lens).all()), what
    total = int(off[-1] + lens[-1])
    assert bool(torch.equal(back[:total], d_data[:total])), what
    return nb


def test_batches_beyond_one_launch_slice_equal_oracle():
    """A lane-compressor batch of more than 262 144 fragments runs as several launches over one hash-table workspace (capi_batch.hip, slice_fragments), and
    the decoder's small-block pre-pass sees more blocks than any other test gives it: 600 000 blocks of 0..300 bytes in ONE call, every block equal to
    the oracle and back.  Then the same seam at full fragment size: SNP_OPT_COMPRESS_SLICE = 4096 cuts 9 000 mixed fragments (up to 64 KiB) into three launches."""
    text = np.frombuffer(read_testdata("html") + read_testdata("alice29.txt") + read_testdata("geo.protodata"), dtype=np.uint8)
    rng = np.random.default_rng(262144)
    nb = 600000
    lens = rng.integers(0, 301, nb).astype(np.int32)
    lens[rng.integers(0, nb, 2000)] = rng.choice(np.array([0, 1, 14, 15, 16, 17, 300], dtype=np.int32), 2000)
    off = np.zeros(nb, dtype=np.int64)
    off[1:] = np.cumsum(lens[:-1].astype(np.int64))
    total = int(off[-1] + lens[-1])
    data = np.resize(text, total + 64).copy()
    k = total // 40
    data[rng.integers(0, total, k)] = rng.integers(0, 256, k, dtype=np.uint8)        # no two windows of the text alike
    compared = 0
    for variant in (O.HASH_CRC32C, O.HASH_MUL):
        cd = SB.BlockCodec(0, variant)
        compared += _compare_batch(cd, data, off, lens, variant, f"600 000 small blocks in one call, hash {variant}")
        assert cd.ctx.counter(2) >= 0
    rng = np.random.default_rng(4096)
    blocks = [make_block(rng, text) for _ in range(9000)]
    d2, o2, l2 = batch_of(blocks)
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    cd.ctx.set_option(N.OPT_COMPRESS_SLICE, 4096)
    cd.ctx.set_option(N.OPT_COMPRESS_LAYOUT, N.COMPRESS_LANES)
    cd.ctx.set_option(N.OPT_TABLE_PROBE_TRIES, 1)
    compared += _compare_batch(cd, d2, o2, l2, O.HASH_CRC32C, "9 000 mixed fragments in slices of 4 096")
    log_session(test="batches_beyond_one_launch_slice", blocks_compared=compared, result="all equal, and back")
