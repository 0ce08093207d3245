"""Deterministic synthetic inputs shared by tests and bench.py (numpy, CPU).

SURVEY.md section 8(d): config 2 = html-like blocks (html fixture tiled at a per-block offset + 1 % byte mutations),
config 3 = low-entropy ~90 %-match blocks, RandomData = restatement of SnappyTests.cs:401-446 with a documented
PRNG (System.Random(301) itself is not reproducible here: parity is on the property, not the sequence).
"""
import numpy as np

MASK64 = (1 << 64) - 1


def splitmix64(state: int):
    """One step of splitmix64 -> (new_state, output)."""
    state = (state + 0x9E3779B97F4A7C15) & MASK64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return state, z ^ (z >> 31)


def splitmix64_vec(seed: np.ndarray, n: int) -> np.ndarray:
    """n outputs of splitmix64 for each seed (vectorised): out[i, k] = k-th output of stream seed[i]."""
    seed = seed.astype(np.uint64)
    k = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        st = seed[:, None] + k[None, :] * np.uint64(0x9E3779B97F4A7C15)
        z = st
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


HTML_SEED = 0x5EED0001
LOWENT_SEED = 0x5EED0003
GAMMA = 0x9E3779B97F4A7C15


def mix64(z: int) -> int:
    """splitmix64's output function on an already-advanced state."""
    z &= MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


MIXED_SEED = 0x5EED0005


def corpus_blocks(files, first_block: int, nblocks: int, seed: int, block: int = 65536) -> np.ndarray:
    """Configs 2 and 5: block b takes file f = b mod len(files), tiles it cyclically from offset (b*4099) mod len_f;
    byte q of the block is then decided by draw q of the block's splitmix64 stream (seed ^ b), i.e.
    r = mix64((seed ^ b) + (q+1)*GAMMA): if r % 100 == 0 the byte is replaced by (r >> 32) & 0xff (~1 % mutations).
    Same arithmetic as the device generator (snappier_amd/csrc/datagen.hip)."""
    out = np.empty((nblocks, block), dtype=np.uint8)
    q = np.arange(block, dtype=np.uint64)
    for i in range(nblocks):
        b = first_block + i
        src = np.frombuffer(files[b % len(files)], dtype=np.uint8)
        L = len(src)
        start = (b * 4099) % L
        out[i] = src[((np.uint64(start) + q) % np.uint64(L)).astype(np.int64)]
    bb = np.arange(first_block, first_block + nblocks, dtype=np.uint64)
    r = splitmix64_vec(np.uint64(seed) ^ bb, block)
    mut = (r % np.uint64(100)) == 0
    val = ((r >> np.uint64(32)) & np.uint64(0xFF)).astype(np.uint8)
    return np.ascontiguousarray(np.where(mut, val, out).reshape(-1))


def html_like_blocks(html: bytes, first_block: int, nblocks: int, block: int = 65536) -> np.ndarray:
    """Config 2 = corpus_blocks over the single html fixture with HTML_SEED."""
    return corpus_blocks([html], first_block, nblocks, HTML_SEED, block)


PERIODS = [1, 2, 3, 4, 7, 8, 16, 64]


def low_entropy_block(b: int, block: int = 65536) -> np.ndarray:
    """Config 3: a block is a sequence of runs.  Per run draw r from splitmix64(seed = LOWENT_SEED ^ b):
    r % 10 != 0 (p = 0.9): pattern run, period P = PERIODS[(r >> 8) & 7], length L = 16 + (r >> 16) % 497 (16..512);
    else: noise run, P = L = 1 + (r >> 16) % 16.  A second draw r2 seeds the run's bytes:
    byte j of the run = (mix64(r2 + (j % P)) >> 24) & 0xff.  Scalar loop: small test sizes only; the device
    generator (snappier_amd/csrc/datagen.hip) does the same arithmetic, one block per thread."""
    out = np.empty(block, dtype=np.uint8)
    st = (LOWENT_SEED ^ b) & MASK64
    pos = 0
    while pos < block:
        st, r = splitmix64(st)
        st, r2 = splitmix64(st)
        if r % 10 != 0:
            P = PERIODS[(r >> 8) & 7]
            L = 16 + (r >> 16) % 497
        else:
            L = 1 + (r >> 16) % 16
            P = L
        pat = np.array([(mix64(r2 + j) >> 24) & 0xFF for j in range(P)], dtype=np.uint8)
        L = min(L, block - pos)
        out[pos:pos + L] = np.tile(pat, L // P + 1)[:L]
        pos += L
    return out


def random_data_case(i: int, rng: np.random.Generator) -> bytes:
    """Restatement of the generator in SnappyTests.RandomData (SnappyTests.cs:401-446): run-length-skewed bytes;
    the first 100 cases are 64-128 KiB over the full byte range, the rest < 4 KiB over tiny alphabets."""
    length = int(rng.integers(0, 4095))
    if i < 100:
        length = 65536 + int(rng.integers(0, 65535))
    buf = np.zeros(length, dtype=np.uint8)
    size = 0
    while size < length:
        run = 1
        if rng.integers(0, 9) == 0:
            skewed = int(rng.integers(0, 8))
            hi = (1 << skewed) - 1
            run = int(rng.integers(0, hi)) if hi > 0 else 0
        c = int(rng.integers(0, 255))
        if i >= 100:
            skewed = int(rng.integers(0, 3))
            hi = (1 << skewed) - 1
            c = int(rng.integers(0, hi)) if hi > 0 else 0
        buf[size:size + min(run, length - size)] = c
        size += run
    return buf.tobytes()
