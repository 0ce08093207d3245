"""64-lane executable model of the HIP compress kernel's algorithm (snappier_amd/csrc/compress.hip).

The kernel cannot run in the build container (no GPU), so its *algorithm* -- the speculative 64-probe round, the
in-register conflict resolution and the LDS table fix-up that together must reproduce the serial greedy parse of
SnappyCompressor.CompressFragment (SnappyCompressor.cs:174-415) bit-for-bit -- is modelled here lane by lane with
numpy arrays of 64 and checked against the C oracle on CPU (tests/test_wave_model.py).  Test infrastructure only.

Round structure (one wavefront = one fragment):
  kind A (fresh scan)  lane j            probes p = start + D[kbase + j]
  kind B (after a copy ending at ip < limit)
                       lane 0            inserts ip-1 only             (SnappyCompressor.cs:393-394)
                       lane 1            probes ip; a hit = back-to-back copy, no literal   (:395-398)
                       lane 2+j          probes p = ip+1 + D[j]        (next OUTER iteration, :198-341)
D[] is the universal probe-offset sequence of the skip heuristic (:227,319-320): D[0]=0, D[k+1]=D[k]+1+(D[k]>>5)
(skip_k = 32 + D[k]; the unrolled 16-probe section :230-313 is the same sequence).  Probe k is legal iff
start + D[k+1] <= limit (:323).
"""
import numpy as np

W = 64
NARROW = 16     # lanes speculated in the first round after a match
SMALL_R = 8     # up to this many real lanes, bucket conflicts are found by comparing hashes in registers
BLOCK = 65536


def probe_offsets(maxv=BLOCK + 4096):
    d = [0]
    while d[-1] <= maxv:
        d.append(d[-1] + 1 + (d[-1] >> 5))
    for _ in range(2 * W + 2):                           # slack so kbase + lane + 1 never indexes past the end
        d.append(d[-1] + 1 + (d[-1] >> 5))
    return np.array(d, dtype=np.int64)


D = probe_offsets()


def crc_lmap_rows():
    """Rows M_i such that bit i of crc32c_step32(x) = parity(x & M_i)  (the map is GF(2)-linear)."""
    def step32(x):
        for _ in range(32):
            x = (x >> 1) ^ (0x82F63B78 if x & 1 else 0)
        return x
    cols = [step32(1 << b) for b in range(32)]
    rows = []
    for i in range(32):
        m = 0
        for b in range(32):
            if (cols[b] >> i) & 1:
                m |= 1 << b
        rows.append(m)
    return rows


CRC_ROWS = crc_lmap_rows()


def popcount32(x):
    x = x.astype(np.uint64)
    c = np.zeros_like(x)
    for _ in range(32):
        c += x & np.uint64(1)
        x >>= np.uint64(1)
    return c


def hash_lanes(d, mask, variant):
    """Table entry index for each lane's 4 input bytes (HashTable.cs:91-126), as the kernel computes it."""
    d = d.astype(np.uint64)
    if variant == 0:   # crc32c: bits 1..14 via parity-of-AND rows; H(b) = Lmap(b ^ mask)
        x = d ^ np.uint64(mask)
        h = np.zeros_like(x)
        for i in range(1, 15):
            h |= (popcount32(x & np.uint64(CRC_ROWS[i])) & np.uint64(1)) << np.uint64(i)
    else:
        h = ((d * np.uint64(0x1E35A7BD)) & np.uint64(0xFFFFFFFF)) >> np.uint64(17)
    return ((h & np.uint64(mask)) >> np.uint64(1)).astype(np.int64)


def table_size(n):
    if n > 16384:
        return 16384
    if n < 256:
        return 256
    return 2 << (int(n - 1).bit_length() - 1)


def ld32(buf, pos):
    pos = np.asarray(pos, dtype=np.int64)
    return (buf[pos].astype(np.uint32) | (buf[pos + 1].astype(np.uint32) << 8) | (buf[pos + 2].astype(np.uint32) << 16)
            | (buf[pos + 3].astype(np.uint32) << 24))


def emit_literal(out, buf, s, length):
    k = length - 1
    if k < 60:
        out.append(k << 2)
    else:
        c = (int(k).bit_length() - 1) // 8 + 1
        out.append((59 + c) << 2)
        out.extend((k >> (8 * i)) & 0xFF for i in range(c))
    out.extend(buf[s:s + length].tolist())


def copy_tokens(off, length):
    """Closed form of EmitCopyLenLessThan12 / EmitCopyLenGreaterThanOrEqualTo12 (SnappyCompressor.cs:507-543):
    q tokens of 64, optionally one of 60, then the final 4..64-byte token."""
    out = []
    q = (length - 4) // 64 if length >= 68 else 0
    r = length - 64 * q
    tok64 = [2 | (63 << 2), off & 0xFF, off >> 8]
    out.extend(tok64 * q)
    if r > 64:
        out.extend([2 | (59 << 2), off & 0xFF, off >> 8])
        r -= 60
    if r < 12 and off < 2048:
        out.extend([1 | ((r - 4) << 2) | ((off >> 8) << 5), off & 0xFF])
    else:
        out.extend([2 | ((r - 1) << 2), off & 0xFF, off >> 8])
    return out


def _publish(table, h, p, who, rng):
    """Lanes `who` store p to table[h] in one LDS instruction; which lane wins a shared bucket is unspecified."""
    idx = np.nonzero(who)[0]
    if rng is not None:
        idx = rng.permutation(idx)
    for j in idx:
        table[h[j]] = p[j]


def compress_fragment_wave(frag: bytes, variant: int, stats=None, rng=None):
    """The kernel's algorithm for one fragment (<= 65536 B); returns the compressed fragment (no varint)."""
    n = len(frag)
    buf = np.frombuffer(frag + bytes(8), dtype=np.uint8)
    out = []
    if n < 15:                                           # SnappyCompressor.cs:190
        if n:
            emit_literal(out, buf, 0, n)
        return bytes(out)
    tsize = table_size(n)
    mask = 2 * (tsize - 1)
    table = np.zeros(tsize, dtype=np.int64)              # LDS, u16 entries
    limit = n - 15
    lanes = np.arange(W)

    next_emit = 0
    kind_b = False
    ip = 0            # kind B: position right after the last copy
    start = 1         # scan start (first probe position)
    kbase = 0
    rounds = slow = 0
    width = NARROW    # speculation width: NARROW lanes after a match, all 64 once a round found nothing
    while True:
        rounds += 1
        # ---- 1. positions and validity -----------------------------------------------------------------
        spec = lanes < width
        if kind_b and kbase == 0:
            k = lanes - 2
            p = np.where(lanes == 0, ip - 1, np.where(lanes == 1, ip, start + D[np.maximum(k, 0)]))
            legal = np.where(lanes < 2, True, start + D[np.maximum(k, 0) + 1] <= limit)
            probing = lanes >= 1
        else:
            k = kbase + lanes
            p = start + D[k]
            legal = start + D[k + 1] <= limit
            probing = np.ones(W, dtype=bool)
        valid = legal & spec
        # legality is monotone: once a lane is illegal all later ones are
        pv = np.where(valid, p, 0)
        d = ld32(buf, pv)
        h = hash_lanes(d, mask, variant)
        c = table[h]                                    # 2. LDS gather of pre-round candidates
        e = ld32(buf, c)                                # 3. candidate bytes
        stale = valid & probing & (e == d)
        stop = (stale | ~legal) & spec
        first0 = int(np.argmax(stop)) if stop.any() else width
        terminated = first0 < width and not legal[first0]
        in_r = (lanes < first0) | ((lanes == first0) & (not terminated))   # lanes whose effects may be real
        in_r &= valid
        # ---- 4. conflict detection -------------------------------------------------------------------------
        published = int(in_r.sum()) > SMALL_R
        if published:                                    # large R: publish p to the table, read back
            _publish(table, h, p, in_r, rng)
            r = table[h]
            conflict = bool((in_r & (r != p)).any())
        else:                                            # small R: pairwise bucket comparison in registers
            conflict = any(bool((in_r & (h == h[j]) & (lanes < j)).any()) for j in np.nonzero(in_r)[0])
            if not conflict:
                _publish(table, h, p, in_r, rng)         # buckets are pairwise distinct: plain stores
        m = first0 if (first0 < width and not terminated) else -1
        cand = int(c[m]) if m >= 0 else -1
        if conflict:
            slow += 1
            # ---- 5. exact resolution in registers (scalar loop over R) ---------------------------------
            m, cand = -1, -1
            for j in np.nonzero(in_r)[0]:
                if not probing[j]:
                    continue
                prev = np.nonzero(in_r & (h == h[j]) & (lanes < j))[0]
                if len(prev):
                    a = int(prev[-1])
                    if d[a] == d[j]:
                        m, cand = int(j), int(p[a])
                        break
                elif stale[j]:
                    m, cand = int(j), int(c[j])
                    break
            # ---- 6. table fix-up: restore, then lanes <= m (or all of R if no match) republish, max wins ----
            keep = in_r & ((lanes <= m) if m >= 0 else True)
            if published:
                for j in np.nonzero(in_r)[0]:
                    table[h[j]] = c[j]
            active = keep.copy()
            while active.any():
                _publish(table, h, p, active, rng)
                active = keep & (table[h] < p)
        if m < 0:
            if terminated:
                break                                    # SnappyCompressor.cs:323-327 -> emit_remainder from next_emit
            # No match among the lanes processed.  Normally that is all 64; if the slow path *destroyed* the stale
            # hit at first0 (an earlier lane in the same bucket with different bytes is the real candidate), only
            # lanes 0..first0 were processed and the scan resumes right after first0.
            done = width if first0 == width else first0 + 1
            if kind_b and kbase == 0:
                next_emit = ip                           # the post-copy probe missed: new OUTER iteration starts at ip
                kbase = max(done - 2, 0)
                kind_b = False
            else:
                kbase += done
            width = W
            continue
        # ---- 7. emit literal + copy ------------------------------------------------------------------------
        pm = int(p[m])
        if kind_b and kbase == 0 and m >= 2:
            next_emit = ip
        if pm > next_emit:
            emit_literal(out, buf, next_emit, pm - next_emit)
        # FindMatchLength(cand+4, pm+4, n): 64-lane byte compare + ballot + ctz
        matched = 4
        while pm + matched < n and buf[cand + matched] == buf[pm + matched]:
            matched += 1
        out.extend(copy_tokens(pm - cand, matched))
        ip = pm + matched
        next_emit = ip
        if ip >= limit:                                  # :381-384
            break
        kind_b, kbase, start = True, 0, ip + 1
        width = NARROW
    if next_emit < n:
        emit_literal(out, buf, next_emit, n - next_emit) # :406-411
    if stats is not None:
        stats["rounds"] = stats.get("rounds", 0) + rounds
        stats["slow"] = stats.get("slow", 0) + slow
    return bytes(out)


def varint(v):
    out = []
    while v >= 128:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def compress_wave(data: bytes, variant: int, stats=None, rng=None) -> bytes:
    out = varint(len(data))
    for s in range(0, len(data), BLOCK):
        out += compress_fragment_wave(data[s:s + BLOCK], variant, stats, rng)
    return out
