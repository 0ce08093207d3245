"""Executable CPU model of the OUTPUT-granular batch executor of k_decompress_ring (snappier_amd/csrc/decompress.hip, FRONT = 4).

The kernel's parse (phases A, A', R, T) is the sub-chain parse of FRONT = 3 and has its own model (tests/subchain_model.py); what is new in
FRONT = 4 is how a batch of <= 64 tags EXECUTES, and that is restated here lane by lane with the kernel's own data structures:
  * the ring of the last RING output bytes, indexed by (output position + g0) & (RING - 1);
  * one virtual address space for every source byte: [0, RING) the ring, [K_IN, ..) the staged input (literals), [K_FAR, ..) the far pieces;
  * per batch: `bits` (a mark at the byte BEFORE every tag's first byte), `rec` (one dword D per tag: -offset for a copy inside the ring,
    else (source VA - tag start - pb0) mod 2^20 with bit 30 set), far units (16-byte pieces of copies older than the ring and of literals
    beyond the staged input, prefix-numbered over the far lanes);
  * sub-steps of 64 output bytes: tag of a byte = popcount of the marks below it (v_mbcnt), source address from D, sources inside the
    sub-step resolved by pointer doubling over the lanes (ds_bpermute), one byte store per lane;
  * the ring leaves for "global memory" in aligned 16-byte units after every batch (bytes at the block's two ends).
tests/test_ring_model.py checks the model against a plain sequential decode on the corpus, on low-entropy blocks (pattern copies: long
doubling chains), for several ring sizes and misalignments g0.  Pure numpy; no GPU."""
import numpy as np

WAVE = 64
K_STAGE = 2048          # staged input bytes per super-window (kW)
FAR_UNITS = 64


def parse_tags(z: bytes):
    """Snappy block -> (declared length, [(is_literal, length, offset, body position in z)])."""
    n, i, declared, shift = len(z), 0, 0, 0
    while True:
        c = z[i]
        i += 1
        declared |= (c & 127) << shift
        shift += 7
        if c < 128:
            break
    tags = []
    while i < n:
        c = z[i]
        t = c & 3
        if t == 0:
            ln = (c >> 2) + 1
            i += 1
            if ln > 60:
                k = ln - 60
                ln = int.from_bytes(z[i:i + k], "little") + 1
                i += k
            tags.append((True, ln, 0, i))
            i += ln
        elif t == 1:
            tags.append((False, ((c >> 2) & 7) + 4, ((c >> 5) << 8) | z[i + 1], 0))
            i += 2
        elif t == 2:
            tags.append((False, (c >> 2) + 1, z[i + 1] | (z[i + 2] << 8), 0))
            i += 3
        else:
            tags.append((False, (c >> 2) + 1, int.from_bytes(z[i + 1:i + 5], "little"), 0))
            i += 5
    return declared, tags


class RingExecutor:
    """State of one wavefront while it decodes one block."""

    def __init__(self, z: bytes, declared: int, ring: int = 2048, g0: int = 0, span_cap: int = 1024):
        assert ring & (ring - 1) == 0 and span_cap + 158 <= ring and span_cap % 64 == 0
        self.z = np.frombuffer(z, dtype=np.uint8)
        self.RING, self.g0, self.SPAN = ring, g0, span_cap
        self.K_IN = ring
        self.K_FAR = ring + K_STAGE + 16
        self.lds = np.zeros(self.K_FAR + FAR_UNITS * 16 + 64, dtype=np.uint8)   # ring | staged input | far pieces
        self.out = np.zeros(declared, dtype=np.uint8)                           # "global memory"
        self.op = 0
        self.wo = 0                                                             # output bytes below this have been written out
        self.stats = dict(batches=0, substeps=0, dep_substeps=0, rounds=0, far_tags=0, far_overflow_cuts=0)

    # -- ring -> global memory: whole 16-byte units of the BIASED position below op (final: everything) -----------------------------------
    def write_out(self, final: bool):
        lim = self.op if final else ((self.op + self.g0) & ~15) - self.g0
        lim = max(lim, self.wo)
        for p in range(self.wo, lim):
            self.out[p] = self.lds[(p + self.g0) & (self.RING - 1)]
        self.wo = lim

    def stage_window(self, wbase: int):
        """The super-window that starts at input position wbase: min(K_STAGE, rest) bytes into the LDS."""
        self.wbase = wbase
        self.staged = min(K_STAGE, len(self.z) - wbase)
        self.lds[self.K_IN:self.K_IN + self.staged] = self.z[wbase:wbase + self.staged]

    def long_literal(self, body: int, ln: int):
        """A literal of more than 64 bytes: input -> output directly, its last bytes into the ring (kernel: ne == 0, f0 == 3)."""
        self.write_out(True)
        self.out[self.op:self.op + ln] = self.z[body:body + ln]
        keep = min(ln, self.RING)
        for i in range(keep):
            self.lds[(self.op + self.g0 + ln - keep + i) & (self.RING - 1)] = self.z[body + ln - keep + i]
        self.op += ln
        self.wo = self.op

    def batch(self, tags):
        """tags: the next <= 64 (is_literal, len <= 64, offset, body position).  Returns how many of them the batch took."""
        R, g0 = self.RING, self.g0
        mark, pb0 = self.op, self.op + g0
        # --- decode(): prefix sum, far classification, far units, how many tags fit ---
        incl, units_before, ne = 0, 0, 0
        rel, D, far = [], [], []
        for is_lit, ln, off, body in tags[:WAVE]:
            start = incl
            incl += ln
            body_w = body - self.wbase                                         # position inside the staged window
            is_far = (body_w + ln > self.staged) if is_lit else (off > R - 64)
            units = (ln + 15) >> 4
            if incl > self.SPAN or (is_far and units_before + units > FAR_UNITS):
                self.stats["far_overflow_cuts"] += int(incl <= self.SPAN)
                break
            if is_far:
                va = self.K_FAR + 16 * units_before
                src = self.z[body:body + ln] if is_lit else self.out[mark + start - off: mark + start - off + ln]
                assert is_lit or mark + start - off + ln <= self.wo, "a far copy reads bytes that have not left the ring yet"
                self.lds[va:va + ln] = src                                      # (the kernel: 16-byte pieces loaded from global memory)
                units_before += units
                self.stats["far_tags"] += 1
            else:
                va = self.K_IN + body_w
            D.append((((va - start - pb0) & 0xFFFFF) | 0x40000000) if (is_lit or is_far) else ((-off) & 0xFFFFFFFF))
            rel.append(start)
            ne += 1
        assert ne >= 1
        span = rel[-1] + tags[ne - 1][1]
        # --- install(): marks and recs ---
        bits = np.zeros(2048, dtype=bool)
        for r in rel:
            if r:
                bits[r - 1] = True
        rec = np.array(D, dtype=np.uint64)
        # --- sub-steps ---
        lane = np.arange(WAVE)
        tbase = 0
        for sb in range(0, span, WAVE):
            w = bits[sb:sb + WAVE]
            below = np.concatenate([[0], np.cumsum(w)[:-1]])                    # v_mbcnt: marks below each lane
            ti = tbase + below
            tbase += int(w.sum())
            Dl = rec[np.minimum(ti, ne - 1)].astype(np.int64)
            pr = sb + lane
            inside = pr < span
            pb = pb0 + pr
            ring_copy = (Dl & 0x80000000) != 0                                  # D < 0 as an i32
            Ds = np.where(ring_copy, Dl - (1 << 32), Dl)                         # signed value
            a = np.where(ring_copy, (pb + Ds) & (R - 1), (pb + Ds) & 0xFFFFF)
            ptr = lane + Ds                                                     # ring copies: lane - off
            dep = inside & ring_copy & (ptr >= 0)
            byte = self.lds[np.where(dep, 0, np.minimum(a, len(self.lds) - 1))].astype(np.int64)
            self.stats["substeps"] += 1
            if dep.any():
                self.stats["dep_substeps"] += 1
                p4 = np.where(dep, ptr, lane | 64)                              # bit 6 here (bit 8 of the byte address there): a root
                while True:
                    p4 = p4[p4 & 63]                                            # ds_bpermute
                    self.stats["rounds"] += 1
                    if ((p4 & 64) != 0).all():
                        break
                byte = byte[p4 & 63]
            for k in np.nonzero(inside)[0]:
                self.lds[int(pb[k]) & (R - 1)] = byte[k]
        self.op += span
        self.stats["batches"] += 1
        self.write_out(False)
        return ne


def decode_with_ring_model(z: bytes, ring: int = 2048, g0: int = 0, span_cap: int = 1024):   # (the kernel's shipped geometry)
    """Decodes a WELL-FORMED Snappy block through the model; returns (bytes, stats)."""
    declared, tags = parse_tags(z)
    ex = RingExecutor(z, declared, ring, g0, span_cap)
    i = 0
    # super-windows: as in the kernel a window is re-staged whenever the next tag starts beyond the staged bytes (here: by tag index)
    pos_of = []
    # input position of every tag start (needed to know which window it lies in)
    p = 0
    while z[p] & 0x80:
        p += 1
    p += 1
    for is_lit, ln, off, body in tags:
        pos_of.append(p)
        c = z[p]
        t = c & 3
        if t == 0:
            p = body + ln
        else:
            p += 1 + (4 if t == 3 else t)
    ex.stage_window(pos_of[0] if pos_of else 0)
    while i < len(tags):
        if pos_of[i] >= ex.wbase + ex.staged - 8:                               # L = staged - 8: tags start below it
            ex.stage_window(pos_of[i])
        is_lit, ln, off, body = tags[i]
        if is_lit and ln > 64:
            ex.long_literal(body, ln)
            i += 1
            continue
        j = i
        while j < len(tags) and j - i < WAVE and pos_of[j] < ex.wbase + ex.staged - 8 and not (tags[j][0] and tags[j][1] > 64):
            j += 1
        if j == i:                                                              # (the tail of the input: the kernel's serial loop)
            ex.write_out(True)
            if is_lit:
                ex.out[ex.op:ex.op + ln] = ex.z[body:body + ln]
            else:
                for k in range(ln):
                    ex.out[ex.op + k] = ex.out[ex.op + k - off]
            for k in range(ln):
                ex.lds[(ex.op + ex.g0 + k) & (ex.RING - 1)] = ex.out[ex.op + k]
            ex.op += ln
            ex.wo = ex.op
            i += 1
            continue
        i += ex.batch(tags[i:j])
    ex.write_out(True)
    assert ex.op == declared
    return ex.out.tobytes(), ex.stats
