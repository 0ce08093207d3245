"""CPU model of the tag index of one large block (snappier_amd/csrc/tag_index.hip, round 5): which entry point of every chunk of the stream is the true
one -- without walking the stream from its first tag.  Same steps as the kernels, small chunk sizes allowed so that tests reach every path:
  table      (output bytes, first tag start at or after the end of the sub-chunk) for a walk that enters at byte j (pointer doubling in the kernel;
             here: a backward sweep, same result);
  candidates walks that enter a chunk at its first `probe` bytes have merged by its end: their landings in the next chunk are its candidate entries;
  rows       for each candidate: the sub-chunk entry points and the exit, output bytes from the entry; the exit as a row of the next chunk, the end of
             the stream, or a POSITION (not a candidate of the next chunk, or beyond it);
  scan       follows rows from chunk 0; a position is looked up when its chunk comes; a landing that has no row stops the scan (pending);
  fix        follows the true chain from the pending landing, adding rows (a tag that leaves its chunk by itself: from its own bytes), until the
             chain is on a candidate again; the scan runs again.
The reference result is the plain serial walk (SnappyDecompressor.DecompressAllTags, Snappier/Internal/SnappyDecompressor.cs:184-347, visits the tags
in this order)."""
from dataclasses import dataclass, field

BAD = 0xFFFFFFFF
MAXC = 8
WIDE = 0xFF
DONE, FAIL, BYPOS = MAXC, MAXC + 1, MAXC + 2


def decode_tag(s: bytes, p: int):
    """-> (output bytes, next tag start) of the tag that would start at p, or None when it is truncated."""
    n = len(s)
    c = s[p]
    t, hi6 = c & 3, c >> 2
    extra = (hi6 - 59 if hi6 >= 60 else 0) if t == 0 else (4 if t == 3 else t)
    if p + 1 + extra > n:
        return None
    trailer = int.from_bytes(s[p + 1: p + 1 + extra], "little")
    if t == 0:
        ln = trailer + 1 if hi6 >= 60 else hi6 + 1
        return ln, p + 1 + extra + ln
    return ((hi6 & 7) + 4 if t == 1 else hi6 + 1), p + 1 + extra


def serial_walk(s: bytes, hb: int):
    """The true tag starts and the output offset before each (the reference's order)."""
    out, p, op, n = [], hb, 0, len(s)
    while p < n:
        out.append((p, op))
        d = decode_tag(s, p)
        if d is None or d[1] > n:
            return out, None
        op += d[0]
        p = d[1]
    return out, (p, op)


@dataclass
class Chunk:
    base: int
    table: dict = field(default_factory=dict)          # j -> (sum, nx): first tag start at or after the end of j's sub-chunk
    ncand: int = 0
    key: list = field(default_factory=list)
    rec: list = field(default_factory=list)            # per row: [(ip, op)] for sc = 1..subs (the last: the next chunk's entry)
    nxt: list = field(default_factory=list)


class TagIndexModel:
    def __init__(self, stream: bytes, hb: int, chunk: int = 256, sub: int = 64, probe: int = 16):
        self.s, self.hb, self.chunk, self.sub, self.probe = stream, hb, chunk, sub, probe
        self.n = len(stream)
        self.nchunks = (self.n - hb + chunk - 1) // chunk
        self.chunks = [Chunk(hb + k * chunk) for k in range(self.nchunks)]
        self.tables_built = 0
        self.passes = 0

    # -- step 1: the table of one chunk (what 12 rounds of pointer doubling leave) ---------------------------------------------------------
    def build_table(self, k: int):
        c = self.chunks[k]
        if c.table:
            return
        self.tables_built += 1
        end = min(c.base + self.chunk, self.n)
        for j in range(end - 1, c.base - 1, -1):
            d = decode_tag(self.s, j)
            if d is None or d[1] > self.n:
                c.table[j] = None                                          # kFar: irregular from here
                continue
            ln, nx = d
            sub_end = c.base + ((j - c.base) // self.sub + 1) * self.sub
            if nx < sub_end and nx < self.n:
                t = c.table[nx]
                c.table[j] = None if t is None else (ln + t[0], t[1])
            else:
                c.table[j] = (ln, nx)

    def walk_chunk(self, k: int, ip: int):
        """Step 2's walk: rec[sc] = (ip, op) at the entry of sub-chunk sc, rec[subs] = the next chunk's entry; op counted from the entry."""
        c = self.chunks[k]
        subs = self.chunk // self.sub
        rec, op = [], 0
        for sc in range(subs):
            rec.append((ip, op))
            sub_end = c.base + (sc + 1) * self.sub
            if ip == BAD or ip >= self.n or ip >= sub_end:
                continue
            t = c.table[ip]
            if t is None:
                ip = BAD
                continue
            op += t[0]
            ip = t[1]
        rec.append((ip, op))
        return rec

    # -- step 2a: candidates and their rows ------------------------------------------------------------------------------------------------
    def candidates(self):
        lands = [[self.hb]] + [[] for _ in range(self.nchunks)]
        for k in range(self.nchunks):
            self.build_table(k)
            c = self.chunks[k]
            end = c.base + self.chunk
            got, wide = [], False
            for j in range(c.base, min(c.base + self.probe, self.n)):
                land = self.walk_chunk(k, j)[-1][0]
                if land != BAD and end <= land < end + self.chunk and land < self.n and land not in got:
                    if len(got) == MAXC:
                        wide = True
                        break
                    got.append(land)
            lands[k + 1] = None if wide else got
        for k in range(self.nchunks):
            c = self.chunks[k]
            if lands[k] is None:
                c.ncand = WIDE
                continue
            c.ncand = len(lands[k])
            for key in lands[k]:
                self._add_row(k, key, lands[k + 1])

    def _add_row(self, k: int, key: int, next_keys):
        c = self.chunks[k]
        rec = self.walk_chunk(k, key)
        c.key.append(key)
        c.rec.append(rec[1:])
        out = rec[-1][0]
        nx = FAIL if out == BAD else DONE if out == self.n else BYPOS
        if nx == BYPOS and next_keys is not None and out in next_keys:
            nx = next_keys.index(out)
        c.nxt.append(nx)

    # -- step 2b: the scan -------------------------------------------------------------------------------------------------------------------
    def scan(self):
        """-> ('done', entries) | ('pending', k, ip, op) | ('fail',); entries[k] = [(ip, op)] * subs, then the final entry."""
        subs = self.chunk // self.sub
        entries, kind, v, op = [], "row", 0, 0
        for k in range(self.nchunks):
            c = self.chunks[k]
            end = c.base + self.chunk
            if kind == "end":
                entries.append([(self.n, op)] * subs)
                continue
            if kind == "pos":
                if v >= end:
                    entries.append([(v, op)] * subs)
                    continue
                if c.ncand == WIDE or v not in c.key[: c.ncand]:
                    return ("pending", k, v, op)
                kind, v = "row", c.key.index(v)
            if c.ncand == WIDE or v >= c.ncand:
                return ("fail",)
            rec = c.rec[v]
            if any(ip == BAD for ip, _ in rec[:-1]):
                return ("fail",)
            entries.append([(c.key[v], op)] + [(ip, op + o) for ip, o in rec[:-1]])
            op += rec[-1][1]
            nx = c.nxt[v]
            if nx < MAXC:
                kind, v = "row", nx
            elif nx == DONE:
                kind = "end"
            elif nx == BYPOS:
                kind, v = "pos", rec[-1][0]
            else:
                return ("fail",)
        if kind != "end":
            return ("fail",)
        return ("done", entries, (self.n, op))

    # -- step 2c: rows for a landing that was no candidate ----------------------------------------------------------------------------------------
    def fix(self, k: int, ip: int):
        for _ in range(4096):
            c = self.chunks[k]
            end = c.base + self.chunk
            d = decode_tag(self.s, ip)
            far = d is not None and (self.s[ip] & 3) == 0 and end <= d[1] <= self.n
            if c.ncand == WIDE:
                c.ncand, c.key, c.rec, c.nxt = 0, [], [], []
            if c.ncand >= MAXC:
                return False
            if far:                                                         # the row from the tag's own bytes: no table
                subs = self.chunk // self.sub
                rec = [((d[1], d[0]) if c.base + sc * self.sub > ip else (ip, 0)) for sc in range(1, subs)] + [(d[1], d[0])]
            else:
                self.build_table(k)
                rec = self.walk_chunk(k, ip)[1:]
            out = rec[-1][0]
            nx = FAIL if out == BAD else DONE if out == self.n else BYPOS
            joined, ko = nx != BYPOS, 0
            if nx == BYPOS:
                ko = (out - self.hb) // self.chunk
                o = self.chunks[ko]
                if o.ncand != WIDE and out in o.key[: o.ncand]:
                    joined = True
                    if ko == k + 1:
                        nx = o.key.index(out)
            c.key.append(ip)
            c.rec.append(rec)
            c.nxt.append(nx)
            c.ncand += 1
            if joined:
                return True
            k, ip = ko, out
        return True

    def run(self, max_passes: int = 64):
        self.candidates()
        for self.passes in range(max_passes + 1):
            r = self.scan()
            if r[0] != "pending":
                return r
            if not self.fix(r[1], r[2]):
                return ("fail",)
        return ("fail",)


def reference_entries(stream: bytes, hb: int, chunk: int, sub: int):
    """What the serial walk gives: for every sub-chunk the first true tag start at or after... the entry the look-back records -- the state (ip, op)
    of the walk when it first stands at or beyond the sub-chunk's first byte."""
    tags, final = serial_walk(stream, hb)
    if final is None:
        return None
    n = len(stream)
    nchunks = (n - hb + chunk - 1) // chunk
    pts = tags + [final]
    out, i = [], 0
    for k in range(nchunks):
        row = []
        for sc in range(chunk // sub):
            start = hb + k * chunk + sc * sub
            while pts[i][0] < start and i + 1 < len(pts):
                i += 1
            row.append(pts[i] if pts[i][0] >= start else final)
            if sc == 0 and k == 0:
                row[0] = pts[0]
        out.append(row)
    return out, final
