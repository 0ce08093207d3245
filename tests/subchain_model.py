"""Executable CPU model of the sub-chain tag parse of k_decompress_chains (snappier_amd/csrc/decompress.hip, FRONT = 3; DESIGN.md §4.1, HISTORY.md §4.1c).

TEST INFRASTRUCTURE ONLY.  It restates, lane by lane, what one wavefront does to find the tag starts of a 2 KiB super-window:
  A   lane k walks a chain of tags through region k (32 bytes) from the region's first byte, recording the positions visited (V_k);
  A'  it walks on until it lands on a position the owner of that region visited (merge: m_k, nx_k), recording the overrun
      positions (O_k), at most CAP bytes past the start of the region its overrun began in (then nx_k = UNMERGED);
  R   the lanes reachable from lane 0 along nx are on the true chain, each entered at its predecessor's merge position; an
      UNMERGED true chain is walked on tag by tag until it merges or leaves the window (the kernel's scalar fallback);
  T   true tag starts = V_k from the entry on, plus O_k, of every lane on the chain.
The claim the kernel rests on -- T equals the tag starts a sequential walk from the window's first byte visits, and `consumed` is
where that walk leaves the window -- is what tests/test_subchain_model.py checks on the corpus, on hand-built streams and on garbage.
"""
from __future__ import annotations

R = 32                 # bytes per lane region
LANES = 64
W = R * LANES          # the super-window
CAP = 128              # overrun bytes before a chain is left to the fallback
END, UNMERGED = 64, 65


def tag_advance(buf: bytes, p: int) -> int:
    """Bytes from the tag at p to the next tag (decompress.hip tag_advance_staged): at least 2; a long literal's length saturates."""
    c = buf[p]
    t, h = c & 3, c >> 2
    if t:
        return (2, 3, 5)[t - 1]
    if h < 60:
        return h + 2
    ex = h - 59
    tr = int.from_bytes(buf[p + 1:p + 1 + ex], "little")
    return 2 + ex + min(tr, 0x3FFFFFFF)


def sequential(buf: bytes, L: int):
    """The reference: tag starts below L visited from position 0, and the first position at or beyond L."""
    pos, p = [], 0
    while p < L:
        pos.append(p)
        p += tag_advance(buf, p)
    return pos, p


def window(buf: bytes, avail: int, stats: dict | None = None):
    """One super-window over buf (the input from the window's first byte on, avail >= 72 bytes of it valid).
    Returns (tag starts, consumed) as the kernel computes them."""
    assert avail >= 72
    L = min(W, avail) - 8
    V = [0] * LANES
    x = [0] * LANES
    for k in range(LANES):                                   # A
        p = R * k
        lim = min(p + R, L)
        while p < lim:
            V[k] |= 1 << (p - R * k)
            p += tag_advance(buf, p)
        x[k] = p
    m, nx, O = [0] * LANES, [END] * LANES, [dict() for _ in range(LANES)]
    trips = 0
    for k in range(LANES):                                   # A'
        p = x[k]
        obase = p & ~(R - 1)
        go = p < L
        t = 0
        while go:
            hit = (V[p >> 5] >> (p & 31)) & 1
            rel = p - obase
            if hit or rel >= CAP:
                nx[k] = (p >> 5) if hit else UNMERGED
                break
            O[k][p] = True
            p += tag_advance(buf, p)
            go = p < L
            t += 1
        m[k] = p
        trips = max(trips, t)
    # R: reachability from lane 0 (the kernel does it by pointer doubling; the result is this walk)
    T = set()
    k, e = 0, 0
    active, unmerged, slow = 0, 0, 0
    while True:
        active += 1
        T.update(R * k + i for i in range(R) if (V[k] >> i) & 1 and R * k + i >= e)
        T.update(O[k])
        mk, nk = m[k], nx[k]
        if nk == UNMERGED:                                   # the fallback: walk on until it merges or leaves
            unmerged += 1
            nk = END
            while mk < L:
                if (V[mk >> 5] >> (mk & 31)) & 1:
                    nk = mk >> 5
                    break
                T.add(mk)
                mk += tag_advance(buf, mk)
                slow += 1
        if nk >= END:
            consumed = mk
            break
        e, k = mk, nk
    if stats is not None:
        stats["windows"] = stats.get("windows", 0) + 1
        stats["tokens"] = stats.get("tokens", 0) + len(T)
        stats["overrun_trips"] = stats.get("overrun_trips", 0) + trips
        stats["active"] = stats.get("active", 0) + active
        stats["lanes"] = stats.get("lanes", 0) + min(LANES, (L + R - 1) // R)
        stats["unmerged"] = stats.get("unmerged", 0) + unmerged
        stats["slow_tags"] = stats.get("slow_tags", 0) + slow
    return sorted(T), consumed


def stream_windows(comp: bytes, stats: dict | None = None):
    """All super-windows of one compressed block, as the kernel takes them: (window start, tag starts, consumed) each."""
    n = len(comp)
    ip = 0
    while comp[ip] & 0x80:
        ip += 1
    ip += 1
    out = []
    pad = comp + bytes(W + 16)
    while ip + 72 <= n:
        pos, consumed = window(pad[ip:], n - ip, stats)
        out.append((ip, pos, consumed))
        ip += consumed
    return out
