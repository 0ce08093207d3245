"""The copy step of the small-block team decoder (snappier_amd/csrc/decompress_small.hip, k_decompress_teams), restated on the CPU:
a tag's bytes move TEAM at a time, one per lane; a tag that reads nothing it writes takes independent steps, a pattern copy
(offset < length) reads `dist` bytes back and doubles `dist` while it is shorter than what the tag has produced.  For every
(offset, length, team size) the result must equal the reference's byte-serial IncrementalCopy (CopyHelpers.cs:222-230), and no
step may read a byte that the same step writes (the lanes of a step run at once)."""
import itertools

import pytest


def team_copy(buf: bytearray, op: int, off: int, length: int, team: int):
    """out[op + k] = out[op - off + k] for k < length, as the kernel does it.  Returns the number of steps."""
    steps = 0
    if off >= length:                                         # dist >= len: independent steps (four in flight in the kernel)
        for k0 in range(0, length, team):
            src = [buf[op - off + k] for k in range(k0, min(k0 + team, length))]
            for i, v in enumerate(src):
                buf[op + k0 + i] = v
            steps += 1
        return steps
    frm, dist, done = op - off, off, 0
    while done < length:
        w = min(dist, team, length - done)
        reads = [frm + done + t for t in range(w)]
        writes = [op + done + t for t in range(w)]
        assert not set(reads) & set(writes), "a step reads what it writes"
        assert all(r < op + done for r in reads), "a step reads a byte that does not exist yet"
        vals = [buf[r] for r in reads]
        for wpos, v in zip(writes, vals):
            buf[wpos] = v
        done += w
        steps += 1
        if dist <= done and dist < team:
            frm -= dist
            dist *= 2
    return steps


@pytest.mark.parametrize("team", [4, 8, 16])
def test_team_copy_equals_incremental_copy(team):
    for off, length in itertools.product(range(1, 70), range(1, 65)):
        op = 80
        base = bytearray((i * 37 + 11) & 255 for i in range(op)) + bytearray(length + 8)
        want = bytearray(base)
        for k in range(length):                               # IncrementalCopySlow: byte by byte
            want[op + k] = want[op - off + k]
        got = bytearray(base)
        steps = team_copy(got, op, off, length, team)
        assert got == want, (team, off, length)
        assert steps <= (length + team - 1) // team + 5       # doubling: a run-length pattern takes log steps, not `length`
