"""pymodel.py -- a SECOND, independent statement of the codec semantics: pure Python, transcribed from the
language-neutral specification in SURVEY.md Appendix A (itself derived from Snappier/Internal/SnappyCompressor.cs:24-83,
174-688, HashTable.cs:57-126, SnappyDecompressor.cs:184-347, SnappyStreamCompressor.cs:194-261), not from snappy_oracle.c.

TEST INFRASTRUCTURE ONLY.  Its job is to pin snappy_oracle.c from a different lineage -- above all the hash of the
crc32c variant, which no reference fixture covers: here H_crc is the bit-serial definition ("32 reflected shift-xor
steps of bytes XOR mask with 0x82F63B78"), there it is the x86 crc32 instruction.  tests/test_pymodel.py requires
the two to agree byte for byte on the corpus, on edge lengths and on fuzzed inputs, for both hashes.
Pure-Python loops: small cases only (a 64 KiB block takes ~0.1 s).
"""
from __future__ import annotations

POLY = 0x82F63B78
HASH_CRC32C, HASH_MUL = 0, 1


def tsize(n: int) -> int:                                  # HashTable.cs:57-71
    if n > 16384:
        return 16384
    if n < 256:
        return 256
    return 2 << ((n - 1).bit_length() - 1)


def crc32c_step32(x: int) -> int:                          # Sse42.Crc32(crc, data) with x = crc ^ data   HashTable.cs:109-112
    for _ in range(32):
        x = (x >> 1) ^ (POLY if x & 1 else 0)
    return x


_STEP8 = None


def _step32_fast(x: int) -> int:
    """The same map through its linearity (four byte tables built from the bit-serial definition)."""
    global _STEP8
    if _STEP8 is None:
        _STEP8 = [[crc32c_step32(b << (8 * k)) for b in range(256)] for k in range(4)]
    t = _STEP8
    return t[0][x & 255] ^ t[1][(x >> 8) & 255] ^ t[2][(x >> 16) & 255] ^ t[3][x >> 24]


def h_crc(b: int, mask: int) -> int:
    return (_step32_fast(b ^ mask) & mask) >> 1


def h_mul(b: int, mask: int) -> int:                       # HashTable.cs:121-125
    return ((((0x1E35A7BD * b) & 0xFFFFFFFF) >> 17) & mask) >> 1


def varint32(v: int) -> bytes:
    out = bytearray()
    while v >= 128:
        out.append((v & 127) | 128)
        v >>= 7
    out.append(v)
    return bytes(out)


def _literal(out: bytearray, f: bytes, s: int, l: int) -> None:
    k = l - 1
    if k < 60:
        out.append(k << 2)
    else:
        c = (k.bit_length() - 1) // 8 + 1
        out.append((59 + c) << 2)
        out += k.to_bytes(c, "little")
    out += f[s:s + l]


def _c64(out: bytearray, off: int, l: int) -> None:
    if l < 12 and off < 2048:
        out.append(1 | ((l - 4) << 2) | ((off >> 8) << 5))
        out.append(off & 255)
    else:
        out.append(2 | ((l - 1) << 2))
        out += off.to_bytes(2, "little")


def _copy(out: bytearray, off: int, l: int) -> None:
    if l < 12:
        _c64(out, off, l)
        return
    while l >= 68:
        _c64(out, off, 64)
        l -= 64
    if l > 64:
        _c64(out, off, 60)
        l -= 60
    _c64(out, off, l)


def fragment(f: bytes, variant: int) -> bytes:             # Appendix A "fragment(f)"
    n = len(f)
    out = bytearray()
    H = h_crc if variant == HASH_CRC32C else h_mul
    ld32 = lambda p: int.from_bytes(f[p:p + 4], "little")   # noqa: E731
    ts = tsize(n)
    mask = 2 * (ts - 1)
    table = [0] * ts
    ip = 0
    if n >= 15:
        limit = n - 15
        while True:                                        # OUTER
            next_emit = ip
            ip += 1
            skip = 32
            found = False
            cand = 0
            if limit - ip >= 16:
                for j in range(16):
                    p = ip + j
                    d = ld32(p)
                    h = H(d, mask)
                    c = table[h]
                    table[h] = p
                    if ld32(c) == d:
                        ip, cand, found = p, c, True
                        break
                if not found:
                    ip += 16
                    skip += 16
            remainder = False
            if not found:
                while True:
                    d = ld32(ip)
                    bb = skip >> 5
                    skip += bb
                    nxt = ip + bb
                    if nxt > limit:
                        ip = next_emit
                        remainder = True
                        break
                    h = H(d, mask)
                    c = table[h]
                    table[h] = ip
                    if ld32(c) == d:
                        cand = c
                        break
                    ip = nxt
            if remainder:
                break
            _literal(out, f, next_emit, ip - next_emit)
            while True:                                    # repeat ... until ld32(cand) != d
                base = ip
                m = 4
                while ip + m < n and f[cand + m] == f[ip + m]:
                    m += 1
                ip += m
                _copy(out, base - cand, m)
                if ip >= limit:
                    remainder = True
                    break
                table[H(ld32(ip - 1), mask)] = ip - 1
                d = ld32(ip)
                h = H(d, mask)
                cand = table[h]
                table[h] = ip
                if ld32(cand) != d:
                    break
            if remainder:
                break
    if ip < n:
        _literal(out, f, ip, n - ip)
    return bytes(out)


def compress(x: bytes, variant: int = HASH_CRC32C) -> bytes:
    out = bytearray(varint32(len(x)))
    for o in range(0, len(x), 65536):
        out += fragment(x[o:o + 65536], variant)
    return bytes(out)


class Invalid(Exception):
    pass


def decompress(z: bytes) -> bytes:                          # Appendix A "decompress(z)" (strict: declared length is the bound)
    L, shift, i = 0, 0, 0
    while True:
        if i >= len(z) or i >= 5:
            raise Invalid("length")
        c = z[i]
        i += 1
        if (c & 127) & ~(0xFFFFFFFF >> shift) & 0xFF:
            raise Invalid("length")
        L |= (c & 127) << shift
        shift += 7
        if c < 128:
            break
    out = bytearray()
    n = len(z)
    while i < n and len(out) < L:
        c = z[i]
        t = c & 3
        if t == 0:
            l = (c >> 2) + 1
            i += 1
            if l > 60:
                k = l - 60
                if i + k > n:
                    raise Invalid("incomplete")
                l = int.from_bytes(z[i:i + k], "little") + 1
                i += k
            take = min(l, n - i)
            if len(out) + take > L:
                raise Invalid("too long")
            out += z[i:i + take]
            i += take
            if take < l:
                raise Invalid("incomplete")
            continue
        if t == 1:
            if i + 2 > n:
                break
            l, off = ((c >> 2) & 7) + 4, ((c >> 5) << 8) | z[i + 1]
            i += 2
        elif t == 2:
            if i + 3 > n:
                break
            l, off = (c >> 2) + 1, z[i + 1] | (z[i + 2] << 8)
            i += 3
        else:
            if i + 5 > n:
                break
            l, off = (c >> 2) + 1, int.from_bytes(z[i + 1:i + 5], "little")
            i += 5
        if off == 0 or off > len(out):
            raise Invalid("offset")
        if len(out) + l > L:
            raise Invalid("too long")
        for _ in range(l):
            out.append(out[-off])
    if len(out) < L:
        raise Invalid("incomplete")
    return bytes(out)


def crc32c(data: bytes) -> int:                             # Crc32CAlgorithm.cs:41-158, bit-serial
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (POLY if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def mask_crc(x: int) -> int:
    return (((x >> 15) | (x << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def frame(stream: bytes, variant: int = HASH_CRC32C) -> bytes:   # Appendix A "frame(stream)"
    out = bytearray(bytes([0xFF, 0x06, 0x00, 0x00, 0x73, 0x4E, 0x61, 0x50, 0x70, 0x59]))
    for o in range(0, len(stream), 65536):
        r = stream[o:o + 65536]
        z = compress(r, variant)
        typ, payload = (0, z) if len(z) < len(r) else (1, r)
        out.append(typ)
        out += (len(payload) + 4).to_bytes(3, "little")
        out += mask_crc(crc32c(r)).to_bytes(4, "little")
        out += payload
    return bytes(out)
