"""Known-answer vectors harvested from the reference's own unit tests (data only)."""

# Snappier.Tests/Internal/Crc32CAlgorithmTests.cs:8-11
CRC32C = [
    (b"123456789", 0xE3069283),
    (b"1234567890123456", 0x9AA4287F),
    (b"123456789012345612345678901234", 0xECC74934),
    (b"12345678901234561234567890123456", 0xCD486B4B),
]

# Snappier.Tests/Internal/VarIntEncodingReadTests.cs:7-22, VarIntEncodingWriteTests.cs:5-20
VARINT = [
    (0x00, bytes([0x00])),
    (0x01, bytes([0x01])),
    (0x7F, bytes([0x7F])),
    (0x80, bytes([0x80, 0x01])),
    (0x555, bytes([0xD5, 0x0A])),
    (0x7FFF, bytes([0xFF, 0xFF, 0x01])),
    (0xBFFF, bytes([0xFF, 0xFF, 0x02])),
    (0xFFFF, bytes([0xFF, 0xFF, 0x03])),
    (0x8000, bytes([0x80, 0x80, 0x02])),
    (0x5555, bytes([0xD5, 0xAA, 0x01])),
    (0xCAFEF00, bytes([0x80, 0xDE, 0xBF, 0x65])),
    (0xCAFEF00D, bytes([0x8D, 0xE0, 0xFB, 0xD7, 0x0C])),
    (0xFFFFFFFF, bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F])),
]
# VarIntEncodingReadTests.cs:24-36
VARINT_INCOMPLETE = [bytes([0x80]), bytes([0xD5]), bytes([0xFF, 0xFF]), bytes([0x80, 0x80]), bytes([0xD5, 0xAA]),
                     bytes([0x80, 0xDE, 0xBF]), bytes([0x8D, 0xE0, 0xFB, 0xD7]), bytes([0xFF, 0xFF, 0xFF, 0xFF])]
# VarIntEncodingReadTests.cs:84-89
VARINT_BAD = bytes([0xFF] * 6)

# Snappier.Tests/Internal/SnappyCompressorTests.cs:10-81  (expected, s1, s2, length)
FIND_MATCH_LENGTH = [
    (6, "012345", "012345", 6),
    (11, "01234567abc", "01234567abc", 11),
    (9, "01234567abc", "01234567axc", 9),
    (11, "01234567abc!", "01234567abc!", 11),
    (11, "01234567abc!", "01234567abc?", 11),
    (0, "01234567xxxxxxxx", "?1234567xxxxxxxx", 16),
    (1, "01234567xxxxxxxx", "0?234567xxxxxxxx", 16),
    (4, "01234567xxxxxxxx", "01237654xxxxxxxx", 16),
    (7, "01234567xxxxxxxx", "0123456?xxxxxxxx", 16),
    (8, "abcdefgh01234567xxxxxxxx", "abcdefgh?1234567xxxxxxxx", 24),
    (9, "abcdefgh01234567xxxxxxxx", "abcdefgh0?234567xxxxxxxx", 24),
    (12, "abcdefgh01234567xxxxxxxx", "abcdefgh01237654xxxxxxxx", 24),
    (15, "abcdefgh01234567xxxxxxxx", "abcdefgh0123456?xxxxxxxx", 24),
    (0, "01234567", "?1234567", 8),
    (1, "01234567", "0?234567", 8),
    (2, "01234567", "01?34567", 8),
    (3, "01234567", "012?4567", 8),
    (4, "01234567", "0123?567", 8),
    (5, "01234567", "01234?67", 8),
    (6, "01234567", "012345?7", 8),
    (7, "01234567", "0123456?", 8),
    (7, "01234567", "0123456?", 7),
    (7, "01234567!", "0123456??", 7),
    (10, "xxxxxxabcd", "xxxxxxabcd", 10),
    (10, "xxxxxxabcd?", "xxxxxxabcd?", 10),
    (13, "xxxxxxabcdef", "xxxxxxabcdefx", 13),
    (12, "xxxxxx0123abc!", "xxxxxx0123abc!", 12),
    (12, "xxxxxx0123abc!", "xxxxxx0123abc?", 12),
    (11, "xxxxxx0123abc", "xxxxxx0123axc", 13),
    (6, "xxxxxx0123xxxxxxxx", "xxxxxx?123xxxxxxxx", 18),
    (7, "xxxxxx0123xxxxxxxx", "xxxxxx0?23xxxxxxxx", 18),
    (8, "xxxxxx0123xxxxxxxx", "xxxxxx0132xxxxxxxx", 18),
    (9, "xxxxxx0123xxxxxxxx", "xxxxxx012?xxxxxxxx", 18),
    (6, "xxxxxx0123", "xxxxxx?123", 10),
    (7, "xxxxxx0123", "xxxxxx0?23", 10),
    (8, "xxxxxx0123", "xxxxxx0132", 10),
    (9, "xxxxxx0123", "xxxxxx012?", 10),
    (10, "xxxxxxabcd0123xx", "xxxxxxabcd?123xx", 16),
    (11, "xxxxxxabcd0123xx", "xxxxxxabcd0?23xx", 16),
    (12, "xxxxxxabcd0123xx", "xxxxxxabcd0132xx", 16),
    (13, "xxxxxxabcd0123xx", "xxxxxxabcd012?xx", 16),
    (10, "xxxxxxabcd0123", "xxxxxxabcd?123", 14),
    (11, "xxxxxxabcd0123", "xxxxxxabcd0?23", 14),
    (12, "xxxxxxabcd0123", "xxxxxxabcd0132", 14),
    (13, "xxxxxxabcd0123", "xxxxxxabcd012?", 14),
]

# Snappier.Tests/HelpersTests.cs:7-34
LEFT_SHIFT_OVERFLOWS_TRUE = [(2, 31), (0xFF, 25)]
LEFT_SHIFT_OVERFLOWS_FALSE = [(1, 31), (0xFF, 24), (0, 31)]

# Snappier.Tests/SnappyTests.cs:178-189
STRING_CASES = [
    b"",
    b"a",
    b"ab",
    b"abc",
    b"aaaaaaa" + b"b" * 16 + b"aaaaaabc",
    b"aaaaaaa" + b"b" * 256 + b"aaaaaabc",
    b"aaaaaaa" + b"b" * 2047 + b"aaaaaabc",
    b"aaaaaaa" + b"b" * 65536 + b"aaaaaabc",
    b"abcaaaaaaa" + b"b" * 65536 + b"aaaaaabc",
]

# model KATs for compress(html[0:65536]) -- mul == golden chunk 0 of html_x_4.snappy (pinned),
# crc32c == probe-model value (parity unpinned against real Snappier bytes).
HTML64K_MUL = (16533, "2f8a1e2979f6b2cb256046dab000c012c0bbca0846be628a26911220beac16f4")
HTML64K_CRC = (16446, "822945612f80e8d49f087640acbe415b9b11c59f686e17b8bfe833fe4768251e")
