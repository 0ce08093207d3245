/*
 * snappy_oracle.c -- CPU restatement of Snappier's Snappy block codec, framing rules and CRC-32C.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (snappier_amd/, libsnappier_hip.so) may call, link or
 * import this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
 * or as the timed CPU baseline ("kind": "port").
 *
 * Parity pinning ("Oracle"):
 *   - decompress, CRC-32C, varint, FindMatchLength, framing and the hash=MUL compressor are pinned bit-exactly
 *     against the reference's own fixtures/KATs (tests/test_oracle_golden.py): all 10 chunks of html_x_4.snappy
 *     and alice29.snappy, baddata{1,2,3}.snappy, Crc32CAlgorithmTests, VarIntEncoding*Tests, SnappyCompressorTests.
 *   - hash=CRC32C compressor bytes: PARITY UNPINNED -- no reference test asserts compressed bytes, no fixture
 *     was produced with that hash and no .NET runtime exists here.  It differs from the pinned MUL model only
 *     in orc_hash(); the CRC-32C polynomial itself is pinned by the KATs.
 *
 * Every function cites the reference file:line (relative to /root/reference) that it restates.
 * Plain C11, little-endian host assumed (x86-64).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_OUTPUT_TOO_SMALL 1
#define ORC_ERR_BAD_OFFSET 2
#define ORC_ERR_TOO_LONG 3
#define ORC_ERR_INCOMPLETE 4
#define ORC_ERR_BAD_LENGTH 5
#define ORC_ERR_CRC_MISMATCH 6
#define ORC_ERR_CHUNK_TYPE 7
#define ORC_ERR_OVERLAP 8
#define ORC_ERR_BAD_ARG 9
#define ORC_ERR_TRUNCATED_STREAM 11

#define ORC_HASH_CRC32C 0
#define ORC_HASH_MUL 1

#define ORC_BLOCK_SIZE 65536u   /* Constants.cs:25-26 */
#define ORC_INPUT_MARGIN 15u    /* Constants.cs:27 */

static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }   /* Helpers.cs:87-98 */
static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }   /* Helpers.cs:100-111 */
static inline void st32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }                         /* Helpers.cs:113-122 */

/* Helpers.Log2Floor  Helpers.cs:171-179 (0 -> 0 like BitOperations.Log2) */
int orc_log2_floor(uint32_t n) { return n == 0 ? 0 : 31 - __builtin_clz(n); }

/* Helpers.LeftShiftOverflows  Helpers.cs:65-70 */
int orc_left_shift_overflows(uint8_t value, int shift) { return (value & ~(0xffffffffu >> shift)) != 0; }

/* Helpers.MaxCompressedLength  Helpers.cs:17-46 */
int64_t orc_max_fragment_compressed_length(int64_t n) { return 32 + n + n / 6 + 1; }
/* Snappy.GetMaxCompressedLength  Snappy.cs:20-24 */
int64_t orc_max_compressed_length(int64_t n) { return orc_max_fragment_compressed_length(n) + 5; }

/* ---------------------------------------------------------------- CRC-32C ------------------------------- */

/* Crc32CAlgorithm static ctor, first slice only  Crc32CAlgorithm.cs:15-36 (Poly 0x82F63B78, reflected) */
static uint32_t g_crc_table[8][256];
static int g_crc_table_ready = 0;
__attribute__((unused)) static void crc_init(void)
{
    if (g_crc_table_ready) return;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t r = i;
        for (int t = 0; t < 8; t++) {
            for (int k = 0; k < 8; k++) r = (r & 1) ? (0x82F63B78u ^ (r >> 1)) : (r >> 1);
            g_crc_table[t][i] = r;
        }
    }
    g_crc_table_ready = 1;
}

/* Crc32CAlgorithm.Append  Crc32CAlgorithm.cs:46-154 (table path semantics; init/xorout 0xFFFFFFFF) */
uint32_t orc_crc32c_append(uint32_t crc, const uint8_t* p, size_t n)
{
    uint32_t c = 0xffffffffu ^ crc;
#if defined(__SSE4_2__)
    while (n >= 8) { c = (uint32_t)__builtin_ia32_crc32di(c, ld64(p)); p += 8; n -= 8; }   /* :86-101 */
    while (n >= 4) { c = __builtin_ia32_crc32si(c, ld32(p)); p += 4; n -= 4; }               /* :104-108 */
    while (n--) c = __builtin_ia32_crc32qi(c, *p++);                                         /* :111-115 */
#else
    crc_init();
    while (n--) c = g_crc_table[0][(uint8_t)(c ^ *p++)] ^ (c >> 8);                          /* :148-151 */
#endif
    return c ^ 0xffffffffu;
}
/* Crc32CAlgorithm.Compute  Crc32CAlgorithm.cs:41-44 */
uint32_t orc_crc32c(const uint8_t* p, size_t n) { return orc_crc32c_append(0, p, n); }
/* pure bitwise definition, used by the tests to cross-check the fast paths above */
uint32_t orc_crc32c_bitwise(const uint8_t* p, size_t n)
{
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c & 1) ? (0x82F63B78u ^ (c >> 1)) : (c >> 1);
    }
    return c ^ 0xffffffffu;
}
/* Crc32CAlgorithm.ApplyMask  Crc32CAlgorithm.cs:156-158 */
uint32_t orc_crc32c_mask(uint32_t x) { return ((x >> 15) | (x << 17)) + 0xa282ead8u; }

/* ---------------------------------------------------------------- varint -------------------------------- */

/* VarIntEncoding.TryWriteSlow  VarIntEncoding.Write.cs:5-79.  Returns bytes written, 0 if cap too small. */
int orc_varint_write(uint8_t* out, size_t cap, uint32_t v)
{
    int need = v < (1u << 7) ? 1 : v < (1u << 14) ? 2 : v < (1u << 21) ? 3 : v < (1u << 28) ? 4 : 5;
    if (cap < (size_t)need) return 0;
    for (int i = 0; i < need - 1; i++) { out[i] = (uint8_t)(v | 0x80); v >>= 7; }
    out[need - 1] = (uint8_t)v;
    return need;
}

/* VarIntEncoding.TryReadSlow  VarIntEncoding.Read.cs:38-79.
 * Returns ORC_OK (Done), ORC_ERR_INCOMPLETE (NeedMoreData) or ORC_ERR_BAD_LENGTH (InvalidData). */
int orc_varint_read(const uint8_t* in, size_t n, uint32_t* value, int* bytes_read)
{
    uint32_t result = 0;
    int shift = 0;
    size_t i = 0;
    *bytes_read = 0;
    *value = 0;
    while (i < n) {
        uint8_t c = in[i++];
        uint8_t val = c & 0x7f;
        if (orc_left_shift_overflows(val, shift)) return ORC_ERR_BAD_LENGTH;   /* :50-54 */
        result |= (uint32_t)val << shift;
        shift += 7;
        if (c < 128) { *value = result; *bytes_read = (int)i; return ORC_OK; }
        if (shift >= 32) return ORC_ERR_BAD_LENGTH;                            /* :65-69 */
    }
    return ORC_ERR_INCOMPLETE;                                                 /* :72-76 */
}

/* Snappy.GetUncompressedLength  Snappy.cs:136-137 / VarIntEncoding.Read  VarIntEncoding.Read.cs:16-24:
 * anything but Done is "Invalid stream length". */
int orc_get_uncompressed_length(const uint8_t* in, size_t n, uint32_t* out_len, uint32_t* header_bytes)
{
    int br = 0;
    uint32_t v = 0;
    int st = orc_varint_read(in, n, &v, &br);
    if (st != ORC_OK) return ORC_ERR_BAD_LENGTH;
    if (out_len) *out_len = v;
    if (header_bytes) *header_bytes = (uint32_t)br;
    return ORC_OK;
}

/* ---------------------------------------------------------------- hash table ---------------------------- */

/* HashTable.CalculateTableSize  HashTable.cs:57-71 */
uint32_t orc_table_size(uint32_t input_size)
{
    if (input_size > 16384) return 16384;
    if (input_size < 256) return 256;
    return 2u << orc_log2_floor(input_size - 1);
}

/* the x86 crc32 r32, r/m32 instruction = 32 reflected shift/xor steps over (crc ^ data), no inversion */
static inline uint32_t crc32c_u32_step(uint32_t crc, uint32_t data)
{
#if defined(__SSE4_2__)
    return __builtin_ia32_crc32si(crc, data);
#else
    uint32_t x = crc ^ data;
    for (int k = 0; k < 32; k++) x = (x & 1) ? (0x82F63B78u ^ (x >> 1)) : (x >> 1);
    return x;
#endif
}
uint32_t orc_crc32c_u32_step_bitwise(uint32_t crc, uint32_t data)
{
    uint32_t x = crc ^ data;
    for (int k = 0; k < 32; k++) x = (x & 1) ? (0x82F63B78u ^ (x >> 1)) : (x >> 1);
    return x;
}

/* HashTable.TableEntry  HashTable.cs:91-126: returns the BYTE offset into the ushort table (hash & mask). */
uint32_t orc_hash(uint32_t bytes, uint32_t mask, int variant)
{
    uint32_t hash;
    if (variant == ORC_HASH_CRC32C) hash = crc32c_u32_step(bytes, mask);   /* Sse42.Crc32(bytes, mask)  :109-112 */
    else hash = (0x1e35a7bdu * bytes) >> (31 - 14);                        /* :121-122 */
    return hash & mask;                                                    /* :125 */
}

/* ---------------------------------------------------------------- compressor ---------------------------- */

/* SnappyCompressor.FindMatchLength  SnappyCompressor.cs:562-688.  The *data side effect is an optimisation that
 * never changes results; the restatement returns the length only. */
int orc_find_match_length(const uint8_t* s1, const uint8_t* s2, const uint8_t* s2_limit)
{
    int matched = 0;
    while (s2_limit - s2 >= 8) {                                   /* 64-bit blocks  :644-667 */
        uint64_t a1 = ld64(s1 + matched), a2 = ld64(s2);
        if (a1 == a2) { s2 += 8; matched += 8; }
        else return matched + (__builtin_ctzll(a1 ^ a2) >> 3);
    }
    while (s2 < s2_limit && s1[matched] == *s2) { s2++; matched++; }   /* :669-685 */
    return matched;
}

/* SnappyCompressor.EmitLiteralSlow (+ EmitLiteralFast, identical bytes)  SnappyCompressor.cs:418-464 */
static uint8_t* emit_literal(uint8_t* op, const uint8_t* lit, uint32_t len)
{
    uint32_t n = len - 1;
    if (n < 60) { *op++ = (uint8_t)(n << 2); }
    else {
        int count = (orc_log2_floor(n) >> 3) + 1;                  /* :447 */
        *op++ = (uint8_t)((59 + count) << 2);                      /* :451 */
        for (int i = 0; i < count; i++) op[i] = (uint8_t)(n >> (8 * i));   /* :458-459 (4 bytes written, count kept) */
        op += count;
    }
    memcpy(op, lit, len);
    return op + len;
}

/* SnappyCompressor.EmitCopyAtMost64LenLessThan12 / ...GreaterThanOrEqualTo12  SnappyCompressor.cs:467-505 */
static uint8_t* emit_copy_at_most_64(uint8_t* op, uint32_t offset, uint32_t len)
{
    if (len < 12 && offset < 2048) {                               /* :476-489 */
        *op++ = (uint8_t)(1 | ((len - 4) << 2) | ((offset >> 8) << 5));
        *op++ = (uint8_t)(offset & 0xff);
    } else {                                                       /* :478,502 */
        *op++ = (uint8_t)(2 | ((len - 1) << 2));
        *op++ = (uint8_t)(offset & 0xff);
        *op++ = (uint8_t)(offset >> 8);
    }
    return op;
}

/* SnappyCompressor.EmitCopyLenLessThan12 / EmitCopyLenGreaterThanOrEqualTo12  SnappyCompressor.cs:507-543 */
static uint8_t* emit_copy(uint8_t* op, uint32_t offset, uint32_t len)
{
    if (len < 12) return emit_copy_at_most_64(op, offset, len);
    while (len >= 68) { op = emit_copy_at_most_64(op, offset, 64); len -= 64; }   /* :524-528 */
    if (len > 64) { op = emit_copy_at_most_64(op, offset, 60); len -= 60; }       /* :531-534 */
    return emit_copy_at_most_64(op, offset, len);                                 /* :537-541 */
}

/* SnappyCompressor.CompressFragment  SnappyCompressor.cs:174-415.
 * input <= 65536 bytes, output must hold orc_max_fragment_compressed_length(n), table = table_size zeroed u16. */
size_t orc_compress_fragment(const uint8_t* in, uint32_t n, uint8_t* out, uint16_t* table, uint32_t table_size,
                             int variant)
{
    const uint32_t mask = 2 * (table_size - 1);                    /* :181 */
    uint32_t ip = 0;
    uint8_t* op = out;
#define TABLE_AT(bytes) (table[orc_hash((bytes), mask, variant) >> 1])   /* Unsafe.AddByteOffset  HashTable.cs:125 */

    if (n >= ORC_INPUT_MARGIN) {                                   /* :190 */
        const uint32_t ip_limit = n - ORC_INPUT_MARGIN;            /* :192 */
        for (;;) {
            uint32_t next_emit = ip;                               /* :198 */
            ip++;                                                  /* :199 */
            uint32_t skip = 32;                                    /* :227 */
            uint32_t candidate = 0;
            int found = 0;

            if (ip_limit >= ip && ip_limit - ip >= 16) {           /* :230  (ByteOffset(ip, ipLimit) >= 16) */
                for (uint32_t j = 0; j < 16; j++) {                /* :233-309, unrolled x4 in the reference */
                    uint32_t dword = ld32(in + ip + j);
                    uint16_t* e = &TABLE_AT(dword);
                    candidate = *e;
                    *e = (uint16_t)(ip + j);                       /* :243 */
                    if (ld32(in + candidate) == dword) {           /* :245 */
                        ip += j;
                        found = 1;
                        break;
                    }
                }
                if (!found) { ip += 16; skip += 16; }              /* :311-312 */
            }
            if (!found) {
                for (;;) {                                         /* :315-341 */
                    uint32_t data = ld32(in + ip);
                    uint32_t bytes_between = skip >> 5;            /* :319 */
                    skip += bytes_between;                         /* :320 */
                    uint32_t next_ip = ip + bytes_between;
                    if (next_ip > ip_limit) {                      /* :323-327 */
                        ip = next_emit;
                        goto emit_remainder;
                    }
                    uint16_t* e = &TABLE_AT(data);
                    candidate = *e;                                /* :329 */
                    *e = (uint16_t)ip;                             /* :333 */
                    if (data == ld32(in + candidate)) break;       /* :334 */
                    ip = next_ip;                                  /* :339-340 */
                }
            }

            op = emit_literal(op, in + next_emit, ip - next_emit); /* :347 (and the inlined fast forms :247-250) */

            for (;;) {                                             /* emit_match do-while  :358-398 */
                uint32_t base = ip;
                uint32_t matched = 4 + (uint32_t)orc_find_match_length(in + candidate + 4, in + ip + 4, in + n);
                ip += matched;                                     /* :368-369 */
                op = emit_copy(op, base - candidate, matched);     /* :371-379 */
                if (ip >= ip_limit) goto emit_remainder;           /* :381-384 */
                TABLE_AT(ld32(in + ip - 1)) = (uint16_t)(ip - 1);  /* :393-394 */
                uint32_t data = ld32(in + ip);
                uint16_t* e = &TABLE_AT(data);                     /* :395 */
                candidate = *e;                                    /* :396 */
                *e = (uint16_t)ip;                                 /* :397 */
                if (data != ld32(in + candidate)) break;           /* :398 */
            }
        }
    }
emit_remainder:
    if (ip < n) op = emit_literal(op, in + ip, n - ip);            /* :406-411 */
#undef TABLE_AT
    return (size_t)(op - out);
}

/* SnappyCompressor.TryCompress  SnappyCompressor.cs:24-83 (+ Snappy.TryCompress empty-output check  Snappy.cs:57-62) */
int orc_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int variant, size_t* written)
{
    static _Thread_local uint16_t table[16384];
    static _Thread_local uint8_t scratch[76491 + 16];
    *written = 0;
    if (n > 0xffffffffu) return ORC_ERR_BAD_ARG;
    if (cap == 0) return ORC_ERR_OUTPUT_TOO_SMALL;                 /* Snappy.cs:57-62 */
    if (n && in < out + cap && out < in + n) return ORC_ERR_OVERLAP;   /* :27-30 */
    int hb = orc_varint_write(out, cap, (uint32_t)n);              /* :34-37 */
    if (!hb) return ORC_ERR_OUTPUT_TOO_SMALL;
    size_t w = (size_t)hb;
    while (n > 0) {                                                /* :40-80 */
        uint32_t frag = n < ORC_BLOCK_SIZE ? (uint32_t)n : ORC_BLOCK_SIZE;
        uint32_t ts = orc_table_size(frag);                        /* HashTable.GetHashTable  HashTable.cs:38-55 */
        memset(table, 0, ts * sizeof(uint16_t));                   /* HashTable.cs:52 */
        size_t max_out = (size_t)orc_max_fragment_compressed_length(frag);
        if (cap - w >= max_out) {                                  /* :49-55 */
            w += orc_compress_fragment(in, frag, out + w, table, ts, variant);
        } else {                                                   /* :56-74 */
            size_t got = orc_compress_fragment(in, frag, scratch, table, ts, variant);
            if (cap - w < got) { *written = 0; return ORC_ERR_OUTPUT_TOO_SMALL; }
            memcpy(out + w, scratch, got);
            w += got;
        }
        in += frag;
        n -= frag;
    }
    *written = w;
    return ORC_OK;
}

/* ---------------------------------------------------------------- decompressor -------------------------- */

/* Snappy.TryDecompress  Snappy.cs:172-186 -> SnappyDecompressor.Decompress + DecompressAllTags
 * SnappyDecompressor.cs:43-92,184-347 for one whole block handed over in one piece.
 * Strictness (SURVEY.md 8a "quirks"): the declared length is the hard output bound (the reference checks against
 * its pooled buffer, which can be larger); a declared length >= 2^31 is BAD_LENGTH; cap < declared length is
 * OUTPUT_TOO_SMALL before any tag is looked at. */
int orc_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    uint32_t expected = 0;
    int hb = 0;
    *written = 0;
    int st = orc_varint_read(in, n, &expected, &hb);               /* TryReadUncompressedLength  :110-173 */
    if (st == ORC_ERR_BAD_LENGTH) return ORC_ERR_BAD_LENGTH;       /* :53-56 */
    if (st != ORC_OK) return ORC_ERR_INCOMPLETE;                   /* NeedMoreData -> never AllDataDecompressed  Snappy.cs:178-181 */
    if (expected > 0x7fffffffu) return ORC_ERR_BAD_LENGTH;         /* (int)length goes negative in the reference  :129,160 */
    if (cap < expected) return ORC_ERR_OUTPUT_TOO_SMALL;           /* Snappy.cs:183-185 / ThrowHelper.cs:18-19 */
    size_t ip = (size_t)hb;
    size_t op = 0;
    while (ip < n) {                                               /* :234-341 */
        uint8_t c = in[ip];
        uint32_t tag_extra = (c & 3) == 0 ? ((c >> 2) >= 60 ? (uint32_t)(c >> 2) - 59 : 0)   /* CharTable[c] >> 11  Constants.cs:42-76 */
                             : (c & 3) == 1 ? 1 : (c & 3) == 2 ? 2 : 4;
        if (n - ip < 1 + tag_extra) break;                         /* RefillTag: insufficient -> stop  :464-483 */
        uint32_t trailer = 0;
        for (uint32_t i = 0; i < tag_extra; i++) trailer |= (uint32_t)in[ip + 1 + i] << (8 * i);   /* ExtractLowBytes  Helpers.cs:72-85 */
        ip += 1 + tag_extra;
        if ((c & 3) == 0) {                                        /* literal  :262-302 */
            size_t len = (c >> 2) >= 60 ? (size_t)trailer + 1 : (size_t)(c >> 2) + 1;
            size_t avail = n - ip;
            size_t take = len < avail ? len : avail;               /* :290-297: partial literal, then wait for more input */
            if (take > expected - op) return ORC_ERR_TOO_LONG;     /* Append  :570-573 */
#ifdef ORACLE_FAST
            /* bench.py's cpu_baseline only: what Snappier does with one vector move (SnappyDecompressor.cs:262-288: a literal of <= 16
             * bytes with 16 bytes of slack on both sides is one unaligned 16-byte load + store) */
            if (take <= 16 && n - ip >= 16 && expected - op >= 16) { uint64_t a = ld64(in + ip), b = ld64(in + ip + 8); memcpy(out + op, &a, 8); memcpy(out + op + 8, &b, 8); }
            else
#endif
            memcpy(out + op, in + ip, take);
            op += take;
            ip += take;
            if (take < len) break;
        } else {                                                   /* copies  :305-339 */
            size_t len, off;
            if ((c & 3) == 1) { len = ((c >> 2) & 7) + 4; off = ((size_t)(c >> 5) << 8) | trailer; }
            else { len = (size_t)(c >> 2) + 1; off = trailer; }
            if (off == 0 || op < off) return ORC_ERR_BAD_OFFSET;   /* AppendFromSelf  :598-601 */
            if (len > expected - op) return ORC_ERR_TOO_LONG;      /* :603-606 */
#ifdef ORACLE_FAST
            /* bench.py's cpu_baseline only: CopyHelpers.IncrementalCopy (CopyHelpers.cs:64-230) moves 16 bytes at a time when the
             * offset allows it and the output has slack, 8 at a time from offset 8 on; shorter offsets (pattern copies) stay byte-wise
             * here (the reference expands the pattern with a shuffle first).  Same bytes as the loop below in every case. */
            if (off >= 16 && expected - op >= len + 16) {
                for (size_t k = 0; k < len; k += 16) { uint64_t a = ld64(out + op - off + k), b = ld64(out + op - off + k + 8); memcpy(out + op + k, &a, 8); memcpy(out + op + k + 8, &b, 8); }
            } else if (off >= 8 && expected - op >= len + 8) {
                for (size_t k = 0; k < len; k += 8) { uint64_t a = ld64(out + op - off + k); memcpy(out + op + k, &a, 8); }
            } else
#endif
            for (size_t k = 0; k < len; k++) out[op + k] = out[op - off + k];   /* IncrementalCopySlow  CopyHelpers.cs:222-230 */
            op += len;
        }
    }
    if (op < expected) return ORC_ERR_INCOMPLETE;                  /* !AllDataDecompressed  Snappy.cs:178-181,229-232 */
    *written = op;
    return ORC_OK;
}

/* ---------------------------------------------------------------- framing ------------------------------- */

static const uint8_t k_stream_header[10] = {0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59};   /* SnappyStreamCompressor.cs:18-21 */

int64_t orc_frame_max_encoded_length(int64_t n)
{
    int64_t chunks = (n + ORC_BLOCK_SIZE - 1) / ORC_BLOCK_SIZE;
    return 10 + chunks * 8 + n;   /* a chunk never grows: type 0x01 is used when compression does not shrink it  :212-229 */
}

/* SnappyStreamCompressor.Write(whole buffer) + Flush  SnappyStreamCompressor.cs:40-55,82-97,166-261 */
int orc_frame_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int variant, size_t* written)
{
    static _Thread_local uint8_t body[76496 + 16];
    *written = 0;
    if (cap < (size_t)orc_frame_max_encoded_length((int64_t)n)) return ORC_ERR_OUTPUT_TOO_SMALL;
    size_t w = 0;
    memcpy(out, k_stream_header, 10);                              /* EnsureStreamHeaderWritten  :148-157 */
    w += 10;
    while (n > 0) {
        uint32_t chunk = n < ORC_BLOCK_SIZE ? (uint32_t)n : ORC_BLOCK_SIZE;   /* CompressInput  :166-192 */
        size_t clen = 0;
        int st = orc_compress(in, chunk, body, sizeof(body), variant, &clen);   /* CompressBlock  :206 */
        if (st != ORC_OK) return st;
        uint32_t crc = orc_crc32c_mask(orc_crc32c(in, chunk));     /* :243-245,258-260 */
        uint32_t payload = clen < chunk ? (uint32_t)clen : chunk;  /* :212 */
        uint32_t block_size = payload + 4;
        out[w + 0] = clen < chunk ? 0x00 : 0x01;                   /* :240,255 */
        out[w + 1] = (uint8_t)block_size;                          /* blockSize << 8, little-endian  :237,252 */
        out[w + 2] = (uint8_t)(block_size >> 8);
        out[w + 3] = (uint8_t)(block_size >> 16);
        st32(out + w + 4, crc);
        memcpy(out + w + 8, clen < chunk ? body : in, payload);    /* :216-218 / :225-228 */
        w += 8 + payload;
        in += chunk;
        n -= chunk;
    }
    *written = w;
    return ORC_OK;
}

/* Walk chunk headers only (SnappyStreamDecompressor.ReadChunkHeader  SnappyStreamDecompressor.cs:215-254) and
 * sum decoded sizes: type 0x01 -> size-4, type 0x00 -> varint preamble of the body. */
int orc_frame_decoded_length(const uint8_t* in, size_t n, uint64_t* out_len)
{
    size_t ip = 0;
    uint64_t total = 0;
    *out_len = 0;
    while (ip < n) {
        if (n - ip < 4) return ORC_ERR_TRUNCATED_STREAM;
        uint32_t hdr = ld32(in + ip);
        uint32_t type = hdr & 0xff, size = hdr >> 8;               /* :64-65 */
        ip += 4;
        if (n - ip < size) return ORC_ERR_TRUNCATED_STREAM;
        if (type == 0x00 || type == 0x01) {
            if (size < 4) return ORC_ERR_TRUNCATED_STREAM;
            if (type == 0x01) total += size - 4;
            else {
                uint32_t len = 0, hb = 0;
                if (orc_get_uncompressed_length(in + ip + 4, size - 4, &len, &hb) != ORC_OK) return ORC_ERR_BAD_LENGTH;
                total += len;
            }
        } else if (type < 0x80) return ORC_ERR_CHUNK_TYPE;         /* :182-185 */
        ip += size;
    }
    *out_len = total;
    return ORC_OK;
}

/* SnappyStreamDecompressor.Decompress over a whole framed stream  SnappyStreamDecompressor.cs:38-208 */
int orc_frame_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    size_t ip = 0, op = 0;
    *written = 0;
    while (ip < n) {
        if (n - ip < 4) return ORC_ERR_TRUNCATED_STREAM;
        uint32_t hdr = ld32(in + ip);
        uint32_t type = hdr & 0xff, size = hdr >> 8;               /* :64-65 */
        ip += 4;
        if (n - ip < size) return ORC_ERR_TRUNCATED_STREAM;
        if (type == 0x00 || type == 0x01) {                        /* :71-178 */
            if (size < 4) return ORC_ERR_TRUNCATED_STREAM;
            uint32_t expected_crc = ld32(in + ip);                 /* ReadChunkCrc  :260-289 */
            const uint8_t* body = in + ip + 4;
            size_t body_len = size - 4, got = 0;
            if (type == 0x00) {
                int st = orc_decompress(body, body_len, out + op, cap - op, &got);   /* :90-120 */
                if (st != ORC_OK) return st;
            } else {
                if (cap - op < body_len) return ORC_ERR_OUTPUT_TOO_SMALL;
                memcpy(out + op, body, body_len);                  /* :154-163 */
                got = body_len;
            }
            if (orc_crc32c_mask(orc_crc32c(out + op, got)) != expected_crc) return ORC_ERR_CRC_MISMATCH;   /* :127-131,170-174 */
            op += got;
        } else if (type < 0x80) return ORC_ERR_CHUNK_TYPE;         /* :182-185 */
        /* 0x80..0xff (stream identifier 0xff, padding 0xfe included): skipped unvalidated  :187-196 */
        ip += size;
    }
    *written = op;
    return ORC_OK;
}

/* ---------------------------------------------------------------- batch helpers (cpu_baseline / tests) -- */

/* Independent blocks, same shape as snp_compress_batch / snp_decompress_batch; [first, last) lets the Python
 * side stripe blocks over threads (ctypes releases the GIL). */
void orc_compress_batch(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t first,
                        uint32_t last, uint8_t* out, const uint64_t* out_off, uint32_t* out_len, int32_t* status,
                        int variant)
{
    for (uint32_t b = first; b < last; b++) {
        size_t w = 0;
        if (in_len[b] > ORC_BLOCK_SIZE) { status[b] = ORC_ERR_BAD_ARG; out_len[b] = 0; continue; }
        status[b] = orc_compress(in + in_off[b], in_len[b], out + out_off[b],
                                 (size_t)orc_max_compressed_length(in_len[b]), variant, &w);
        out_len[b] = (uint32_t)w;
    }
}

void orc_decompress_batch(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t first,
                          uint32_t last, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap,
                          uint32_t* out_len, int32_t* status)
{
    for (uint32_t b = first; b < last; b++) {
        size_t w = 0;
        status[b] = orc_decompress(in + in_off[b], in_len[b], out + out_off[b], out_cap[b], &w);
        out_len[b] = (uint32_t)w;
    }
}

void orc_crc32c_batch(const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t first,
                      uint32_t last, int masked, uint32_t* out_crc)
{
    for (uint32_t b = first; b < last; b++) {
        uint32_t c = orc_crc32c(in + in_off[b], in_len[b]);
        out_crc[b] = masked ? orc_crc32c_mask(c) : c;
    }
}

/* bench.py's cpu_baseline only: a FOREIGN block codec (C++ snappy's snappy_compress / snappy_uncompress, found with dlopen by the
 * caller and handed over as a function pointer) run over blocks [first, last) inside one call, so that a thread pool of Python
 * threads scales (one ctypes call per thread instead of one per block).  Returns the number of blocks the function failed on. */
typedef int (*orc_foreign_block_fn)(const char* in, size_t n, char* out, size_t* out_len);
uint64_t orc_foreign_codec_batch(orc_foreign_block_fn fn, const uint8_t* in, const uint64_t* in_off, const uint64_t* in_len, uint64_t first,
                                 uint64_t last, uint8_t* out, const uint64_t* out_off, uint64_t out_cap, uint64_t* out_len)
{
    uint64_t bad = 0;
    for (uint64_t b = first; b < last; b++) {
        size_t w = (size_t)out_cap;
        if (fn((const char*)in + in_off[b], (size_t)in_len[b], (char*)out + out_off[b], &w) != 0) bad++;
        out_len[b] = (uint64_t)w;
    }
    return bad;
}
