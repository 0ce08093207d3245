/*
 * abi_conformance.c -- the drop-in boundary seen from the FOREIGN side: plain C, dlopen, no HIP, no Python.
 * It makes exactly the calls csharp/Snappier.Gpu makes (NativeMethods.cs: symbol names, argument order and widths;
 * Snappy.cs / SnappyStreamChunkCodec.cs: call sequences and the status codes they branch on), so the C# shim's
 * assumptions are tested even though no .NET toolchain exists in the build image.
 *
 *   abi_conformance <libsnappier_hip.so> host      host-only part (no device needed): every symbol resolves, enum values,
 *                                                  length arithmetic, varint KATs, snp_ctx_create -> SNP_ERR_DEVICE or SNP_OK
 *   abi_conformance <libsnappier_hip.so> device <datafile>   the full call sequences on a GPU, round trips compared
 * Exit code 0 = conforming; every failed expectation prints a line and counts.
 * TEST INFRASTRUCTURE.
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/snappier_hip.h"

static int g_fail = 0;
#define EXPECT(cond)                                                                     \
    do {                                                                                 \
        if (!(cond)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)

/* the values csharp/Snappier.Gpu/NativeMethods.cs hard-codes (enum SnpStatus / SnpHash, constants) */
_Static_assert(SNP_OK == 0 && SNP_ERR_OUTPUT_TOO_SMALL == 1 && SNP_ERR_BAD_OFFSET == 2 && SNP_ERR_TOO_LONG == 3 &&
               SNP_ERR_INCOMPLETE == 4 && SNP_ERR_BAD_LENGTH == 5 && SNP_ERR_CRC_MISMATCH == 6 && SNP_ERR_CHUNK_TYPE == 7 &&
               SNP_ERR_OVERLAP == 8 && SNP_ERR_BAD_ARG == 9 && SNP_ERR_DEVICE == 10 && SNP_ERR_TRUNCATED_STREAM == 11,
               "SnpStatus values");
_Static_assert(SNP_HASH_CRC32C == 0 && SNP_HASH_MUL == 1, "SnpHash values");
_Static_assert(SNP_BLOCK_SIZE == 65536 && SNP_MAX_BLOCK_COMPRESSED == 76491 && SNP_VARINT_MAX == 5 &&
               SNP_STREAM_HEADER_LEN == 10 && SNP_CHUNK_HEADER_LEN == 8, "constants");
_Static_assert(sizeof(snp_status) == 4 && sizeof(size_t) == 8 && sizeof(void*) == 8, "int32 status, 64-bit size_t (nuint)");

/* one function-pointer type per DllImport, spelled with the C types the C# marshaller produces */
typedef int32_t (*fn_ctx_create)(int32_t, int32_t, void*, void**);
typedef void (*fn_ctx_destroy)(void*);
typedef int32_t (*fn_ctx_set_stream)(void*, void*);
typedef const char* (*fn_ctx_last_error)(const void*);
typedef int32_t (*fn_ctx_synchronize)(void*);
typedef uint64_t (*fn_ctx_counter)(const void*, int32_t);
typedef int32_t (*fn_ctx_set_option)(void*, int32_t, int64_t);
typedef int32_t (*fn_ctx_get_option)(const void*, int32_t, int64_t*);
typedef int32_t (*fn_ctx_reserve)(void*, uint32_t);
typedef const char* (*fn_status_string)(int32_t);
typedef const char* (*fn_version)(void);
typedef int64_t (*fn_len)(int64_t);
typedef int32_t (*fn_get_ulen)(const uint8_t*, size_t, uint32_t*, uint32_t*);
typedef int32_t (*fn_buf)(void*, const uint8_t*, size_t, uint8_t*, size_t, size_t*);
typedef int32_t (*fn_segs)(void*, const uint8_t* const*, const size_t*, uint32_t, uint8_t*, size_t, size_t*);
typedef int32_t (*fn_crc)(void*, const uint8_t*, size_t, int32_t, uint32_t*);
typedef int32_t (*fn_frame_len)(const uint8_t*, size_t, uint64_t*);

/* ---- several callers at once: Snappy.* is re-entrant (Snappy.cs:64,174,225 news up a compressor per call), so the shim keeps one
 * context per calling thread (GpuContext, [ThreadStatic]) -- and one per DEVICE when the host has several (the multi-GPU path below
 * Python: a C# service on an 8-GPU node opens eight contexts in one process).  Each thread: its own context on device
 * `device`, round trips of its own slice of the data, results compared with what the main thread's context produced. */
typedef struct {
    fn_ctx_create ctx_create; fn_ctx_destroy ctx_destroy; fn_buf try_compress, try_decompress; fn_len max_len;
    int device; const uint8_t* data; size_t n; const uint8_t* want; size_t want_len; int rounds; int fails; int created;
} worker_arg;

static void* worker(void* p)
{
    worker_arg* a = (worker_arg*)p;
    void* ctx = NULL;
    if (a->ctx_create(a->device, SNP_HASH_CRC32C, NULL, &ctx) != SNP_OK || !ctx) { a->created = 0; return NULL; }
    a->created = 1;
    const size_t cap = (size_t)a->max_len((int64_t)a->n);
    uint8_t* comp = malloc(cap);
    uint8_t* back = malloc(a->n + 16);
    for (int r = 0; r < a->rounds; ++r) {
        size_t w = 0, w2 = 0;
        if (a->try_compress(ctx, a->data, a->n, comp, cap, &w) != SNP_OK || w != a->want_len || memcmp(comp, a->want, w) != 0) ++a->fails;
        if (a->try_decompress(ctx, comp, w, back, a->n, &w2) != SNP_OK || w2 != a->n || memcmp(back, a->data, a->n) != 0) ++a->fails;
    }
    a->ctx_destroy(ctx);
    free(comp); free(back);
    return NULL;
}

/* ---- a device LIST behind one caller (csharp/Snappier.Gpu/MultiDeviceChunkCodec.cs, snappier_amd/multidevice.py): the run of chunks is cut into
 * contiguous whole-chunk ranges, each range is framed by its own context on its own thread, the host concatenates (stream identifier once). */
typedef struct {
    fn_ctx_create ctx_create; fn_ctx_destroy ctx_destroy; fn_buf frame_encode; fn_len frame_max;
    int device; const uint8_t* piece; size_t n; uint8_t* out; size_t written; int ok;
} range_arg;

static void* range_worker(void* p)
{
    range_arg* a = (range_arg*)p;
    void* ctx = NULL;
    a->ok = 0;
    if (a->ctx_create(a->device, SNP_HASH_CRC32C, NULL, &ctx) != SNP_OK || !ctx) return NULL;
    const size_t cap = (size_t)a->frame_max((int64_t)a->n);
    a->out = malloc(cap);
    a->ok = a->frame_encode(ctx, a->piece, a->n, a->out, cap, &a->written) == SNP_OK;
    a->ctx_destroy(ctx);
    return NULL;
}

static void* must(void* lib, const char* name)
{
    void* p = dlsym(lib, name);
    if (!p) { printf("FAIL missing symbol %s\n", name); ++g_fail; }
    return p;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <lib> host|device [datafile]\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { printf("FAIL dlopen: %s\n", dlerror()); return 1; }
    const int device = strcmp(argv[2], "device") == 0;

    /* every DllImport of NativeMethods.cs must resolve */
    static const char* all[] = {
        "snp_ctx_create", "snp_ctx_destroy", "snp_ctx_set_stream", "snp_ctx_last_error", "snp_ctx_synchronize", "snp_ctx_counter",
        "snp_ctx_set_option", "snp_ctx_get_option", "snp_ctx_reserve_compress", "snp_status_string", "snp_version", "snp_max_compressed_length", "snp_max_fragment_compressed_length",
        "snp_get_uncompressed_length", "snp_try_compress", "snp_try_decompress", "snp_try_compress_segments", "snp_try_decompress_segments", "snp_crc32c", "snp_frame_max_encoded_length",
        "snp_frame_encode", "snp_frame_decoded_length", "snp_frame_decode", "snp_compress_batch", "snp_decompress_batch",
        "snp_crc32c_batch", "snp_concat_batch", "snp_frame_encode_workspace", "snp_frame_encode_device",
        "snp_frame_decode_chunks_device", "snp_frame_decode_workspace", "snp_frame_decode_device"};
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) (void)must(lib, all[i]);
    if (g_fail) return 1;

    fn_ctx_create ctx_create = (fn_ctx_create)must(lib, "snp_ctx_create");
    fn_ctx_destroy ctx_destroy = (fn_ctx_destroy)must(lib, "snp_ctx_destroy");
    fn_ctx_last_error last_error = (fn_ctx_last_error)must(lib, "snp_ctx_last_error");
    fn_ctx_counter counter = (fn_ctx_counter)must(lib, "snp_ctx_counter");
    fn_ctx_set_option set_option = (fn_ctx_set_option)must(lib, "snp_ctx_set_option");
    fn_ctx_get_option get_option = (fn_ctx_get_option)must(lib, "snp_ctx_get_option");
    fn_status_string status_string = (fn_status_string)must(lib, "snp_status_string");
    fn_version version = (fn_version)must(lib, "snp_version");
    fn_len max_len = (fn_len)must(lib, "snp_max_compressed_length");
    fn_len max_frag = (fn_len)must(lib, "snp_max_fragment_compressed_length");
    fn_len frame_max = (fn_len)must(lib, "snp_frame_max_encoded_length");
    fn_get_ulen get_ulen = (fn_get_ulen)must(lib, "snp_get_uncompressed_length");
    fn_buf try_compress = (fn_buf)must(lib, "snp_try_compress");
    fn_buf try_decompress = (fn_buf)must(lib, "snp_try_decompress");
    fn_segs compress_segments = (fn_segs)must(lib, "snp_try_compress_segments");
    fn_segs decompress_segments = (fn_segs)must(lib, "snp_try_decompress_segments");
    fn_buf frame_encode = (fn_buf)must(lib, "snp_frame_encode");
    fn_buf frame_decode = (fn_buf)must(lib, "snp_frame_decode");
    fn_crc crc32c = (fn_crc)must(lib, "snp_crc32c");
    fn_frame_len frame_decoded_length = (fn_frame_len)must(lib, "snp_frame_decoded_length");

    /* ---- host-only: what Snappy.GetMaxCompressedLength / GetUncompressedLength / ThrowIfFailed rely on -------------- */
    EXPECT(max_len(65536) == 76496 && max_frag(65536) == 76491 && max_len(0) == 38 && max_len(-1) == -1);
    EXPECT(frame_max(0) == 10 && frame_max(65537) == 10 + 16 + 65537);
    EXPECT(strcmp(status_string(SNP_ERR_OUTPUT_TOO_SMALL), "Output buffer is too small.") == 0);
    EXPECT(strcmp(status_string(SNP_ERR_BAD_OFFSET), "Invalid copy offset") == 0);
    EXPECT(strcmp(status_string(SNP_ERR_TOO_LONG), "Data too long") == 0);
    EXPECT(strcmp(status_string(SNP_ERR_INCOMPLETE), "Incomplete Snappy block.") == 0);
    EXPECT(strcmp(status_string(SNP_ERR_BAD_LENGTH), "Invalid stream length") == 0);
    EXPECT(strcmp(status_string(SNP_ERR_CRC_MISMATCH), "Chunk CRC mismatch.") == 0);
    EXPECT(strcmp(status_string(SNP_ERR_OVERLAP), "Input and output spans must not overlap.") == 0);
    EXPECT(version() && strstr(version(), "snappier_hip") != NULL);
    {
        const uint8_t v1[] = {0x80, 0x80, 0x04, 0xff};                   /* 65536 */
        const uint8_t bad[] = {0xff, 0xff, 0xff, 0xff, 0x7f, 0x00};      /* overflow in the 5th byte */
        const uint8_t cut[] = {0x80, 0x80};
        uint32_t n = 0, hb = 0;
        EXPECT(get_ulen(v1, sizeof v1, &n, &hb) == SNP_OK && n == 65536 && hb == 3);
        EXPECT(get_ulen(bad, sizeof bad, &n, &hb) == SNP_ERR_BAD_LENGTH);
        EXPECT(get_ulen(cut, sizeof cut, &n, &hb) == SNP_ERR_BAD_LENGTH);
        EXPECT(get_ulen(v1, 0, &n, &hb) == SNP_ERR_BAD_LENGTH);
        uint64_t total = 1;
        const uint8_t hdr_only[] = {0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59};
        EXPECT(frame_decoded_length(hdr_only, sizeof hdr_only, &total) == SNP_OK && total == 0);
    }
    {
        int64_t v = 0;                                                   /* GpuContext.SetOption / GetOption on a dead handle */
        EXPECT(set_option(NULL, SNP_OPT_DECODE_LAYOUT, 1) == SNP_ERR_BAD_ARG && get_option(NULL, SNP_OPT_DECODE_LAYOUT, &v) == SNP_ERR_BAD_ARG);
    }
    void* ctx = NULL;
    EXPECT(ctx_create(0, 7, NULL, &ctx) == SNP_ERR_BAD_ARG && ctx == NULL);      /* unknown hash variant */
    const int32_t st_create = ctx_create(0, SNP_HASH_CRC32C, NULL, &ctx);
    if (!device) {
        /* GpuContext.Create: Ok -> usable, Device -> IsAvailable == false (managed fallback); nothing else is acceptable */
        EXPECT(st_create == SNP_OK || (st_create == SNP_ERR_DEVICE && ctx == NULL));
        if (ctx) ctx_destroy(ctx);
        printf("%s (host part, ctx_create -> %d)\n", g_fail ? "NOT CONFORMING" : "conforming", st_create);
        return g_fail ? 1 : 0;
    }
    EXPECT(st_create == SNP_OK && ctx != NULL);
    if (!ctx) return 1;
    EXPECT(last_error(ctx) != NULL);
    {
        /* GpuContext.SetOption: every option round-trips, out-of-range values and unknown options change nothing */
        int64_t v = -1;
        EXPECT(get_option(ctx, SNP_OPT_DECODE_LAYOUT, &v) == SNP_OK && v == 0);
        EXPECT(set_option(ctx, SNP_OPT_DECODE_LAYOUT, 4) == SNP_OK && get_option(ctx, SNP_OPT_DECODE_LAYOUT, &v) == SNP_OK && v == 4);
        EXPECT(set_option(ctx, SNP_OPT_DECODE_LAYOUT, 7) == SNP_ERR_BAD_ARG && get_option(ctx, SNP_OPT_DECODE_LAYOUT, &v) == SNP_OK && v == 4);
        EXPECT(set_option(ctx, SNP_OPT_DECODE_LAYOUT, 6) == SNP_OK && get_option(ctx, SNP_OPT_DECODE_LAYOUT, &v) == SNP_OK && v == 6);   /* the serial kernel */
        EXPECT(set_option(ctx, SNP_OPT_DECODE_LAYOUT, 0) == SNP_OK && get_option(ctx, SNP_OPT_DECODE_LAYOUT, &v) == SNP_OK && v == 0);
        /* the launch-shape options of round 6: defaults, round trips, ranges; the deprecated name of option 11 still compiles */
        EXPECT(get_option(ctx, SNP_OPT_COMPRESS_LANE_STORES, &v) == SNP_OK && v == -1 && get_option(ctx, SNP_OPT_COMPRESS_SMALL_INPUT_LDS, &v) == SNP_OK && v == -1);
        EXPECT(set_option(ctx, SNP_OPT_COMPRESS_LANE_STORES, 256) == SNP_ERR_BAD_ARG && set_option(ctx, SNP_OPT_COMPRESS_LANE_STORES, 87) == SNP_OK &&
               get_option(ctx, SNP_OPT_COMPRESS_LANE_STORES, &v) == SNP_OK && v == 87 && set_option(ctx, SNP_OPT_COMPRESS_LANE_STORES, -1) == SNP_OK);
        EXPECT(set_option(ctx, SNP_OPT_COMPRESS_LANE_PROBES, 5) == SNP_ERR_BAD_ARG && set_option(ctx, SNP_OPT_COMPRESS_LANES_PER_WAVEFRONT, 24) == SNP_ERR_BAD_ARG &&
               set_option(ctx, SNP_OPT_COMPRESS_SLICE, 100) == SNP_ERR_BAD_ARG && set_option(ctx, SNP_OPT_COMPRESS_WINDOW_POSITIONS, 3) == SNP_ERR_BAD_ARG &&
               set_option(ctx, SNP_OPT_FRAME_SCAN, 2) == SNP_ERR_BAD_ARG && set_option(ctx, SNP_OPT_DECODE_LDS_THROTTLE, -1) == SNP_ERR_BAD_ARG);
        EXPECT(get_option(ctx, SNP_OPT_COMPRESS_SLICE, &v) == SNP_OK && v == 262144 && get_option(ctx, SNP_OPT_COMPRESS_WINDOW_GLOBAL_MIN_BATCH, &v) == SNP_OK && v == 4096);
        EXPECT(set_option(ctx, SNP_OPT_COMPRESS_LAYOUT, 5) == SNP_OK && get_option(ctx, SNP_OPT_COMPRESS_LAYOUT, &v) == SNP_OK && v == 5 &&
               set_option(ctx, SNP_OPT_COMPRESS_LAYOUT, 6) == SNP_ERR_BAD_ARG && set_option(ctx, SNP_OPT_COMPRESS_LAYOUT, 1) == SNP_ERR_BAD_ARG && set_option(ctx, SNP_OPT_COMPRESS_LAYOUT, 0) == SNP_OK);
        EXPECT(get_option(ctx, SNP_OPT_COMPRESS_WINDOW_DUAL_MIN_BATCH, &v) == SNP_OK && v == 1536 && get_option(ctx, SNP_OPT_COMPRESS_WINDOW_GLOBAL_SLOTS, &v) == SNP_OK && v == 0 &&
               get_option(ctx, SNP_OPT_COMPRESS_WINDOW_MAX_BATCH, &v) == SNP_OK && v == 32768 && set_option(ctx, SNP_OPT_COMPRESS_WINDOW_GLOBAL_SLOTS, 70000) == SNP_ERR_BAD_ARG);
        EXPECT(SNP_OPT_CRC_TABLE_FREE == SNP_OPT_CRC_KERNEL && set_option(ctx, SNP_OPT_CRC_TABLE_FREE, 1) == SNP_OK && get_option(ctx, SNP_OPT_CRC_KERNEL, &v) == SNP_OK && v == 1 &&
               set_option(ctx, SNP_OPT_CRC_KERNEL, 0) == SNP_OK);
        EXPECT(set_option(ctx, SNP_OPT_TABLE_PROBE_MAX_BYTES, (int64_t)32 << 30) == SNP_OK && get_option(ctx, SNP_OPT_TABLE_PROBE_MAX_BYTES, &v) == SNP_OK && v == ((int64_t)32 << 30));
        EXPECT(set_option(ctx, SNP_OPT_TABLE_PROBE_TRIES, 0) == SNP_ERR_BAD_ARG && set_option(ctx, SNP_OPT_TABLE_PROBE_TRIES, 3) == SNP_OK);
        EXPECT(get_option(ctx, SNP_OPT_SMALL_BLOCK_MAX, &v) == SNP_OK && v == 512);
        EXPECT(set_option(ctx, 999, 1) == SNP_ERR_BAD_ARG && get_option(ctx, 999, &v) == SNP_ERR_BAD_ARG && get_option(ctx, SNP_OPT_FENCED, NULL) == SNP_ERR_BAD_ARG);
    }
    {
        /* GpuContext.ReserveCompress: nothing to do for 0 fragments, a small workspace for a few, a null context is refused */
        fn_ctx_reserve reserve = (fn_ctx_reserve)must(lib, "snp_ctx_reserve_compress");
        EXPECT(reserve(NULL, 16) == SNP_ERR_BAD_ARG);
        EXPECT(reserve(ctx, 0) == SNP_OK && reserve(ctx, 64) == SNP_OK && reserve(ctx, 32) == SNP_OK);
    }

    /* ---- device: the sequences of Snappy.cs ------------------------------------------------------------------ */
    FILE* f = argc > 3 ? fopen(argv[3], "rb") : NULL;
    if (!f) { printf("FAIL cannot open data file\n"); return 1; }
    static uint8_t data[300000];
    const size_t n = fread(data, 1, sizeof data, f);
    fclose(f);
    EXPECT(n > 70000);
    const size_t cap = (size_t)max_len((int64_t)n);
    uint8_t* comp = malloc(cap);
    uint8_t* back = malloc(n + 16);
    size_t w = 123, w2 = 0;
    /* TryCompress: empty / too-small output -> false, bytesWritten 0 */
    EXPECT(try_compress(ctx, data, n, comp, 0, &w) == SNP_ERR_OUTPUT_TOO_SMALL && w == 0);
    EXPECT(try_compress(ctx, data, n, comp, 10, &w) == SNP_ERR_OUTPUT_TOO_SMALL && w == 0);
    /* overlap -> InvalidOperationException */
    EXPECT(try_compress(ctx, data, 1000, data + 500, 2000, &w) == SNP_ERR_OVERLAP);
    /* Compress into GetMaxCompressedLength, then into exactly the compressed length (SnappyTests.cs:41-63) */
    EXPECT(try_compress(ctx, data, n, comp, cap, &w) == SNP_OK && w > 3 && w < n);
    uint8_t* exact = malloc(w);
    EXPECT(try_compress(ctx, data, n, exact, w, &w2) == SNP_OK && w2 == w && memcmp(exact, comp, w) == 0);
    EXPECT(try_compress(ctx, data, n, exact, w - 1, &w2) == SNP_ERR_OUTPUT_TOO_SMALL && w2 == 0);
    /* DecompressToMemory: GetUncompressedLength sizes the buffer, TryDecompress fills it */
    uint32_t ulen = 0, hb = 0;
    EXPECT(get_ulen(comp, w, &ulen, &hb) == SNP_OK && ulen == n);
    EXPECT(try_decompress(ctx, comp, w, back, n, &w2) == SNP_OK && w2 == n && memcmp(back, data, n) == 0);
    EXPECT(try_decompress(ctx, comp, w, back, n - 1, &w2) == SNP_ERR_OUTPUT_TOO_SMALL && w2 == 0);
    EXPECT(counter(ctx, 0) + counter(ctx, 1) >= 1);                      /* n >= 256 KiB took the per-fragment path or its fallback */
    /* corrupt data -> InvalidDataException family */
    EXPECT(try_decompress(ctx, comp, w / 2, back, n, &w2) == SNP_ERR_INCOMPLETE);
    {
        uint8_t junk[8] = {0x08, 0x0c, 'a', 'b', 'c', 'd', 0x05, 0x09};  /* declared 8: literal "abcd" then a copy with offset 9 > 4 */
        EXPECT(try_decompress(ctx, junk, sizeof junk, back, 8, &w2) == SNP_ERR_BAD_OFFSET);
    }
    /* Snappy.Compress / DecompressToMemory(ReadOnlySequence): the same input as three (two) pinned segments -> the same bytes */
    {
        const uint8_t* seg[3] = {data, data + 70001, data + 70001 + 3};
        const size_t seg_len[3] = {70001, 3, n - 70004};
        uint8_t* comp2 = malloc(cap);
        size_t ws = 0, wd = 0;
        EXPECT(try_compress(ctx, data, n, comp, cap, &w) == SNP_OK);
        EXPECT(compress_segments(ctx, seg, seg_len, 3, comp2, cap, &ws) == SNP_OK && ws == w && memcmp(comp2, comp, w) == 0);
        EXPECT(compress_segments(ctx, seg, seg_len, 3, comp2, 10, &ws) == SNP_ERR_OUTPUT_TOO_SMALL && ws == 0);
        EXPECT(compress_segments(ctx, seg, seg_len, 0, comp2, cap, &ws) == SNP_OK && ws == 1 && comp2[0] == 0);   /* the empty sequence */
        const uint8_t* cseg[2] = {comp, comp + 2};                       /* the varint preamble itself split over two segments */
        const size_t cseg_len[2] = {2, w - 2};
        EXPECT(decompress_segments(ctx, cseg, cseg_len, 2, back, n, &wd) == SNP_OK && wd == n && memcmp(back, data, n) == 0);
        EXPECT(decompress_segments(ctx, cseg, cseg_len, 2, back, n - 1, &wd) == SNP_ERR_OUTPUT_TOO_SMALL && wd == 0);
        const size_t cut_len[2] = {2, w / 2};
        EXPECT(decompress_segments(ctx, cseg, cut_len, 2, back, n, &wd) == SNP_ERR_INCOMPLETE);
        EXPECT(compress_segments(ctx, NULL, NULL, 2, comp2, cap, &ws) == SNP_ERR_BAD_ARG);
        free(comp2);
    }
    /* one small buffer (a single fragment, the Snappy.CompressToArray("hello") case) */
    EXPECT(try_compress(ctx, (const uint8_t*)"hello hello hello hello", 23, comp, cap, &w) == SNP_OK);
    EXPECT(try_decompress(ctx, comp, w, back, 23, &w2) == SNP_OK && w2 == 23 && memcmp(back, "hello hello hello hello", 23) == 0);
    /* empty input: one preamble byte */
    EXPECT(try_compress(ctx, data, 0, comp, cap, &w) == SNP_OK && w == 1 && comp[0] == 0);
    EXPECT(try_decompress(ctx, comp, 1, back, 0 + 1, &w2) == SNP_OK && w2 == 0);

    /* ---- device: the sequences of SnappyStreamChunkCodec.cs ----------------------------------------------------- */
    const size_t fcap = (size_t)frame_max((int64_t)n);
    uint8_t* framed = malloc(fcap);
    EXPECT(frame_encode(ctx, data, n, framed, fcap, &w) == SNP_OK && w > 18 && memcmp(framed, "\xff\x06\x00\x00sNaPpY", 10) == 0);
    uint64_t total = 0;
    EXPECT(frame_decoded_length(framed, w, &total) == SNP_OK && total == n);
    EXPECT(frame_decode(ctx, framed, w, back, n, &w2) == SNP_OK && w2 == n && memcmp(back, data, n) == 0);
    EXPECT(frame_decode(ctx, framed + 10, w - 10, back, n, &w2) == SNP_OK && w2 == n);   /* a run of chunks without the identifier */
    framed[14] ^= 1;                                                     /* first chunk's CRC */
    EXPECT(frame_decode(ctx, framed, w, back, n, &w2) == SNP_ERR_CRC_MISMATCH);
    framed[14] ^= 1;
    framed[10] = 0x02;                                                   /* reserved unskippable chunk type */
    EXPECT(frame_decode(ctx, framed, w, back, n, &w2) == SNP_ERR_CHUNK_TYPE);
    framed[10] = 0x00;
    uint32_t crc = 0;
    EXPECT(crc32c(ctx, (const uint8_t*)"123456789", 9, 0, &crc) == SNP_OK && crc == 0xE3069283u);   /* Crc32CAlgorithmTests.cs */

    /* ---- several contexts in one process: four threads, contexts spread over every device the host has ------------- */
    {
        int ndev = 1;
        for (int d = 1; d < 16; ++d) {                                   /* how many devices?  (no HIP here: ask snp_ctx_create) */
            void* probe = NULL;
            if (ctx_create(d, SNP_HASH_CRC32C, NULL, &probe) != SNP_OK) break;
            ctx_destroy(probe);
            ndev = d + 1;
        }
        size_t wl = 0;
        EXPECT(try_compress(ctx, data, n, comp, cap, &wl) == SNP_OK);
        enum { kThreads = 4 };
        pthread_t th[kThreads];
        worker_arg args[kThreads];
        for (int t = 0; t < kThreads; ++t) {
            args[t] = (worker_arg){ctx_create, ctx_destroy, try_compress, try_decompress, max_len, t % ndev, data, n, comp, wl, 6, 0, 0};
            EXPECT(pthread_create(&th[t], NULL, worker, &args[t]) == 0);
        }
        for (int t = 0; t < kThreads; ++t) {
            pthread_join(th[t], NULL);
            EXPECT(args[t].created == 1 && args[t].fails == 0);
        }
        printf("%d threads x 6 round trips on %d device(s): %s\n", kThreads, ndev, g_fail ? "FAIL" : "ok");
    }
    /* ---- the device-list sequence: devices = {0, 0, ... one entry per device, at least two}: ranges framed on separate contexts and threads,
     * concatenated on the host, must be the bytes ONE context produces for the whole run, and decode back through one context ------------ */
    {
        enum { kRanges = 3 };
        const size_t chunks = (n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE;
        size_t wl = 0;
        EXPECT(frame_encode(ctx, data, n, framed, fcap, &wl) == SNP_OK);                /* the single-context bytes */
        pthread_t th[kRanges];
        range_arg args[kRanges];
        size_t first = 0;
        int used = 0;
        for (int r = 0; r < kRanges && first < chunks; ++r, ++used) {
            size_t cnt = chunks / kRanges + ((size_t)r < chunks % kRanges ? 1 : 0);
            if (cnt == 0) cnt = 1;
            if (first + cnt > chunks || r == kRanges - 1) cnt = chunks - first;
            const size_t lo = first * SNP_BLOCK_SIZE, hi = (first + cnt) * SNP_BLOCK_SIZE < n ? (first + cnt) * SNP_BLOCK_SIZE : n;
            args[r] = (range_arg){ctx_create, ctx_destroy, frame_encode, frame_max, 0, data + lo, hi - lo, NULL, 0, 0};
            EXPECT(pthread_create(&th[r], NULL, range_worker, &args[r]) == 0);
            first += cnt;
        }
        uint8_t* joined = malloc(fcap + 64);
        size_t at = 0;
        for (int r = 0; r < used; ++r) {
            pthread_join(th[r], NULL);
            EXPECT(args[r].ok);
            if (!args[r].ok) continue;
            const size_t skip = r ? SNP_STREAM_HEADER_LEN : 0;                             /* the identifier is written once */
            memcpy(joined + at, args[r].out + skip, args[r].written - skip);
            at += args[r].written - skip;
            free(args[r].out);
        }
        EXPECT(at == wl && memcmp(joined, framed, wl) == 0);
        size_t w2 = 0;
        EXPECT(frame_decode(ctx, joined, at, back, n, &w2) == SNP_OK && w2 == n && memcmp(back, data, n) == 0);
        printf("device list: %d ranges of whole chunks on their own contexts, concatenated: %s\n", used, g_fail ? "FAIL" : "same bytes as one context");
        free(joined);
    }
    ctx_destroy(ctx);
    free(comp); free(back); free(exact); free(framed);
    printf("%s (device part)\n", g_fail ? "NOT CONFORMING" : "conforming");
    return g_fail ? 1 : 0;
}
