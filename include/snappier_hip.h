/*
 * snappier_hip.h -- C-ABI of libsnappier_hip.so, the MI355X (gfx950) Snappy block codec.
 *
 * This is the drop-in boundary.  Snappier itself has no FFI seam (it is pure C#,
 * Snappier/Snappier.csproj:19-22); the entry points below are what a C# shim would P/Invoke to replace the
 * bodies of the reference methods cited on each declaration.  Plain pointers and sizes only; no exceptions
 * cross the boundary -- every call returns an snp_status that maps 1:1 onto the reference's outcome
 * (Snappier/Internal/ThrowHelper.cs:8-36).
 *
 * Memory spaces: entry points suffixed _batch / _device take DEVICE pointers (hipMalloc'd memory, e.g. a torch
 * tensor's data_ptr()) and enqueue on the context's stream; all other entry points take HOST pointers, stage
 * through context-owned HBM scratch, and block until the result is in the caller's buffer (the reference's
 * Span API is synchronous, Snappy.cs:37,153).  There is no CPU fallback: without a HIP device
 * snp_ctx_create fails with SNP_ERR_DEVICE.
 */
#ifndef SNAPPIER_HIP_H
#define SNAPPIER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (reference outcome each one stands for) --------------------------------------------- */
typedef enum snp_status {
    SNP_OK = 0,
    /* Try* returned false / ArgumentException("Output buffer is too small.")  ThrowHelper.cs:18-19 */
    SNP_ERR_OUTPUT_TOO_SMALL = 1,
    /* InvalidDataException("Invalid copy offset")  SnappyDecompressor.cs:598-601 */
    SNP_ERR_BAD_OFFSET = 2,
    /* InvalidDataException("Data too long")  SnappyDecompressor.cs:570-573,603-606 */
    SNP_ERR_TOO_LONG = 3,
    /* InvalidDataException("Incomplete Snappy block.")  Snappy.cs:178-181,229-232 */
    SNP_ERR_INCOMPLETE = 4,
    /* InvalidDataException("Invalid stream length")  VarIntEncoding.Read.cs:18-21 */
    SNP_ERR_BAD_LENGTH = 5,
    /* InvalidDataException("Chunk CRC mismatch.")  SnappyStreamDecompressor.cs:127-131,170-174 */
    SNP_ERR_CRC_MISMATCH = 6,
    /* InvalidDataException("Unknown chunk type ..")  SnappyStreamDecompressor.cs:182-185 */
    SNP_ERR_CHUNK_TYPE = 7,
    /* InvalidOperationException("Input and output spans must not overlap.")  SnappyCompressor.cs:27-30 */
    SNP_ERR_OVERLAP = 8,
    /* ArgumentNullException / ArgumentException on the managed side */
    SNP_ERR_BAD_ARG = 9,
    /* no usable HIP device, or a HIP runtime call failed (no reference analogue; surfaces as InvalidOperation) */
    SNP_ERR_DEVICE = 10,
    /* framed stream ends inside a chunk header or chunk body (reference: the Stream simply returns 0 bytes) */
    SNP_ERR_TRUNCATED_STREAM = 11
} snp_status;

/* Which TableEntry hash the compressor reproduces (Snappier/Internal/HashTable.cs:91-126). */
typedef enum snp_hash_variant {
    SNP_HASH_CRC32C = 0, /* x64 SSE4.2 / ARM CRC path, .NET 8+ (HashTable.cs:109-117) -- Snappier's default on the GPU box host */
    SNP_HASH_MUL = 1     /* (0x1e35a7bd * bytes) >> 17 fallback (HashTable.cs:121-122) -- what the golden .snappy fixtures were made with */
} snp_hash_variant;

enum {
    SNP_BLOCK_SIZE = 65536,               /* Constants.cs:25-26 */
    SNP_MAX_BLOCK_COMPRESSED = 76491,     /* Helpers.cs:49  (32 + n + n/6 + 1 at n = 65536) */
    SNP_VARINT_MAX = 5,                   /* VarIntEncoding.MaxLength */
    SNP_STREAM_HEADER_LEN = 10,           /* SnappyStreamCompressor.cs:18-21 */
    SNP_CHUNK_HEADER_LEN = 8              /* SnappyStreamCompressor.cs:199 */
};

typedef struct snp_ctx snp_ctx;

/* ---- context ------------------------------------------------------------------------------------------ */

/* One context = one HIP device + one stream + reusable HBM scratch.  Snappy.* is re-entrant because it news up
 * a compressor per call (Snappy.cs:64,174,225); the equivalent here is one ctx per calling thread.
 * stream == NULL: the context creates and owns a non-blocking stream; otherwise it enqueues on the given hipStream_t
 * (see snp_ctx_set_stream to bind the legacy default stream). */
snp_status snp_ctx_create(int device, int hash_variant, void* stream, snp_ctx** out_ctx);
void snp_ctx_destroy(snp_ctx* ctx);
/* Enqueue all further work of this context on `stream` (a hipStream_t; NULL = the legacy default stream, which is
 * what torch.cuda.current_stream().cuda_stream is for torch's default stream).  Releases a context-owned stream. */
snp_status snp_ctx_set_stream(snp_ctx* ctx, void* stream);
/* Last HIP error string seen by this context ("" if none); pointer valid until the next call on ctx. */
const char* snp_ctx_last_error(const snp_ctx* ctx);
/* Block until everything enqueued on the context's stream is done (for the *_batch entry points). */
snp_status snp_ctx_synchronize(snp_ctx* ctx);
/* Introspection for tests and tuning: how often this context took a code path since it was created.
 * which: 0 = large single blocks decoded one wavefront per 64 KiB fragment (snp_try_decompress, tag index),
 *        1 = large single blocks that fell back to the single-wavefront decoder (foreign / malformed streams),
 *        2 = microseconds the chosen hash-table workspace took in the placement probe (512 table-walk probes per fragment; 0: no search ran),
 *        3 = candidate pieces the workspace search allocated,
 *        4 = microseconds the whole search took (wall clock),  5 = most bytes it held at once (every candidate coexists until it ends),
 *        6 = large single blocks whose tag index took the look-back pass (hardly compressed or irregular streams, or more incompressible regions
 *            than it pays to resolve one by one; tag_index.hip). */
uint64_t snp_ctx_counter(const snp_ctx* ctx, int which);

/* Per-context configuration.  A library loaded into a long-running service is configured through these, per context and at any
 * time between calls.  The library reads NO environment variable (the SNAPPIER_HIP_* knobs of the A/B scripts exist only in variant
 * builds made with -DSNAPPIER_HIP_DEBUG_ENV, scripts/build_variant.sh).  No option changes a RESULT -- bytes, lengths and status codes are the same under every setting
 * (tests/test_gpu_parity.py, tests/test_gpu_fuzz.py run every layout against the oracle); they choose kernels and memory behaviour.
 * snp_ctx_set_option returns SNP_ERR_BAD_ARG for an unknown option or a value outside its range and changes nothing then. */
typedef enum snp_option {
    /* How snp_decompress_batch lays a batch out.  0 (default): by what the context's PREVIOUS batch looked like -- small blocks
     * (<= SNP_OPT_SMALL_BLOCK_MAX declared bytes) by a lane or a team of lanes each, the rest one block per wavefront; a workload
     * that alternates between block sizes should pin the layout per call instead: 1 = one block per wavefront only (no small-block
     * pre-pass), 2 = pre-pass with one lane per block, 3 / 4 / 5 = pre-pass with a team of 4 / 8 / 16 lanes per block;
     * 6 = every block by the SERIAL kernel (decompress.hip: one wavefront walks the tags one at a time -- the reference's loop as it
     * stands, SnappyDecompressor.cs:184-347; the parity baseline and a debugging aid, ~20 x slower). */
    SNP_OPT_DECODE_LAYOUT = 1,
    SNP_OPT_SMALL_BLOCK_MAX = 2,        /* bytes; blocks declaring at most this many take the pre-pass (0 = never; default 512) */
    SNP_OPT_SMALL_BLOCK_MIN_BATCH = 3,  /* ... in batches of at least this many blocks (default 4096) */
    /* snp_compress_batch: 0 (default) by batch size -- below SNP_OPT_COMPRESS_WINDOW_MAX_BATCH fragments one fragment per wavefront
     * with the hash table in LDS, from there on one fragment per lane with the tables in an HBM workspace; 2 / 3 pin the latter / former;
     * 4 = the per-wavefront kernel with its table in a global-memory slot instead of LDS (12 wavefronts per CU whose slots stay cache resident);
     * 5 = BOTH table forms side by side on two streams, fragments drawn from one ticket counter (what layout 0 runs from
     * SNP_OPT_COMPRESS_WINDOW_DUAL_MIN_BATCH fragments up: 45-47 GB/s against 34.5 / 36.5-39 for either form alone; the call forks onto a
     * context-owned side stream and joins back before it returns control of the stream -- capturable). */
    SNP_OPT_COMPRESS_LAYOUT = 4,
    SNP_OPT_COMPRESS_WINDOW_MAX_BATCH = 5,   /* default 32768 (round 6: the dual form carries the per-wavefront kernel further up) */
    /* The lane compressor keeps 64 KiB of hash table per fragment of a launch in an HBM workspace (10.7 GB for 163 840 fragments; batches
     * above 262 144 fragments run in slices).  The workspace belongs to the DEVICE, not to the context: every context on a device borrows the
     * same one for the duration of a launch sequence (a GPU-side event orders the borrowers; no host thread blocks), so eight caller
     * threads cost one workspace, not eight (the reference pools one table per compressor: HashTable.cs:22-55).  It is built by the first
     * large compress call on the device (or snp_ctx_reserve_compress), grows when a larger batch arrives, and is freed with the device's
     * last context.
     * PLACEMENT: how fast HBM serves the tables' random traffic depends on where the driver placed the memory -- device memory consists of
     * regions, tens of GiB long, of three kinds, and the traffic runs 20-25 % faster spread evenly over two or three kinds than confined to
     * one.  So a workspace of >= 1 GiB is built from up to 16 separately allocated pieces chosen by measurement:
     * candidate pieces (1/16 of the workspace each) are allocated 16 at a time and probed in pairs (5 ms per probe) until a balanced set
     * exists or the budget is spent; the losers are freed before the call returns.
     * BUDGET (round 5): by DEFAULT the candidates never exceed TWO workspaces' worth (one transient extra workspace, a few hundred ms)
     * nor half of the device's free memory -- a library must not take seconds or crowd a shared device on its own account.  Fresh device
     * memory costs the driver ~27 ms per GiB to hand out, and the third kind may lie 150 GB of allocations away: a caller that wants the
     * thorough search (worth ~5 % of the compressor's rate) asks for it -- SNP_OPT_TABLE_PROBE_TRIES = 3..24 workspaces' worth, within half of
     * free memory or, when SNP_OPT_TABLE_PROBE_MAX_BYTES is set, within that many bytes (honoured up to 7/8 of what is free: the third kind has
     * been seen to begin beyond the first half) -- and runs it at start-up through snp_ctx_reserve_compress.
     * SNP_OPT_TABLE_PROBE_TRIES = 1: no search and no pool -- this context keeps a plain one-allocation workspace of its own. */
    SNP_OPT_TABLE_PROBE_TRIES = 6,
    SNP_OPT_TABLE_PROBE_MAX_BYTES = 7,
    SNP_OPT_PARALLEL_DECODE_MIN = 8,    /* snp_try_decompress: declared bytes from which ONE block is decoded a wavefront per 64 KiB fragment (0 = never; default 262144) */
    SNP_OPT_FENCED = 9,                 /* 1 (default): a wavefront drains its stores before it reads output bytes it wrote itself; 0 relies on in-order vector memory */
    SNP_OPT_DECODE_LEFTOVERS = 10,      /* blocks the pre-pass leaves over: 0 (default) by the previous batch, 1 one workgroup per block, 2 a list for persistent wavefronts */
    /* Which CRC-32C kernel runs (named SNP_OPT_CRC_TABLE_FREE until round 4; same number): 0 (default) the GF(2) shift map sliced
     * 11 + 11 + 10 bits out of three tables in LDS (5.6-5.9 TB/s; batches too small to amortise the table copy take form 2); 1 = TABLE-FREE, the
     * map applied bit by bit in registers (1.7 TB/s: VALU-bound; gfx950 has neither a CRC instruction nor a carry-less multiply);
     * 2 = round 3's four 256-entry tables (5.3 TB/s).  Same results. */
    SNP_OPT_CRC_KERNEL = 11,
    /* ---- launch shapes (round 6: every path the library contains is reachable through an option, so that the parity tests exercise the
     * shipped binary; the defaults are the measured best and nothing below changes a result) ---- */
    SNP_OPT_COMPRESS_WINDOW_POSITIONS = 12,        /* per-wavefront compressor: window positions per lane, 1 (default) | 2 (measured slower: 58 % of its rounds are cut) */
    SNP_OPT_COMPRESS_WINDOW_GLOBAL_MIN_BATCH = 13, /* layout 0: window-kernel batches of at least this many fragments keep their table in a global-memory slot (default 4096) */
    /* Lane compressor, how a lane's output and probes are issued: -1 (default) by batch size, else a mask -- 1: a short literal may leave as one
     * 16-byte store that overshoots inside MaxCompressedLength, 2: tag + body of a literal in one store, 4: a copy tag as one 4-byte store,
     * 8: 16- instead of 32-byte match-extension trips, 16: output staged per lane in LDS and written in 64-byte runs, 64: probe + insert as
     * one atomic exchange (one probe per trip only), 128: probe bytes from a 16-byte register window; 0 = exact-length stores only. */
    SNP_OPT_COMPRESS_LANE_STORES = 14,
    SNP_OPT_COMPRESS_LANE_PROBES = 15,             /* lane compressor: probes of a lane's scan issued together, 0 (default: 1 from 131 072 fragments, else 2) | 1..4 */
    SNP_OPT_COMPRESS_LANES_PER_WAVEFRONT = 16,     /* lane compressor: fragments per wavefront, 0 (default: 64 / 32 / 16 by batch size) | 8 | 16 | 32 | 64 */
    SNP_OPT_COMPRESS_SLICE = 17,                   /* lane compressor: fragments per launch (>= 4096; default 262144) -- larger batches run in slices */
    /* Lane compressor on batches of SMALL fragments: a launch that first copies each fragment into LDS.  -1 (default): when the previous
     * batch's longest fragment lay in (80, 768] bytes; 0 never; else the LDS slot in bytes (<= 2048; fragments up to that long take it). */
    SNP_OPT_COMPRESS_SMALL_INPUT_LDS = 18,
    SNP_OPT_COMPRESS_SMALL_INPUT_LANES = 19,       /* ... its lanes per wavefront: 0 (default 32) | 16 | 32 | 64 */
    SNP_OPT_FRAME_SCAN = 20,                       /* snp_frame_decode_device header walk: 0 (default) spans walked concurrently, 1 one lane, serial */
    SNP_OPT_DECODE_LDS_THROTTLE = 21,              /* bytes of dynamic LDS requested per decode wavefront purely to cap wavefronts per CU (0 = none; measurements) */
    SNP_OPT_COMPRESS_WINDOW_GLOBAL_SLOTS = 22,     /* per-wavefront compressor, table in a global slot: wavefronts (= 32 KiB slots) it runs with; 0 (default) = 12 per CU alone, 10 beside the LDS form */
    SNP_OPT_COMPRESS_WINDOW_DUAL_MIN_BATCH = 23    /* layout 0: window-kernel batches of at least this many fragments run both table forms side by side (default 1536) */
} snp_option;
#define SNP_OPT_CRC_TABLE_FREE SNP_OPT_CRC_KERNEL   /* deprecated name (rounds 1-4); same number, values 0 / 1 mean the same */
snp_status snp_ctx_set_option(snp_ctx* ctx, int option, int64_t value);
snp_status snp_ctx_get_option(const snp_ctx* ctx, int option, int64_t* out_value);
/* Builds the device's hash-table workspace for batches of up to `nfragments` 64 KiB fragments NOW instead of on the first large
 * snp_compress_batch / snp_frame_encode* call (see SNP_OPT_TABLE_PROBE_TRIES).  A service calls it once at start-up, before its own buffers
 * crowd the device: the first request then pays nothing, and the placement search sees all of device memory.  With default options the search
 * is the same bounded one a compress call would run (two workspaces' worth of candidates, < 1 s); with SNP_OPT_TABLE_PROBE_TRIES set it goes on
 * looking for a third kind of memory as far as that budget allows (seconds: the driver clears fresh memory at ~27 ms per GiB).  Batches above
 * 262 144 fragments run in slices, so that is the most it reserves.  Later, larger batches still grow the workspace on demand.
 * SNP_OK, or SNP_ERR_DEVICE (snp_ctx_last_error says why). */
snp_status snp_ctx_reserve_compress(snp_ctx* ctx, uint32_t nfragments);
const char* snp_status_string(int status);
const char* snp_version(void);

/* ---- host-only arithmetic (no device needed) ----------------------------------------------------------- */

/* Snappy.GetMaxCompressedLength  Snappy.cs:20-24  (= Helpers.MaxCompressedLength + 5).  Returns -1 if n < 0 or the result overflows int32. */
int64_t snp_max_compressed_length(int64_t n);
/* Helpers.MaxCompressedLength  Helpers.cs:17-46 (no varint padding). */
int64_t snp_max_fragment_compressed_length(int64_t n);
/* Snappy.GetUncompressedLength  Snappy.cs:136-137 -> VarIntEncoding.Read  VarIntEncoding.Read.cs:16-79. */
snp_status snp_get_uncompressed_length(const uint8_t* in, size_t n, uint32_t* out_len, uint32_t* out_header_bytes);

/* ---- single-buffer, host pointers ---------------------------------------------------------------------- */

/* Snappy.TryCompress  Snappy.cs:55-67 -> SnappyCompressor.TryCompress  SnappyCompressor.cs:24-83.
 * Any input length < 2^32: varint preamble + one compressed fragment per 65536 input bytes; fragments are
 * compressed concurrently (one wavefront each) and concatenated on the device.
 * SNP_ERR_OUTPUT_TOO_SMALL (written = 0) when cap cannot hold the result; SNP_ERR_OVERLAP when in/out overlap. */
snp_status snp_try_compress(snp_ctx* ctx, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);

/* Snappy.TryDecompress / Snappy.DecompressToMemory  Snappy.cs:172-186,223-235 -> SnappyDecompressor.Decompress
 * SnappyDecompressor.cs:43-92,184-347.  One whole Snappy block of any declared length. */
snp_status snp_try_decompress(snp_ctx* ctx, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);

/* The same two calls for an input that arrives in SEGMENTS -- Snappy.Compress(ReadOnlySequence<byte>, IBufferWriter<byte>) and
 * Snappy.Decompress / DecompressToMemory(ReadOnlySequence<byte>)  Snappy.cs:82-89,194-212,246-262: the concatenation of the nseg host
 * segments is ONE input (one Snappy block); each segment is uploaded straight from where it lies (pinned by the caller for the
 * duration of the call), so the managed side does not flatten the sequence first.  Same results and status codes as the span forms. */
snp_status snp_try_compress_segments(snp_ctx* ctx, const uint8_t* const* seg, const size_t* seg_len, uint32_t nseg,
                                     uint8_t* out, size_t cap, size_t* written);
snp_status snp_try_decompress_segments(snp_ctx* ctx, const uint8_t* const* seg, const size_t* seg_len, uint32_t nseg,
                                       uint8_t* out, size_t cap, size_t* written);

/* Crc32CAlgorithm.Compute + ApplyMask  Crc32CAlgorithm.cs:41-49,156-158. masked != 0 applies the framing mask. */
snp_status snp_crc32c(snp_ctx* ctx, const uint8_t* in, size_t n, int masked, uint32_t* out_crc);

/* SnappyStreamCompressor.Write + Flush over one whole buffer  SnappyStreamCompressor.cs:40-55,82-97,194-261:
 * stream identifier + one chunk per 65536 input bytes (type 0x00 if the compressed body is smaller than the raw
 * chunk, else type 0x01), each with the masked CRC-32C of its raw bytes. */
int64_t snp_frame_max_encoded_length(int64_t n);
snp_status snp_frame_encode(snp_ctx* ctx, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);

/* SnappyStreamDecompressor.Decompress over one whole framed stream  SnappyStreamDecompressor.cs:38-208:
 * chunk types 0x00/0x01 are decoded and CRC-checked, 0x80..0xff skipped, 0x02..0x7f rejected.
 * snp_frame_decoded_length scans chunk headers on the host and returns the total decoded size. */
snp_status snp_frame_decoded_length(const uint8_t* in, size_t n, uint64_t* out_len);
snp_status snp_frame_decode(snp_ctx* ctx, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written);

/* ---- batch, device pointers (the hot path; asynchronous on the context's stream) ------------------------ */

/* Stream capture: snp_compress_batch, snp_decompress_batch, snp_crc32c_batch and snp_frame_encode_device only enqueue kernels, so they may be called while the context's
 * stream is being captured into a hipGraph and replayed later (the graph reads the device arrays as they are at replay time).  A captured call
 * queries, synchronises and allocates nothing; it therefore needs the workspaces to exist already -- make the same call once before
 * the capture (compress keeps a workspace from 1 536 fragments on, and from there on also forks onto a context-owned side stream that the first such call creates; from 32 768 on snp_ctx_reserve_compress builds the workspace too).  A call that would have to allocate during a capture returns
 * SNP_ERR_DEVICE (snp_ctx_last_error says so) and leaves the capture valid.  The host-pointer entry points synchronise and cannot be captured.
 * LIFETIME: a captured graph holds the addresses of the workspaces it ran on.  From the first captured call on, the context (and the device's
 * table pool) never frees a workspace it has handed out -- a later, larger call allocates a new one next to it -- until snp_ctx_destroy; a graph
 * must not be replayed after its context is destroyed, nor concurrently with another context's compress on the same device (the event that
 * orders borrowers of the table pool is not part of a capture). */

/* nblocks independent inputs, each <= 65536 bytes (one fragment, SnappyCompressor.cs:40-80 loop body):
 * block b reads in[in_off[b] .. +in_len[b]) and writes  varint(in_len[b]) || CompressFragment  at
 * out[out_off[b] ..), which must have room for snp_max_compressed_length(in_len[b]) bytes.
 * out_len[b] = bytes written, status[b] = SNP_OK | SNP_ERR_BAD_ARG (in_len[b] > 65536).
 * Layout by batch size: below 32 768 fragments one fragment per wavefront with the u16 hash table in LDS -- from 1 536 fragments on joined by a second
 * population of wavefronts whose tables sit in cache-resident global-memory slots, both drawing fragments from one ticket counter -- (compress_win.hip), from
 * there on one fragment per LANE with the tables in an HBM workspace the device owns (compress_lanes.hip); all emit the reference's bytes.  All arrays are device memory. */
snp_status snp_compress_batch(snp_ctx* ctx, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                              uint32_t nblocks, uint8_t* out, const uint64_t* out_off, uint32_t* out_len,
                              int32_t* status);

/* nblocks independent Snappy blocks: block b reads in[in_off[b] .. +in_len[b]) (varint preamble + tags) and writes
 * at most out_cap[b] bytes at out[out_off[b] ..).  out_len[b] = declared/decoded length, status[b] per block.
 * SnappyDecompressor.cs:184-347 semantics.  One wavefront per block (decompress.hip); in batches of >= 4096 blocks, clean blocks
 * that declare at most 512 bytes are first decoded by a lane or a team of lanes each (decompress_small.hip), unless the previous
 * batch of this context showed mostly larger blocks (the context keeps a 272-byte asynchronous read-back of how its last batch
 * went: a performance memory only, results do not depend on it).  While the call is in flight status[b] may transiently hold -1
 * (a block the first pass left to the second); it is final when the stream reaches the end of the call's work. */
snp_status snp_decompress_batch(snp_ctx* ctx, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                                uint32_t nblocks, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap,
                                uint32_t* out_len, int32_t* status);

/* CRC-32C (optionally masked) of nblocks independent byte ranges; wave-parallel (each lane folds its dwords with a GF(2)-linear
 * shift map; which form of the map runs is SNP_OPT_CRC_KERNEL: by default sliced 11 + 11 + 10 bits out of three LDS tables, crc32c.hip). */
snp_status snp_crc32c_batch(snp_ctx* ctx, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                            uint32_t nblocks, int masked, uint32_t* out_crc);

/* Concatenate nblocks byte ranges: out[dst_off[b] .. +in_len[b]) = in[in_off[b] .. +in_len[b]).  This is the compaction
 * step after snp_compress_batch (blocks sit at a fixed stride) -- what SnappyCompressor.TryCompress does by advancing
 * its output span fragment after fragment (SnappyCompressor.cs:40-80) -- and what a rank does before the payload gather
 * of a multi-GPU job.  dst_off = exclusive prefix sum of the lengths (device memory, computed by the caller). */
snp_status snp_concat_batch(snp_ctx* ctx, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                            uint32_t nblocks, uint8_t* out, const uint64_t* dst_off);

/* Device-resident framing (config 4): raw stream d_in[0..n) -> framed stream in d_out (capacity cap, device),
 * *d_written (device u64) = encoded size.  d_work must hold snp_frame_encode_workspace(n) bytes. */
uint64_t snp_frame_encode_workspace(uint64_t n);
snp_status snp_frame_encode_device(snp_ctx* ctx, const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap,
                                   uint64_t* d_written, void* d_work);
/* Device-resident decode of a framed stream whose chunk table the caller already has on the device:
 * chunk c has type chunk_type[c] (0 or 1), body d_in[body_off[c] .. +body_len[c]) (after the 4 CRC bytes),
 * expected masked CRC chunk_crc[c], and decodes to d_out[out_off[c] .. +out_cap[c]).  status[c] per chunk. */
snp_status snp_frame_decode_chunks_device(snp_ctx* ctx, const uint8_t* d_in, const uint8_t* chunk_type,
                                          const uint64_t* body_off, const uint32_t* body_len,
                                          const uint32_t* chunk_crc, uint32_t nchunks, uint8_t* d_out,
                                          const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len,
                                          int32_t* status);

/* Fully device-resident decode of a framed stream that arrives WITHOUT a chunk table (SnappyStreamDecompressor.cs:53-199):
 * device kernels walk the chunk headers -- the stream is cut into 1 MiB spans that are walked concurrently, each from a few
 * candidate entry points, and one wavefront then resolves which candidate of every span the true chain enters (frame_scan.hip;
 * ~2 ms per 10 GiB) -- and build the chunk table in d_work; then all chunks are decoded and CRC-checked in one launch each.  max_chunks bounds the table
 * (a stream with more data chunks, or one that decodes to more than cap bytes, ends with SNP_ERR_OUTPUT_TOO_SMALL);
 * d_work must hold snp_frame_decode_workspace(max_chunks) bytes.  d_result (device, 2 x u64): [0] = bytes written
 * (0 unless OK), [1] = status of the stream: the first failing chunk in stream order, else the error that ended the
 * header walk, else SNP_OK.  Everything is enqueued on the context's stream. */
uint64_t snp_frame_decode_workspace(uint32_t max_chunks);
snp_status snp_frame_decode_device(snp_ctx* ctx, const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap,
                                   uint32_t max_chunks, void* d_work, uint64_t* d_result);

#ifdef __cplusplus
}
#endif
#endif /* SNAPPIER_HIP_H */
