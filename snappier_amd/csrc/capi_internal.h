// capi_internal.h -- what the host-side files of libsnappier_hip.so share: the context (snp_ctx), the device's hash-table pool, the kernel
// launchers the other .hip files export.  The C-ABI itself is include/snappier_hip.h; it is implemented in
//     capi_ctx.hip     contexts, options, counters, scratch buffers, host-only arithmetic (varint, MaxCompressedLength)
//     capi_pool.hip    one hash-table workspace per DEVICE (TablePool), its placement search (piece_search.h), borrow / return
//     capi_batch.hip   launch policy of the device-pointer batch entry points (which kernel for which batch), snp_*_batch
//     capi_host.hip    the host-pointer calls: snp_try_compress / snp_try_decompress (+ segments), snp_crc32c
//     capi_frame.hip   framing orchestration: snp_frame_encode* / snp_frame_decode*
// No codec arithmetic happens on the host: every byte of compress / decompress / CRC work is done by the gfx950 kernels in compress_lanes.hip,
// compress_win.hip, decode_chains.hip, decompress.hip, decompress_small.hip, tag_index.hip, crc32c.hip, framing.hip, frame_scan.hip.
// There is no CPU fallback -- without a HIP device snp_ctx_create fails with SNP_ERR_DEVICE.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>
#include <cmath>
#include <chrono>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "snp_device.h"

extern "C" {
hipError_t snp_launch_decompress(const u8*, const u64*, const u32*, u32, u8*, const u64*, const u32*, u32*, i32*,
                                 const u8*, int, hipStream_t, const u32*);
u32 snp_tag_index_entries(u32, u32);
size_t snp_tag_index_workspace_bytes(u32, u32);
size_t snp_tag_index_fallback_offset(u32, u32);
hipError_t snp_launch_tag_index(const u8*, u32, u32, u32, u64*, u64*, u32*, u64*, u32*, u32*, hipStream_t);
u32 snp_tag_index_chunks_ready(u32, u32, u64);
hipError_t snp_launch_tag_index_begin(u64*, u32, u32, int, hipStream_t);
int snp_tag_index_look_back_only(u32, u32);
hipError_t snp_launch_tag_index_chunks(const u8*, u32, u32, u64*, u32, u32, hipStream_t);
hipError_t snp_launch_tag_index_finish(const u8*, u32, u32, u32, u64*, u64*, u32*, u64*, u32*, u32*, hipStream_t);
hipError_t snp_launch_compress_win(const u8*, const u64*, const u32*, u32, u8*, const u64*, u32*, i32*, int, int, int,
                                   hipStream_t, uint16_t*, u32);
size_t snp_compress_win_table_bytes(u32);
hipError_t snp_launch_compress_win_dual(const u8*, const u64*, const u32*, u32, u8*, const u64*, u32*, i32*, int, int, hipStream_t, hipStream_t,
                                        hipEvent_t, hipEvent_t, uint16_t*, u32, u32, u32*);
hipError_t snp_launch_decompress_small(const u8*, const u64*, const u32*, u32, u8*, const u64*, const u32*, u32*, i32*, const u8*,
                                       u32, hipStream_t, u32*, u32*, u32, u32);
hipError_t snp_launch_sample_caps(const u32*, u32, u32, u32*, hipStream_t);
hipError_t snp_launch_decompress_list(const u8*, const u64*, const u32*, u32, u8*, const u64*, const u32*, u32*, i32*, const u8*, int,
                                      hipStream_t, const u32*, u32*, u32, u32);
hipError_t snp_launch_compress_lanes(const u8*, const u64*, const u32*, u32, u8*, const u64*, u32*, i32*, int, int, const snp_table_pieces*,
                                     u32*, hipStream_t, const snp_lane_tuning*);
size_t snp_compress_lanes_workspace(u32);
hipError_t snp_probe_tables(const snp_table_pieces*, u32, u32, hipStream_t, float*);
hipError_t snp_launch_crc32c(const u8*, const u64*, const u32*, u32, int, u32*, const u32*, i32*, hipStream_t);
hipError_t snp_launch_gather(const u8*, const u64*, const u32*, u8*, const u64*, u32, hipStream_t);
hipError_t snp_launch_frame_chunks(u64, u32, u64, u64*, u32*, u64*, hipStream_t);
hipError_t snp_launch_frame_plan(const u32*, const u32*, u32, u8*, u32*, u64*, u64*, hipStream_t);
hipError_t snp_launch_frame_header_only(u8*, u64*, hipStream_t);
hipError_t snp_launch_frame_scan(const u8*, u64, u64, u32, u8*, u64*, u32*, u32*, u64*, u32*, u64*, hipStream_t);
hipError_t snp_launch_frame_result(const i32*, const u64*, u64*, hipStream_t);
size_t snp_frame_scan_workspace(u64);
hipError_t snp_launch_frame_scan_spans(const u8*, u64, u64, u32, u8*, u64*, u32*, u32*, u64*, u32*, u64*, void*, hipStream_t);
hipError_t snp_launch_frame_emit(const u8*, const u64*, const u8*, const u64*, const u8*, const u32*, const u32*,
                                 const u64*, u8*, u64, u32, hipStream_t);
}

constexpr u64 kSnpCompStride = 76496 + 16;   // snp_max_compressed_length(65536), padded to a 16-byte multiple

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

inline u64 snp_align_up(u64 v, u64 a) { return (v + a - 1) / a * a; }

// ---- ONE hash-table workspace per DEVICE, shared by every context on it (capi_pool.hip) ---------------------------------------------------------
// The lane compressor's tables are 64 KiB per fragment in flight (10.7 GB for 163 840): a context per caller thread must not mean a workspace per
// caller thread, nor a placement search per context (the reference pools ONE table per compressor: HashTable.cs:22-55).  Contexts borrow the
// device's workspace for the duration of one launch sequence: lock, make the stream wait for the previous borrower's event (a GPU-side wait: no
// host thread blocks), launch, record the event, unlock.  A launch of >= 16 384 fragments fills the chip, so taking turns costs nothing that
// running side by side would have gained.  The pool is built on first use, grows when a larger batch arrives (the old one is freed once its
// last borrower's work is done -- or kept until the pool dies when a captured hipGraph may still hold its address), and dies with the device's
// last context.
struct TablePool {
    std::mutex mu;
    int users = 0;                                       // live contexts on this device
    snp_table_pieces tp{};
    void* plain = nullptr;                               // the one-allocation form (tp.p[0]) ...
    size_t plain_cap = 0;
    std::vector<void*> pieces;                           // ... or the searched form: up to 16 pieces (PieceSearch)
    hipEvent_t last_use = nullptr;                       // recorded by the previous borrower after its launches
    hipStream_t last_stream = nullptr;
    bool used = false;
    bool pinned = false;                                 // a borrower was capturing a hipGraph: no workspace this pool ever handed out is freed before the pool dies
    std::vector<void*> retired;
    uint64_t stats[4] = {0, 0, 0, 0};                    // chosen set's probe us, candidates, search us, most bytes the search held at once
    void drain();
    void drop_workspace();                               // callers hold mu
    void destroy();                                      // the device's last context is gone
};
// The pool of a device ordinal (created on first use; nullptr for an ordinal the table does not hold: snp_ctx_create refuses such a device).
TablePool* snp_pool_of(int device);
constexpr int kSnpMaxDevices = 64;

// Scope in which this thread's potentially capture-unsafe calls (event / stream queries, a synchronous read-back, an allocation) are legal
// although another thread may be capturing in global mode.
struct RelaxedCaptureMode {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    bool ok;
    RelaxedCaptureMode() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!ok) (void)hipGetLastError(); }
    ~RelaxedCaptureMode() { if (ok && hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
};

struct snp_ctx {
    int device = 0;
    int variant = SNP_HASH_CRC32C;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int fenced = 0;          // decompress kernel mode: bit 0 FENCED, bit 1 the serial kernel (SNP_OPT_DECODE_LAYOUT = 6); higher bits select lab front ends in variant builds
    int dec_lds = 0;         // dynamic LDS bytes per decode wavefront (SNP_OPT_DECODE_LDS_THROTTLE: an occupancy throttle)
    int decode_layout = 0;   // 0 default (small blocks one per lane, the rest one per wavefront), 1 a front end without the pre-pass is pinned
    int table_tries = 2;     // workspaces' worth of candidate pieces the search for a >= 1 GiB hash-table workspace may hold at once (SNP_OPT_TABLE_PROBE_TRIES;
                             // never more than fit in half of the free memory and under the byte cap; 1 = no search, one allocation).
                             // The search stops long before that when it can: piece_search.h
    uint64_t table_probe_max_bytes = 0;   // SNP_OPT_TABLE_PROBE_MAX_BYTES: cap on what the placement probe's candidates may occupy together (0 = half of free memory only)
    u32 par_min = 4 * SNP_BLOCK_SIZE;   // single blocks at least this long are decoded one wavefront per 64 KiB fragment (0 = never)
    int compress_mode = 0;   // 0 auto by batch size, 2 fragment-per-lane with HBM tables (compress_lanes.hip), 3 fragment-per-wavefront
                             // with the table in LDS, multi-token windows (compress_win.hip), 4 the same with the table in a global-memory slot
    int win_np = 1;          // window compressor: positions per lane (SNP_OPT_COMPRESS_WINDOW_POSITIONS = 1 | 2; 2 measured slower)
    u32 small_max = 512;     // blocks declaring at most this many bytes go through the small-block pre-pass (decompress_small.hip: a lane or
                             // a team of lanes per block, out of LDS); 0 = never.  Above 512 bytes the wave kernel is faster (768-1024 B:
                             // teams 180-260 GB/s, wave kernel 280-335; profiles/r02t_team_budget.jsonl).
    u32 small_min_blocks = 4096;   // ... in batches of at least this many blocks
    int crc_kernel = 0;            // SNP_OPT_CRC_KERNEL: 0 = three LDS tables of 11 + 11 + 10 bits (default), 1 = the table-free kernel (1.7 TB/s), 2 = four 8-bit tables (round 3)
    int crc_bits() const { return crc_kernel == 1 ? 2 : crc_kernel == 2 ? 4 : 0; }
    bool no_prepass = false;       // SNP_OPT_DECODE_LAYOUT = 1: every block by the one-block-per-wavefront kernel
    bool small_lanes = false;      // SNP_OPT_DECODE_LAYOUT = 2: the block-per-lane kernel instead of a team of lanes per block
    bool redo_grid = false, redo_list = false;   // SNP_OPT_DECODE_LEFTOVERS pins how the pre-pass's leftovers are decoded (default: by how the previous batch went)
    u32 small_team_log = 0;        // SNP_OPT_DECODE_LAYOUT = 3 / 4 / 5: lanes per block (0 = the kernel's default)
    u32 slice_fragments = 262144;   // fragments per lane-compressor launch (SNP_OPT_COMPRESS_SLICE)
    u32 win_gtab_min = 4096; // auto mode: window-kernel batches of at least this many fragments keep their tables in global memory (SNP_OPT_COMPRESS_WINDOW_GLOBAL_MIN_BATCH):
                             // 36.5 vs 34.6 GB/s from 4 096 fragments up, 19.8 vs 35.5 at 1 024 (profiles/r05zz_compress_by_batch.jsonl)
    u32 win_gslots = 0;      // global-slot window kernel: wavefronts (= 32 KiB table slots) it runs with; 0 = by the form: 12 per CU alone (3 072: 38.9-39.5 GB/s against 36.5 at 32 per CU,
                             // whose slots thrash L2), 10 per CU beside the LDS form (SNP_OPT_COMPRESS_WINDOW_GLOBAL_SLOTS; profiles/r06c_compress_mix_l2_slots*.jsonl)
    u32 win_dual_min = 1536; // auto mode: window-kernel batches of at least this many fragments run BOTH table forms side by side (compress_win.hip, dual form)
    // the dual form's side stream and its fork / join events (created on first use, outside any capture)
    hipStream_t side_stream = nullptr;
    hipEvent_t side_ev[2] = {nullptr, nullptr};
    int side_state = 0;
    bool side_stream_ready();
    u32 win_max = 32768;     // auto mode: batches below this many fragments take the window kernel (SNP_OPT_COMPRESS_WINDOW_MAX_BATCH): the lane kernel needs its
                             // ~31-37 ms whatever the count up to ~20 000 fragments (round 6, against the dual form: 16 383: 32.7 vs 50.2 GB/s; 20 480: 36.1 vs 47.9;
                             // 24 576: 51.3 dual; 32 768: 50.8-51.7 lanes vs 49.1 dual -- profiles/r06d_compress_by_batch.jsonl, r06e_dual_reserve.txt)
    snp_lane_tuning lane_tune{0, -1, 0, -1, 0, 0};   // SNP_OPT_COMPRESS_LANE_*: launch shape of the lane compressor (0 / -1 = by batch size)
    DevBuf in, out, meta, work, fragtab, scan, small, redo, win_tables;
    int frame_scan = 0;      // header walk of snp_frame_decode_device: 0 spans walked concurrently (frame_scan.hip), 1 one lane, serial (SNP_OPT_FRAME_SCAN)
    uint64_t counters[7] = {0, 0, 0, 0, 0, 0, 0};   // snp_ctx_counter
    bool table_tries_set = false;   // SNP_OPT_TABLE_PROBE_TRIES was given: the implicit in-call search honours it as is
    std::string err;

    // ---- capi_batch.hip: one launch sequence of the decompressor / compressor over nblocks blocks, picking the layout --------------------------
    bool launch_decompress(const u8* d_in, const u64* in_off, const u32* in_len, u32 nblocks, u8* d_out, const u64* out_off,
                           const u32* out_cap, u32* out_len, i32* status, const u8* chunk_type);
    bool launch_compress(const u8* d_in, const u64* in_off, const u32* in_len, u32 nblocks, u8* d_out, const u64* out_off,
                         u32* out_len, i32* status, int emit_varint);
    u32* hint = nullptr;                                 // pinned: the previous batch's list length
    hipEvent_t hint_ev = nullptr;
    bool hint_pending = false, hint_mostly_large = false, hint_from_prepass = false, hint_seen = false;
    u32 hint_blocks = 0, hint_mean_cap = 256;            // (no history yet: assume 256-byte blocks)
    bool hint_ready();
    u32 n_waves = 0;
    u32 persistent_waves();                              // one chip-full of 64-thread workgroups at 8 wavefronts per SIMD
    u32* chint = nullptr;                                // pinned: the longest fragment of the previous lane-compressor launch
    hipEvent_t chint_ev = nullptr;
    bool chint_pending = false, chint_small = false, chint_mid = false;
    int chint_tiny = 0;
    bool chint_ready();

    // ---- capi_ctx.hip: errors, scratch, transfers, streams ------------------------------------------------------------------------------------
    bool check(hipError_t e, const char* what);
    bool ensure(DevBuf& b, size_t bytes, const char* what);
    // Host <-> device transfers of the host-pointer entry points: the caller's buffers are pageable (the reference's Span API)
    // and the runtime's own pageable path moves them at ~40 GB/s; a pinned-slice pipeline inside the library measured slower
    // (31 GB/s, profiles/r02b_host_api_rates.jsonl) and was removed.
    bool h2d(void* dev, const void* host, size_t n, const char* what);
    bool d2h(void* host, const void* dev, size_t n, const char* what);
    bool stream_is_capturing();
    bool was_captured = false;                           // a call of this context ran under stream capture: no workspace is freed before snp_ctx_destroy
    std::vector<void*> kept;
    bool use_device() { return check(hipSetDevice(device), "hipSetDevice"); }
    // The context's scratch (hash tables, staging) is ordered by the stream it runs on.  Moving the context to another
    // stream: everything already queued on the old stream must finish before the new stream touches the scratch.
    bool rebind(hipStream_t next);
    hipEvent_t order_ev = nullptr;
    // copy stream: host -> device slices that overlap the kernels of the previous slice
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[2] = {nullptr, nullptr};
    int copy_state = 0;
    bool copy_stream_ready();

    // ---- capi_pool.hip: the hash-table workspace of the lane compressor (64 KiB per fragment in flight) belongs to the DEVICE: one allocation
    // below 1 GiB, above that up to 16 pieces chosen by PieceSearch (why: see there).  A context borrows it per launch sequence. ----------------
    TablePool* pool = nullptr;
    snp_table_pieces tp{};                               // the borrowed view (valid between borrow_tables and return_tables)
    bool borrowed = false;
    DevBuf own_tables;                                   // SNP_OPT_TABLE_PROBE_TRIES = 1: a plain one-allocation workspace of this context's own (no pool, no search)
    // Borrows the device's workspace for batches of up to nblocks fragments: builds or grows it if need be (thorough: snp_ctx_reserve_compress),
    // orders this context's stream behind the previous borrower.  return_tables() must follow the launches.
    bool borrow_tables(u32 nblocks, bool thorough = false);
    void return_tables();
    bool build_tables(TablePool& P, u32 nblocks, bool thorough, bool capturing);   // (callers hold P.mu)

    // ---- capi_host.hip: host input of nf 64 KiB fragments -> this->in, compressed into d_out (sliced upload that overlaps the compressor) ------
    bool upload_and_compress(const u8* host_in, size_t n, u32 nf, const u64* d_in_off, const u32* d_in_len, u8* d_out,
                             const u64* d_out_off, u32* d_out_len, i32* d_status, int emit_varint);
};

// Entry points run on the context's device and leave the caller's current device as they found it.
struct DevGuard {
    int prev = -1, dev = -1;
    bool ok = false;
    explicit DevGuard(snp_ctx* c) : dev(c->device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = prev == dev || c->use_device();
        if (ok && !c->was_captured && c->stream) (void)c->stream_is_capturing();   // (latches was_captured: from then on no workspace is freed before destroy)
    }
    ~DevGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

// capi_frame.hip, used by capi_host.hip's snp_frame_encode
snp_status snp_frame_encode_impl(snp_ctx* c, const uint8_t* d_in, const uint8_t* host_in, uint64_t n, uint8_t* d_out, uint64_t cap,
                                 uint64_t* d_written, void* d_work);
inline bool snp_ranges_overlap(const u8* a, size_t an, const u8* b, size_t bn) { return an && bn && a < b + bn && b < a + an; }
