// capi.hip -- the C-ABI of libsnappier_hip.so (include/snappier_hip.h): contexts, HBM scratch, host staging and
// the launch sequences behind each entry point.  No codec arithmetic happens on the host: every byte of compress /
// decompress / CRC work is done by the gfx950 kernels in compress_lanes.hip, compress_win.hip, decompress.hip,
// decompress_small.hip, tag_index.hip, crc32c.hip, framing.hip, frame_scan.hip.
// There is no CPU fallback -- without a HIP device snp_ctx_create fails with SNP_ERR_DEVICE.
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

#include "piece_search.h"
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
hipError_t snp_launch_decompress_small(const u8*, const u64*, const u32*, u32, u8*, const u64*, const u32*, u32*, i32*, const u8*,
                                       u32, hipStream_t, u32*, u32*, u32, u32);
hipError_t snp_launch_sample_caps(const u32*, u32, u32, u32*, hipStream_t);
hipError_t snp_launch_decompress_list(const u8*, const u64*, const u32*, u32, u8*, const u64*, const u32*, u32*, i32*, const u8*, int,
                                      hipStream_t, const u32*, u32*, u32, u32);
hipError_t snp_launch_compress_lanes(const u8*, const u64*, const u32*, u32, u8*, const u64*, u32*, i32*, int, int, const snp_table_pieces*,
                                     u32*, hipStream_t, int);
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

namespace {

constexpr int kDefaultDecLds = 0;
constexpr u64 kCompStride = 76496 + 16;   // snp_max_compressed_length(65536), padded to a 16-byte multiple

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

inline u64 align_up(u64 v, u64 a) { return (v + a - 1) / a * a; }

using snp_piece_search::PieceSearch;   // piece_search.h: why the workspace is made of pieces, and how they are chosen


// ---- ONE hash-table workspace per DEVICE, shared by every context on it -----------------------------------------------------------------------
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
    uint64_t serves(u32 nblocks_cap_of = 0) const { (void)nblocks_cap_of; return pieces.empty() ? 0 : static_cast<uint64_t>(tp.piece_frags) * tp.n; }
    void drain()
    {
        if (used && last_use) (void)hipEventSynchronize(last_use);
    }
    void drop_workspace()                                // callers hold mu
    {
        drain();
        auto gone = [&](void* q) { if (!q) return; if (pinned) retired.push_back(q); else (void)hipFree(q); };
        gone(plain);
        for (void* q : pieces) gone(q);
        plain = nullptr;
        plain_cap = 0;
        pieces.clear();
        tp = snp_table_pieces{};
    }
    void destroy()                                       // the device's last context is gone
    {
        pinned = false;
        drop_workspace();
        for (void* q : retired) (void)hipFree(q);
        retired.clear();
        if (last_use) (void)hipEventDestroy(last_use);
        last_use = nullptr;
        used = false;
    }
};
std::mutex g_pools_mu;
TablePool* pool_of(int device)
{
    static TablePool* pools[64] = {};
    std::lock_guard<std::mutex> g(g_pools_mu);
    if (device < 0 || device >= 64) device = 0;
    if (!pools[device]) pools[device] = new TablePool();
    return pools[device];
}

}  // namespace

// Scope in which this thread's potentially capture-unsafe calls (event / stream queries, a synchronous read-back) are legal although another
// thread may be capturing in global mode.
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
    int fenced = 0;          // decompress kernel mode: bit 0 FENCED, bit 1 serial-only (debug knobs, see snp_ctx_create)
    int dec_lds = 0;         // dynamic LDS bytes per decode wavefront (occupancy throttle)
    int decode_layout = 0;   // 0 default (small blocks one per lane, the rest one per wavefront), 1 a debug front end is pinned
    int table_tries = 2;     // workspaces' worth of candidate pieces the search for a >= 1 GiB hash-table workspace may hold at once (SNAPPIER_HIP_TABLE_TRIES /
                             // SNP_OPT_TABLE_PROBE_TRIES; never more than fit in half of the free memory and under the byte cap; 1 = no search, one allocation).
                             // The search stops long before that when it can: piece_search.h
    uint64_t table_probe_max_bytes = 0;   // SNP_OPT_TABLE_PROBE_MAX_BYTES: cap on what the placement probe's candidates may occupy together (0 = half of free memory only)
    u32 par_min = 4 * SNP_BLOCK_SIZE;   // single blocks at least this long are decoded one wavefront per 64 KiB fragment (0 = never)
    int compress_mode = 0;   // 0 auto by batch size, 2 fragment-per-lane with HBM tables (compress_lanes.hip), 3 fragment-per-wavefront
                             // with the table in LDS, multi-token windows (compress_win.hip)
    int win_np = 1;          // window compressor: positions per lane (SNAPPIER_HIP_WIN_NP = 1 | 2; 2 measured slower)
    u32 small_max = 512;     // blocks declaring at most this many bytes go through the small-block pre-pass (decompress_small.hip: a lane or
                             // a team of lanes per block, out of LDS); 0 = never.  Above 512 bytes the wave kernel is faster (768-1024 B:
                             // teams 180-260 GB/s, wave kernel 280-335; profiles/r02t_team_budget.jsonl).
    u32 small_min_blocks = 4096;   // ... in batches of at least this many blocks
    int crc_kernel = 0;            // SNP_OPT_CRC_KERNEL: 0 = three LDS tables of 11 + 11 + 10 bits (default), 1 = the table-free kernel (1.7 TB/s), 2 = four 8-bit tables (round 3)
    int crc_bits() const { return crc_kernel == 1 ? 2 : crc_kernel == 2 ? 4 : 0; }
    bool no_prepass = false;       // SNP_OPT_DECODE_LAYOUT = 1: every block by the one-block-per-wavefront kernel
    bool small_lanes = false;      // SNAPPIER_HIP_SMALL=lanes: the block-per-lane kernel instead of a team of lanes per block
    bool redo_grid = false, redo_list = false;   // SNAPPIER_HIP_REDO=grid|list pins how the pre-pass's leftovers are decoded (default: by how the previous batch went)
    u32 small_team_log = 0;        // SNAPPIER_HIP_SMALL=team4|team8|team16: lanes per block (0 = the kernel's default)
    u32 slice_fragments = 262144;   // fragments per lane-compressor launch (SNAPPIER_HIP_SLICE pins it)
    u32 win_gtab_min = 4096; // auto mode: window-kernel batches of at least this many fragments keep their tables in global memory (SNAPPIER_HIP_WIN_GTAB_MIN):
                             // 36.5 vs 34.6 GB/s from 4 096 fragments up, 19.8 vs 35.5 at 1 024 (profiles/r05zz_compress_by_batch.jsonl);
                                      // never by default: measured +5 % only (36.4 vs 34.6 GB/s at 4 096-16 383 fragments -- the kernel turns texture-path-bound, profiles/r04k_pmc_window_kernel.txt)
    u32 win_max = 20480;     // auto mode: batches below this many fragments take the window kernel (SNAPPIER_HIP_WIN_MAX): the lane kernel needs its
                             // ~31-37 ms whatever the count up to ~20 000 fragments (16 384: 33.0 GB/s against the window kernel's 36.5; 20 480: 35.9 against 35.7; 32 768: 52.2 against 36.3)
    DevBuf in, out, meta, work, fragtab, scan, small, redo, win_tables;
    int frame_scan = 0;      // header walk of snp_frame_decode_device: 0 spans walked concurrently (frame_scan.hip), 1 one lane, serial
    uint64_t counters[7] = {0, 0, 0, 0, 0, 0, 0};   // snp_ctx_counter
    bool table_tries_set = false;   // SNP_OPT_TABLE_PROBE_TRIES / SNAPPIER_HIP_TABLE_TRIES was given: the implicit in-call search honours it as is
    std::string err;

    // One launch sequence of the decompressor over nblocks blocks (decompress_small.hip, then decompress.hip).
    bool launch_decompress(const u8* d_in, const u64* in_off, const u32* in_len, u32 nblocks, u8* d_out, const u64* out_off,
                           const u32* out_cap, u32* out_len, i32* status, const u8* chunk_type)
    {
        // Large batches first go through the block-per-lane kernel, which finishes every clean block of <= small_max bytes
        // and marks the rest; the wave kernel then takes exactly those (all of them when every block is a 64 KiB block: the
        // first launch is then 163 840 lanes that read two words each).
        if (small_max && nblocks >= small_min_blocks && decode_layout == 0 && !no_prepass) {
            // Small clean blocks are finished by a pre-pass (decompress_small.hip: a lane or a team of lanes per block), which
            // appends the blocks it leaves over to a list; a chip-full of persistent wavefronts then decodes the list
            // (decompress.hip, k_decompress_chains_list: no launch per finished block -- 4 M blocks of 256 bytes 4.7 -> 3.3 ms, 64-byte
            // blocks 180 -> 420 GB/s).  For a batch of LARGE blocks all of that is overhead (a pre-pass that rejects 2 M blocks of
            // 512 bytes costs as much as decoding them, and the list kernel is 5-9 % slower than one workgroup per block), and what
            // a batch is like is known only on the device.  So the context remembers the previous batch: its leftover count, or a
            // sample of its capacities, comes back with an asynchronous copy that is read here only once it has landed, and this
            // batch is assumed to be alike (first batch: pre-pass).  The results are the same either way; SNAPPIER_HIP_REDO pins it.
            const u32 sub_cap = nblocks / 64 + 128;                     // a sub-list holds the leftovers of every 64th wavefront of the pre-pass
            if (!ensure(redo, (static_cast<size_t>(sub_cap) * 64 + 128) * 4, "hipMalloc(redo list)")) return false;
            u32* const ctl = static_cast<u32*>(redo.p);                 // [0..63] sub-list lengths, [64] ticket, [65..67] size sample
            u32* const list = ctl + 128;
            // (a stream that is being captured into a graph is never queried or synchronised, and no event of this context is: all of that would
            //  invalidate the capture.  A captured call goes by what the context knew before the capture began and leaves no hint behind.)
            const bool capturing = stream_is_capturing();
            // (the queries and the one-time synchronous sample below touch only this context's stream and events; in relaxed mode they do not
            //  invalidate a hipStreamCaptureModeGlobal capture that ANOTHER thread of the process has in progress: ADVICE r4)
            RelaxedCaptureMode relaxed;
            if (capturing) {
            } else if (hint && hint_ev && hint_pending && hipEventQuery(hint_ev) == hipSuccess) {
                hint_pending = false;
                if (hint_from_prepass) {
                    u64 left = 0;
                    for (int k = 0; k < 64; ++k) left += hint[k];
                    hint_mostly_large = left * 2 > hint_blocks;
                } else if (hint[67]) {
                    hint_mostly_large = static_cast<u64>(hint[65]) * 2 < hint[67];
                }
                if (hint[67]) hint_mean_cap = static_cast<u32>((static_cast<u64>(hint[66]) << 4) / hint[67]);
            } else {
                (void)hipGetLastError();
            }
            if (!hint_seen && !hint_pending && !capturing && hint_ready() && hipStreamQuery(stream) == hipSuccess) {
                // The FIRST batch of a context has no previous batch to go by.  When its stream is idle (nothing queued that a wait would
                // sit behind), a 10 us sample of this batch's capacities (k_sample_caps: <= 16 384 of them, strided) is read back at once:
                // a first call of 64 KiB blocks then goes straight to one block per wavefront instead of the pre-pass + list kernel (12.10 vs
                // 11.55 ms per 10 GiB).  A busy stream keeps the old default (pre-pass): results are the same either way.
                if (snp_zero_words_async(ctl, 68, stream) == hipSuccess &&
                    snp_launch_sample_caps(out_cap, nblocks, small_max, ctl, stream) == hipSuccess &&
                    hipMemcpyAsync(hint, ctl, 68 * 4, hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess && hint[67]) {
                    hint_mostly_large = static_cast<u64>(hint[65]) * 2 < hint[67];
                    hint_mean_cap = static_cast<u32>((static_cast<u64>(hint[66]) << 4) / hint[67]);
                } else {
                    (void)hipGetLastError();
                }
            }
            if (!capturing) hint_seen = true;
            const bool chains = (fenced & 8) != 0;
            const bool pinned = small_lanes || small_team_log != 0;          // the caller chose the pre-pass layout: the previous batch is not asked
            const bool prepass = redo_list || !chains || pinned || (!redo_grid && !hint_mostly_large);
            if (!check(snp_zero_words_async(ctl, 68, stream), "zero(redo list)")) return false;
            bool ok;
            if (prepass) {
                // lanes per block and LDS per wavefront, by the mean block size of the previous batch (GB/s, profiles/r02t_small_block_layouts.jsonl
                // and r02t_team_budget.jsonl: 32 B: one lane 558, 4 lanes 395; 64 B: 418 / 444; 128 B: 4 lanes 435, 8 lanes 340; 256 B: 4 lanes
                // 302, 8 lanes 351, 16 lanes 235; 384 B: 8 lanes 221 with 4.5 KiB of LDS per wavefront, 356 with 6.75 KiB (all eight blocks in
                // one round), 16 lanes 238; 512 B: 8 lanes 319 with 9 KiB, 16 lanes 237; 768-1024 B: teams 180-260, the wave kernel 335)
                const u32 lim = small_max > 512u ? 512u : small_max;
                u32 lay, budget = 0;
                if (small_lanes) lay = (small_max & 0x0fffffffu) | 0x80000000u;
                else if (small_team_log) lay = lim | (small_team_log << 28);
                else if (hint_mean_cap <= 48) lay = lim | 0x80000000u;
                else if (hint_mean_cap <= 144) lay = lim | (2u << 28);
                else {
                    lay = lim | (3u << 28);
                    const u32 want = (8u * (2u * (hint_mean_cap > 512u ? 512u : hint_mean_cap) + 64u) + 255u) & ~255u;   // eight blocks in one round
                    budget = want < 4608u ? 4608u : want > 9216u ? 9216u : want;
                }
                if (!check(snp_launch_decompress_small(d_in, in_off, in_len, nblocks, d_out, out_off, out_cap, out_len, status,
                                                       chunk_type, lay, stream, chains ? list : nullptr, ctl, sub_cap, budget), "decompress (small blocks) launch"))
                    return false;
                if (chains)
                    ok = check(snp_launch_decompress_list(d_in, in_off, in_len, nblocks, d_out, out_off, out_cap, out_len, status, chunk_type,
                                                          fenced | ((dec_lds / 256) << 8), stream, list, ctl, persistent_waves(), sub_cap), "decompress (list) launch");
                else
                    ok = check(snp_launch_decompress(d_in, in_off, in_len, nblocks, d_out, out_off, out_cap, out_len, status,
                                                     chunk_type, fenced | 16 | ((dec_lds / 256) << 8), stream, nullptr), "decompress launch");
            } else {
                ok = check(snp_launch_decompress(d_in, in_off, in_len, nblocks, d_out, out_off, out_cap, out_len, status,
                                                 chunk_type, fenced | ((dec_lds / 256) << 8), stream, nullptr), "decompress launch") &&
                     check(snp_launch_sample_caps(out_cap, nblocks, small_max, ctl, stream), "sample launch");
            }
            if (ok && !capturing && hint_ready() && !hint_pending) {     // how this batch went, for the next one
                hint_blocks = nblocks;
                hint_from_prepass = prepass;
                if (hipMemcpyAsync(hint, ctl, 68 * 4, hipMemcpyDeviceToHost, stream) == hipSuccess && hipEventRecord(hint_ev, stream) == hipSuccess)
                    hint_pending = true;
                else
                    (void)hipGetLastError();
            }
            return ok;
        }
        return check(snp_launch_decompress(d_in, in_off, in_len, nblocks, d_out, out_off, out_cap, out_len, status,
                                           chunk_type, fenced | ((dec_lds / 256) << 8), stream, nullptr), "decompress launch");
    }
    bool stream_is_capturing()
    {
        hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(stream, &cap_st) != hipSuccess || cap_st != hipStreamCaptureStatusNone;
        if (capturing) { (void)hipGetLastError(); was_captured = true; }
        return capturing;
    }
    u32* hint = nullptr;                                 // pinned: the previous batch's list length
    hipEvent_t hint_ev = nullptr;
    bool hint_pending = false, hint_mostly_large = false, hint_from_prepass = false, hint_seen = false;
    u32 hint_blocks = 0, hint_mean_cap = 256;            // (no history yet: assume 256-byte blocks)
    bool hint_ready()
    {
        if (!hint && hipHostMalloc(reinterpret_cast<void**>(&hint), 68 * 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); hint = nullptr; return false; }
        if (!hint_ev && hipEventCreateWithFlags(&hint_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); hint_ev = nullptr; return false; }
        return true;
    }
    u32 n_waves = 0;
    u32 persistent_waves()                               // one chip-full of 64-thread workgroups at 8 wavefronts per SIMD
    {
        if (!n_waves) {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
            n_waves = static_cast<u32>(cus) * 32u;
        }
        return n_waves;
    }

    // One launch of the compressor over nblocks fragments, picking the layout (see compress_lanes.hip).
    bool launch_compress(const u8* d_in, const u64* in_off, const u32* in_len, u32 nblocks, u8* d_out, const u64* out_off,
                         u32* out_len, i32* status, int emit_varint)
    {
        // Measured on MI355X (profiles/r02p_compress_by_batch.jsonl, r02_window_kernel.jsonl): the window kernel (LDS tables, 1024 fragments in flight)
        // runs at the same rate at any batch size and beats the single-token wave kernel everywhere; the lane kernel (HBM
        // tables) needs >= 16 384 fragments in flight before its memory-level parallelism overtakes it.
        // ... and between the two, from win_gtab_min fragments on, the window kernel keeps its u16 tables in a 256 MiB global-memory workspace that
        // stays in L2 / Infinity Cache instead of in LDS (compress_win.hip, WinTable): 32 wavefronts per CU instead of 4.
        const bool win = compress_mode == 3 || compress_mode == 4 || (compress_mode == 0 && nblocks < win_max);
        if (win) {
            const bool gtab = compress_mode == 4 || (compress_mode == 0 && nblocks >= win_gtab_min);
            uint16_t* tabs = nullptr;
            u32 slots = 0;
            if (gtab) {
                slots = persistent_waves();
                if (nblocks < slots) slots = nblocks;
                if (!ensure(win_tables, snp_compress_win_table_bytes(slots), "hipMalloc(window tables)")) return false;
                tabs = static_cast<uint16_t*>(win_tables.p);
            }
            return check(snp_launch_compress_win(d_in, in_off, in_len, nblocks, d_out, out_off, out_len, status, variant,
                                                 emit_varint, gtab ? 1 : win_np, stream, tabs, slots), "compress (windows) launch");
        }
        // 64 KiB of table per fragment in flight: very large batches (millions of small blocks) go in slices, so the
        // workspace stays <= 16 GiB; 262 144 fragments per launch still fill the chip many times over
        const bool capturing = stream_is_capturing();                   // (as in launch_decompress: a captured call neither reads nor leaves a hint)
        if (capturing) {
        } else if (chint && chint_ev && chint_pending && hipEventQuery(chint_ev) == hipSuccess) {
            chint_pending = false;
            chint_small = chint[0] != 0 && chint[0] <= 512;
            chint_mid = chint[0] != 0 && chint[0] <= 4096;
            chint_tiny = (chint[0] > 80 && chint[0] <= 768) ? static_cast<int>((chint[0] + 15u) >> 4) : 0;   // LDS slot of the input-in-LDS launch, in 16-byte units
        } else {
            (void)hipGetLastError();
        }
        const u32 kSlice = slice_fragments;   // (round 2 cut batches of small fragments into launches of 65 536; with the round-3 kernel 262 144 per launch is faster there too:
                                              //  256 B 43.5 -> 46.5 GB/s, 1 KiB 41.0 -> 49.0, profiles/r03p_small_compress_sweep.jsonl)
        if (!ensure(small, 256, "hipMalloc(scalars)") || !borrow_tables(nblocks < kSlice ? nblocks : kSlice)) return false;
        // Launch shape by fragment size: 64 KiB fragments want 64 fragments per wavefront and 262 144 per launch, fragments of at most
        // 512 bytes 32 per wavefront and 65 536 per launch (256-byte blocks 36 -> 43.5 GB/s, 64-byte 31 -> 35; 1 KiB and up prefer the
        // former: profiles/r02w_small_block_compress.jsonl).  What the fragments are like is known only on the device, so the longest
        // fragment of the previous launch comes back with an asynchronous 4-byte copy (read above, only once it has landed) and this
        // batch is assumed to be alike; results do not depend on it.
        // (bit 8: two speculative probes per trip whatever the batch size -- fragments of at most 4 KiB are latency-bound, not request-bound:
        //  1 KiB blocks 45.6 GB/s with one exchange probe, 48.3-49.0 with two probes, profiles/r03p_small_compress_sweep.jsonl)
        const int lanes_per_wave = (chint_small ? 32 : 0) | (chint_mid ? 256 : 0) | (chint_tiny << 9);   // bits 9-14: slot size of the launch with the input in LDS, which then goes first (compress_lanes.hip, SMALL)
        for (u32 first = 0; first < nblocks; first += kSlice) {
            const u32 cnt = nblocks - first < kSlice ? nblocks - first : kSlice;
            if (!check(snp_launch_compress_lanes(d_in, in_off + first, in_len + first, cnt, d_out, out_off + first,
                                                 out_len + first, status + first, variant, emit_varint, &tp,
                                                 static_cast<u32*>(small.p), stream, lanes_per_wave),
                       "compress (lanes) launch")) {
                return_tables();
                return false;
            }
        }
        return_tables();
        if (!capturing && chint_ready() && !chint_pending) {            // this batch's longest fragment, for the next one
            if (hipMemcpyAsync(chint, small.p, 4, hipMemcpyDeviceToHost, stream) == hipSuccess && hipEventRecord(chint_ev, stream) == hipSuccess)
                chint_pending = true;
            else
                (void)hipGetLastError();
        }
        return true;
    }
    u32* chint = nullptr;                                // pinned: the longest fragment of the previous lane-compressor launch
    hipEvent_t chint_ev = nullptr;
    bool chint_pending = false, chint_small = false, chint_mid = false;
    int chint_tiny = 0;
    bool chint_ready()
    {
        if (!chint && hipHostMalloc(reinterpret_cast<void**>(&chint), 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); chint = nullptr; return false; }
        if (!chint_ev && hipEventCreateWithFlags(&chint_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); chint_ev = nullptr; return false; }
        return true;
    }

    bool check(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return true;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    }
    bool ensure(DevBuf& b, size_t bytes, const char* what)
    {
        if (bytes <= b.cap) return true;
        if (stream_is_capturing()) {                                    // hipFree / hipMalloc would invalidate the capture: refuse and leave it intact
            err = std::string(what) + ": a workspace would have to grow while the stream is being captured -- make the same call once before the capture (or snp_ctx_reserve_compress)";
            return false;
        }
        if (b.p) {
            if (was_captured) kept.push_back(b.p);                      // a graph captured earlier may still hold this address: kept until snp_ctx_destroy
            else (void)hipFree(b.p);
        }
        b.p = nullptr;
        b.cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        if (!check(hipMalloc(&b.p, want), what)) return false;
        b.cap = want;
        return true;
    }
    // Host <-> device transfers of the host-pointer entry points: the caller's buffers are pageable (the reference's Span API)
    // and the runtime's own pageable path moves them at ~40 GB/s; a pinned-slice pipeline inside the library measured slower
    // (31 GB/s, profiles/r02b_host_api_rates.jsonl) and was removed.
    bool h2d(void* dev, const void* host, size_t n, const char* what)
    {
        return check(hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, stream), what);
    }
    bool d2h(void* host, const void* dev, size_t n, const char* what)
    {
        return check(hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, stream), what) && check(hipStreamSynchronize(stream), what);
    }

    // The hash-table workspace of the lane compressor (64 KiB per fragment in flight) belongs to the DEVICE (TablePool above): one allocation below
    // 1 GiB, above that up to 16 pieces chosen by PieceSearch (why: see there).  A context borrows it per launch sequence.
    TablePool* pool = nullptr;
    snp_table_pieces tp{};                               // the borrowed view (valid between borrow_tables and return_tables)
    bool borrowed = false;
    // Borrows the device's workspace for batches of up to nblocks fragments: builds or grows it if need be (thorough: snp_ctx_reserve_compress),
    // orders this context's stream behind the previous borrower.  return_tables() must follow the launches.
    bool was_captured = false;                           // a call of this context ran under stream capture: no workspace is freed before snp_ctx_destroy
    std::vector<void*> kept;
    DevBuf own_tables;                                   // SNP_OPT_TABLE_PROBE_TRIES = 1: a plain one-allocation workspace of this context's own (no pool, no search)
    bool borrow_tables(u32 nblocks, bool thorough = false)
    {
        if (table_tries_set && table_tries == 1) {
            const size_t bytes = snp_compress_lanes_workspace(nblocks);
            if (bytes > own_tables.cap) {
                if (stream_is_capturing()) { err = "hash-table workspace: it would have to grow while the stream is being captured"; return false; }
                (void)hipStreamSynchronize(stream);
                if (own_tables.p) { if (was_captured) kept.push_back(own_tables.p); else (void)hipFree(own_tables.p); }
                own_tables = DevBuf{};
                if (!check(hipMalloc(&own_tables.p, bytes + 4096), "hipMalloc(hash tables)")) { own_tables.p = nullptr; return false; }
                own_tables.cap = bytes + 4096;
            }
            tp = snp_table_pieces{};
            tp.p[0] = static_cast<u32*>(own_tables.p);
            tp.piece_frags = 0xffffffc0u;
            tp.n = 1;
            counters[2] = counters[3] = counters[4] = counters[5] = 0;
            return true;                                 // (borrowed stays false: return_tables has nothing to do)
        }
        TablePool& P = *pool;
        P.mu.lock();
        const bool capturing = stream_is_capturing();
        if (!build_tables(P, nblocks, thorough, capturing)) { P.mu.unlock(); return false; }
        if (capturing) {
            P.pinned = true;                             // the graph keeps the address: nothing this pool handed out is freed before the pool dies
        } else if (P.used && P.last_stream != stream) {
            if (!check(hipStreamWaitEvent(stream, P.last_use, 0), "hipStreamWaitEvent(table pool)")) { P.mu.unlock(); return false; }
        }
        tp = P.tp;
        counters[2] = P.stats[0]; counters[3] = P.stats[1]; counters[4] = P.stats[2]; counters[5] = P.stats[3];
        borrowed = true;
        return true;
    }
    void return_tables()
    {
        if (!borrowed) return;
        TablePool& P = *pool;
        if (!stream_is_capturing()) {                    // (a captured launch is ordered by its graph; see INTEGRATION.md "Inside a hipGraph")
            if (!P.last_use && hipEventCreateWithFlags(&P.last_use, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); P.last_use = nullptr; }
            if (P.last_use && hipEventRecord(P.last_use, stream) == hipSuccess) { P.used = true; P.last_stream = stream; }
            else { (void)hipGetLastError(); (void)hipStreamSynchronize(stream); P.used = false; }
        }
        borrowed = false;
        P.mu.unlock();
    }
    // (callers hold P.mu)
    bool build_tables(TablePool& P, u32 nblocks, bool thorough, bool capturing)
    {
        const size_t bytes = snp_compress_lanes_workspace(nblocks);
        if (!P.pieces.empty() && static_cast<uint64_t>(nblocks) <= static_cast<uint64_t>(P.tp.piece_frags) * P.tp.n) return true;
        if (P.pieces.empty() && P.plain && bytes <= P.plain_cap) return true;
        if (capturing) {                                 // (as in ensure(): no allocation, no search, no synchronisation inside a capture)
            err = "hash-table workspace: it would have to be built while the stream is being captured -- call snp_ctx_reserve_compress (or make the same call once) before the capture";
            return false;
        }
        // How far the placement search may go.  By default it holds at most TWO workspaces' worth of candidate pieces at once (one transient extra
        // workspace, a few hundred ms) and never more than half of what is free: a library must not take seconds or crowd a shared device on its own
        // account.  A caller that wants the thorough search -- device memory comes in regions of three kinds tens of GiB long,
        // and the third may lie 150 GB of allocations away -- says so: SNP_OPT_TABLE_PROBE_TRIES workspaces' worth (up to 24), within
        // SNP_OPT_TABLE_PROBE_MAX_BYTES, at start-up through snp_ctx_reserve_compress.
        const int tries = table_tries_set ? table_tries : 2;
        P.drop_workspace();                              // (waits for the previous borrower; a pinned pool keeps the old memory until it dies)
        P.stats[0] = P.stats[1] = P.stats[2] = P.stats[3] = 0;
        if (bytes < (1ull << 30) || tries <= 1) {
            // (a GiB-sized workspace gets no growth slack: 25 % of 10.7 GB is 2.7 GB that nothing ever uses)
            const size_t want = bytes >= (1ull << 30) ? bytes + 4096 : bytes + bytes / 4 + 4096;
            if (!check(hipMalloc(&P.plain, want), "hipMalloc(hash tables)")) { P.plain = nullptr; return false; }
            P.plain_cap = want;
            P.tp = snp_table_pieces{};
            P.tp.p[0] = static_cast<u32*>(P.plain);
            P.tp.piece_frags = 0xffffffc0u;
            P.tp.n = 1;
            return true;
        }
        PieceSearch ps{};
        // capacity = the batch + 1/16 of slack (at most one slice): a later batch of slightly more fragments must not repeat the search
        const uint64_t with_slack = static_cast<uint64_t>(nblocks) + nblocks / 16u;
        const u32 cap_frags = static_cast<u32>(with_slack < slice_fragments ? with_slack : (nblocks > slice_fragments ? nblocks : slice_fragments));
        const u32 piece_frags = ((cap_frags + SNP_TABLE_PIECES_MAX - 1) / SNP_TABLE_PIECES_MAX + 63u) / 64u * 64u;
        const size_t piece_bytes = static_cast<size_t>(piece_frags) * 65536u;
        ps.n = (cap_frags + piece_frags - 1) / piece_frags;
        ps.piece_gib = piece_bytes / 1073741824.0;
        ps.max_cand = static_cast<size_t>(ps.n) * static_cast<size_t>(tries);
        ps.dbg = SNP_GETENV("SNAPPIER_HIP_DEBUG") != nullptr;
        if (thorough && table_tries_set) ps.patience = 64;                   // the caller asked for it and has time: look for a third kind as far as max_cand allows
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {                // the candidates coexist: never more than half of what is free,
            size_t room = free_b / 2;                                         // ... unless the caller set a byte cap of its own (SNP_OPT_TABLE_PROBE_MAX_BYTES),
            if (table_probe_max_bytes) {                                      // which is honoured up to 7/8 of what is free: an explicit decision, not a default
                room = static_cast<size_t>(table_probe_max_bytes);
                if (room > free_b / 8 * 7) room = free_b / 8 * 7;
            }
            if (room / piece_bytes < ps.max_cand) ps.max_cand = room / piece_bytes;
        }
        if (ps.max_cand < ps.n) ps.max_cand = ps.n;                          // the workspace itself is not optional
        std::vector<u32*> cand;
        ps.alloc_one = [&]() {
            void* q = nullptr;
            if (hipMalloc(&q, piece_bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
            cand.push_back(static_cast<u32*>(q));
            return true;
        };
        ps.probe_set = [&](const std::vector<u32>& pick) {
            snp_table_pieces t{};
            for (size_t i = 0; i < pick.size(); ++i) t.p[i] = cand[pick[i]];
            t.piece_frags = piece_frags;
            t.n = static_cast<u32>(pick.size());
            float ms = 1e30f;
            if (snp_probe_tables(&t, nblocks, 512u, stream, &ms) != hipSuccess) { (void)hipGetLastError(); ms = 1e30f; }
            return ms;
        };
        std::vector<u32> set;
        const auto t_search = std::chrono::steady_clock::now();
        const float ms = ps.run(set);
        if (ms < 0) {
            for (u32* q : cand) (void)hipFree(q);
            err = "hipMalloc(hash tables): out of memory";
            return false;
        }
        std::vector<char> used(cand.size(), 0);
        P.tp = snp_table_pieces{};
        for (u32 i = 0; i < ps.n; ++i) { P.tp.p[i] = cand[set[i]]; used[set[i]] = 1; P.pieces.push_back(cand[set[i]]); }
        P.tp.piece_frags = piece_frags;
        P.tp.n = ps.n;
        for (size_t k = 0; k < cand.size(); ++k)
            if (!used[k]) (void)hipFree(cand[k]);
        P.stats[0] = ms < 1e6f ? static_cast<uint64_t>(ms * 1000.0f) : 0;     // (a probe that failed reports 1e30: the set then is whatever the arithmetic picked)
        P.stats[1] = static_cast<uint64_t>(cand.size());
        P.stats[2] = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_search).count());
        P.stats[3] = static_cast<uint64_t>(cand.size()) * piece_bytes;        // most bytes the search held at once (all candidates coexist until it ends)
        return true;
    }
    bool use_device() { return check(hipSetDevice(device), "hipSetDevice"); }
    // The context's scratch (hash tables, staging) is ordered by the stream it runs on.  Moving the context to another
    // stream: everything already queued on the old stream must finish before the new stream touches the scratch.
    bool rebind(hipStream_t next)
    {
        if (next == stream) return true;
        if (!order_ev && !check(hipEventCreateWithFlags(&order_ev, hipEventDisableTiming), "hipEventCreate")) return false;
        return check(hipEventRecord(order_ev, stream), "hipEventRecord") &&
               check(hipStreamWaitEvent(next, order_ev, 0), "hipStreamWaitEvent");
    }
    hipEvent_t order_ev = nullptr;
    // Host input of nf 64 KiB fragments -> this->in, compressed into d_out.  Large inputs go up in slices on a copy stream and
    // slice i is compressed while slice i+1 crosses PCIe (the fragments are independent; one launch per slice, small enough
    // for the LDS-table kernel).  1 GiB: 57 -> 44 ms (profiles/r02i_host_api_rates.jsonl).
    bool upload_and_compress(const u8* host_in, size_t n, u32 nf, const u64* d_in_off, const u32* d_in_len, u8* d_out,
                             const u64* d_out_off, u32* d_out_len, i32* d_status, int emit_varint)
    {
        if (nf < 4096 || !copy_stream_ready())
            return h2d(in.p, host_in, n, "H2D input") &&
                   launch_compress(static_cast<const u8*>(in.p), d_in_off, d_in_len, nf, d_out, d_out_off, d_out_len, d_status, emit_varint);
        const u32 per = nf >= 16384 ? 4096u : (nf + 3) / 4;
        // the copy stream overwrites this->in: everything already queued on `stream` that reads it goes first (correctness must not
        // rest on the previous call having synchronised)
        bool ok = check(hipEventRecord(copy_ev[0], stream), "event") && check(hipStreamWaitEvent(copy_stream, copy_ev[0], 0), "wait");
        u32 k = 0;
        for (u32 first = 0; first < nf && ok; first += per, ++k) {
            const u32 cnt = nf - first < per ? nf - first : per;
            const size_t off = static_cast<size_t>(first) * SNP_BLOCK_SIZE;
            const size_t len = n - off < static_cast<size_t>(cnt) * SNP_BLOCK_SIZE ? n - off : static_cast<size_t>(cnt) * SNP_BLOCK_SIZE;
            hipEvent_t ev = copy_ev[k & 1];
            ok = check(hipMemcpyAsync(static_cast<u8*>(in.p) + off, host_in + off, len, hipMemcpyHostToDevice, copy_stream), "H2D input") &&
                 check(hipEventRecord(ev, copy_stream), "event") && check(hipStreamWaitEvent(stream, ev, 0), "wait") &&
                 launch_compress(static_cast<const u8*>(in.p), d_in_off + first, d_in_len + first, cnt, d_out, d_out_off + first,
                                 d_out_len + first, d_status + first, emit_varint);
        }
        // a failed step returns to the caller, who may free or reuse host_in at once: no copy from it may still be in flight
        if (!ok) (void)hipStreamSynchronize(copy_stream);
        return ok;
    }
    // copy stream: host -> device slices that overlap the kernels of the previous slice
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[2] = {nullptr, nullptr};
    int copy_state = 0;
    bool copy_stream_ready()
    {
        if (copy_state) return copy_state > 0;
        copy_state = -1;
        if (hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); copy_stream = nullptr; return false; }
        for (auto& e : copy_ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
        copy_state = 1;
        return true;
    }
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

extern "C" {

// ---- context ---------------------------------------------------------------------------------------------

snp_status snp_ctx_create(int device, int hash_variant, void* stream, snp_ctx** out_ctx)
{
    if (!out_ctx || (hash_variant != SNP_HASH_CRC32C && hash_variant != SNP_HASH_MUL)) return SNP_ERR_BAD_ARG;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return SNP_ERR_DEVICE;
    snp_ctx* c = new (std::nothrow) snp_ctx();
    if (!c) return SNP_ERR_DEVICE;
    c->device = device;
    c->variant = hash_variant;
    DevGuard dg(c);
    if (!dg.ok) { delete c; return SNP_ERR_DEVICE; }
    if (stream) c->stream = static_cast<hipStream_t>(stream);
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SNP_ERR_DEVICE; }
        c->own_stream = true;
    }
    c->pool = pool_of(device);
    {
        std::lock_guard<std::mutex> g(c->pool->mu);
        ++c->pool->users;
    }
    // debug knobs: SNAPPIER_HIP_FENCED=1 drains vmcnt before reading young output; SNAPPIER_HIP_DECODE=serial
    // disables the token-parallel front end of the decompressor (bit 1 of the kernel mode)
    // FENCED (drain vmcnt before a wave reads output bytes it stored itself) is the default: measured 0.9 % slower than relying
    // on in-order vector memory (17.22 vs 17.38 ms per 10 GiB, profiles/r02c_fenced_ab.jsonl); SNAPPIER_HIP_FENCED=0 turns it off
    const char* f = SNP_GETENV("SNAPPIER_HIP_FENCED");
    c->fenced = (f && f[0] == '0') ? 0 : 1;
    const char* m = SNP_GETENV("SNAPPIER_HIP_DECODE");
    if (m && strcmp(m, "serial") == 0) c->fenced |= 2;
    if (m && strcmp(m, "batched") == 0) c->fenced |= 4;                 // token-parallel batches without the execution queue
    if (m && strcmp(m, "ring") == 0) c->fenced |= 8 | 32;              // sub-chain parse + output-granular execution through an LDS ring (FRONT = 4)
    else if (m && strcmp(m, "chains_r04") == 0) c->fenced |= 8 | 64;   // the round-4 form of the default decoder (A/B)
    else if (!m || strcmp(m, "chains") == 0 || (strcmp(m, "queued") != 0 && strcmp(m, "serial") != 0 && strcmp(m, "batched") != 0))
        c->fenced |= 8;                                                 // default: sub-chain parse (decompress.hip, FRONT = 3); "queued" = 64-byte windows + queue
    c->decode_layout = (m && (strcmp(m, "serial") == 0 || strcmp(m, "batched") == 0)) ? 1 : 0;
    // SNAPPIER_HIP_DEC_LDS=<bytes>: dynamic LDS per decode wavefront, an occupancy throttle (160 KiB / bytes blocks per CU)
    const char* dl = SNP_GETENV("SNAPPIER_HIP_DEC_LDS");
    c->dec_lds = dl ? (atoi(dl) / 256) * 256 : kDefaultDecLds;
    // SNAPPIER_HIP_COMPRESS=win|lanes pins the compressor layout (default: by batch size)
    const char* cm = SNP_GETENV("SNAPPIER_HIP_COMPRESS");
    c->compress_mode = (cm && strcmp(cm, "lanes") == 0) ? 2 : (cm && strcmp(cm, "wing") == 0) ? 4 : (cm && strncmp(cm, "win", 3) == 0) ? 3 : 0;
    const char* wg = SNP_GETENV("SNAPPIER_HIP_WIN_GTAB_MIN");
    if (wg) c->win_gtab_min = static_cast<u32>(strtoul(wg, nullptr, 10));
    const char* wn = SNP_GETENV("SNAPPIER_HIP_WIN_NP");
    if (wn) c->win_np = atoi(wn) == 2 ? 2 : 1;
    const char* fs = SNP_GETENV("SNAPPIER_HIP_FRAME_SCAN");
    c->frame_scan = (fs && strcmp(fs, "serial") == 0) ? 1 : 0;
    const char* sm = SNP_GETENV("SNAPPIER_HIP_SMALL_MAX");
    if (sm) c->small_max = static_cast<u32>(strtoul(sm, nullptr, 10));
    const char* sl = SNP_GETENV("SNAPPIER_HIP_SMALL");
    c->small_lanes = sl && strcmp(sl, "lanes") == 0;
    c->small_team_log = (sl && strcmp(sl, "team4") == 0) ? 2 : (sl && strcmp(sl, "team8") == 0) ? 3 : (sl && strcmp(sl, "team16") == 0) ? 4 : 0;
    const char* rg = SNP_GETENV("SNAPPIER_HIP_REDO");
    c->redo_grid = rg && strcmp(rg, "grid") == 0;
    c->redo_list = rg && strcmp(rg, "list") == 0;
    const char* sn = SNP_GETENV("SNAPPIER_HIP_SMALL_MIN");
    if (sn) c->small_min_blocks = static_cast<u32>(strtoul(sn, nullptr, 10));
    const char* sf = SNP_GETENV("SNAPPIER_HIP_SLICE");
    if (sf && atoi(sf) >= 4096) c->slice_fragments = static_cast<u32>(atoi(sf));
    const char* wm = SNP_GETENV("SNAPPIER_HIP_WIN_MAX");
    if (wm) c->win_max = static_cast<u32>(strtoul(wm, nullptr, 10));
    // SNAPPIER_HIP_PARALLEL_MIN=<bytes>: declared length from which snp_try_decompress splits ONE block into 64 KiB
    // fragments decoded in parallel (tag_index.hip); 0 = always one wavefront per block
    const char* tt = SNP_GETENV("SNAPPIER_HIP_TABLE_TRIES");
    if (tt) { c->table_tries = atoi(tt) < 1 ? 1 : atoi(tt) > 24 ? 24 : atoi(tt); c->table_tries_set = true; }
    const char* pm = SNP_GETENV("SNAPPIER_HIP_PARALLEL_MIN");
    if (pm) c->par_min = static_cast<u32>(strtoul(pm, nullptr, 10));
    *out_ctx = c;
    return SNP_OK;
}

uint64_t snp_ctx_counter(const snp_ctx* c, int which) { return (c && which >= 0 && which < 7) ? c->counters[which] : 0; }

snp_status snp_ctx_reserve_compress(snp_ctx* c, uint32_t nfragments)
{
    if (!c) return SNP_ERR_BAD_ARG;
    if (nfragments == 0) return SNP_OK;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    if (!c->borrow_tables(nfragments < c->slice_fragments ? nfragments : c->slice_fragments, true)) return SNP_ERR_DEVICE;
    c->return_tables();
    return SNP_OK;
}

snp_status snp_ctx_set_option(snp_ctx* c, int option, int64_t v)
{
    if (!c) return SNP_ERR_BAD_ARG;
    switch (option) {
        case SNP_OPT_DECODE_LAYOUT:
            if (v < 0 || v > 5) return SNP_ERR_BAD_ARG;
            c->no_prepass = v == 1;
            c->small_lanes = v == 2;
            c->small_team_log = v >= 3 ? static_cast<u32>(v - 1) : 0u;   // 3 / 4 / 5 -> teams of 4 / 8 / 16 lanes
            return SNP_OK;
        case SNP_OPT_SMALL_BLOCK_MAX:
            if (v < 0 || v > 0x0fffffff) return SNP_ERR_BAD_ARG;
            c->small_max = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_SMALL_BLOCK_MIN_BATCH:
            if (v < 1 || v > 0xffffffffll) return SNP_ERR_BAD_ARG;
            c->small_min_blocks = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_LAYOUT:
            if (v != 0 && v != 2 && v != 3 && v != 4) return SNP_ERR_BAD_ARG;
            c->compress_mode = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_MAX_BATCH:
            if (v < 0 || v > 0xffffffffll) return SNP_ERR_BAD_ARG;
            c->win_max = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_TABLE_PROBE_TRIES:
            if (v < 1 || v > 24) return SNP_ERR_BAD_ARG;
            c->table_tries = static_cast<int>(v);
            c->table_tries_set = true;
            return SNP_OK;
        case SNP_OPT_TABLE_PROBE_MAX_BYTES:
            if (v < 0) return SNP_ERR_BAD_ARG;
            c->table_probe_max_bytes = static_cast<uint64_t>(v);
            return SNP_OK;
        case SNP_OPT_PARALLEL_DECODE_MIN:
            if (v < 0 || v > 0x7fffffff) return SNP_ERR_BAD_ARG;
            c->par_min = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_FENCED:
            if (v != 0 && v != 1) return SNP_ERR_BAD_ARG;
            c->fenced = (c->fenced & ~1) | static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_DECODE_LEFTOVERS:
            if (v < 0 || v > 2) return SNP_ERR_BAD_ARG;
            c->redo_grid = v == 1;
            c->redo_list = v == 2;
            return SNP_OK;
        case SNP_OPT_CRC_KERNEL:
            if (v < 0 || v > 2) return SNP_ERR_BAD_ARG;
            c->crc_kernel = static_cast<int>(v);
            return SNP_OK;
        default:
            return SNP_ERR_BAD_ARG;
    }
}

snp_status snp_ctx_get_option(const snp_ctx* c, int option, int64_t* out)
{
    if (!c || !out) return SNP_ERR_BAD_ARG;
    switch (option) {
        case SNP_OPT_DECODE_LAYOUT: *out = c->no_prepass ? 1 : c->small_lanes ? 2 : c->small_team_log ? c->small_team_log + 1 : 0; return SNP_OK;
        case SNP_OPT_SMALL_BLOCK_MAX: *out = c->small_max; return SNP_OK;
        case SNP_OPT_SMALL_BLOCK_MIN_BATCH: *out = c->small_min_blocks; return SNP_OK;
        case SNP_OPT_COMPRESS_LAYOUT: *out = c->compress_mode; return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_MAX_BATCH: *out = c->win_max; return SNP_OK;
        case SNP_OPT_TABLE_PROBE_TRIES: *out = c->table_tries; return SNP_OK;
        case SNP_OPT_TABLE_PROBE_MAX_BYTES: *out = static_cast<int64_t>(c->table_probe_max_bytes); return SNP_OK;
        case SNP_OPT_PARALLEL_DECODE_MIN: *out = c->par_min; return SNP_OK;
        case SNP_OPT_FENCED: *out = c->fenced & 1; return SNP_OK;
        case SNP_OPT_DECODE_LEFTOVERS: *out = c->redo_grid ? 1 : c->redo_list ? 2 : 0; return SNP_OK;
        case SNP_OPT_CRC_KERNEL: *out = c->crc_kernel; return SNP_OK;
        default: return SNP_ERR_BAD_ARG;
    }
}

void snp_ctx_destroy(snp_ctx* c)
{
    if (!c) return;
    {
        DevGuard dg(c);
        (void)hipStreamSynchronize(c->stream);
        for (DevBuf* b : {&c->in, &c->out, &c->meta, &c->work, &c->fragtab, &c->scan, &c->small, &c->redo, &c->win_tables, &c->own_tables})
            if (b->p) (void)hipFree(b->p);
        for (void* q : c->kept) (void)hipFree(q);
        if (c->pool) {                                    // the device's last context takes the table pool with it
            std::lock_guard<std::mutex> g(c->pool->mu);
            if (--c->pool->users == 0) c->pool->destroy();
        }
        if (c->order_ev) (void)hipEventDestroy(c->order_ev);
        if (c->hint_ev) (void)hipEventDestroy(c->hint_ev);
        if (c->chint_ev) (void)hipEventDestroy(c->chint_ev);
        if (c->chint) (void)hipHostFree(c->chint);
        if (c->hint) (void)hipHostFree(c->hint);
        if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
        for (auto& e : c->copy_ev) if (e) (void)hipEventDestroy(e);
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

snp_status snp_ctx_set_stream(snp_ctx* c, void* stream)
{
    if (!c) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t next = static_cast<hipStream_t>(stream);
    if (c->own_stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
        c->own_stream = false;
    } else if (!c->rebind(next)) {
        return SNP_ERR_DEVICE;
    }
    c->stream = next;
    return SNP_OK;
}

const char* snp_ctx_last_error(const snp_ctx* c) { return c ? c->err.c_str() : "null context"; }

snp_status snp_ctx_synchronize(snp_ctx* c)
{
    if (!c) return SNP_ERR_BAD_ARG;
    return c->check(hipStreamSynchronize(c->stream), "hipStreamSynchronize") ? SNP_OK : SNP_ERR_DEVICE;
}

const char* snp_status_string(int s)
{
    switch (s) {
        case SNP_OK: return "ok";
        case SNP_ERR_OUTPUT_TOO_SMALL: return "Output buffer is too small.";
        case SNP_ERR_BAD_OFFSET: return "Invalid copy offset";
        case SNP_ERR_TOO_LONG: return "Data too long";
        case SNP_ERR_INCOMPLETE: return "Incomplete Snappy block.";
        case SNP_ERR_BAD_LENGTH: return "Invalid stream length";
        case SNP_ERR_CRC_MISMATCH: return "Chunk CRC mismatch.";
        case SNP_ERR_CHUNK_TYPE: return "Unknown chunk type";
        case SNP_ERR_OVERLAP: return "Input and output spans must not overlap.";
        case SNP_ERR_BAD_ARG: return "bad argument";
        case SNP_ERR_DEVICE: return "HIP device error";
        case SNP_ERR_TRUNCATED_STREAM: return "truncated framed stream";
        default: return "unknown status";
    }
}

const char* snp_version(void) { return "snappier_hip 0.1 (gfx950)"; }

// ---- host-only arithmetic --------------------------------------------------------------------------------

int64_t snp_max_fragment_compressed_length(int64_t n)   // Helpers.MaxCompressedLength  Helpers.cs:17-46
{
    if (n < 0) return -1;
    return 32 + n + n / 6 + 1;
}

int64_t snp_max_compressed_length(int64_t n)            // Snappy.GetMaxCompressedLength  Snappy.cs:20-24
{
    if (n < 0) return -1;
    const int64_t v = snp_max_fragment_compressed_length(n) + SNP_VARINT_MAX;
    return v > 0x7fffffffLL ? -1 : v;
}

// VarIntEncoding.TryReadSlow  VarIntEncoding.Read.cs:38-79 ; anything but Done is "Invalid stream length" (:16-24)
snp_status snp_get_uncompressed_length(const uint8_t* in, size_t n, uint32_t* out_len, uint32_t* out_header_bytes)
{
    if (!in && n) return SNP_ERR_BAD_ARG;
    u32 result = 0;
    int shift = 0;
    for (size_t i = 0; i < n; ++i) {
        const u8 c = in[i];
        const u32 val = c & 0x7fu;
        if (val & ~(0xffffffffu >> shift)) return SNP_ERR_BAD_LENGTH;
        result |= val << shift;
        shift += 7;
        if (c < 128) {
            if (out_len) *out_len = result;
            if (out_header_bytes) *out_header_bytes = static_cast<u32>(i + 1);
            return SNP_OK;
        }
        if (shift >= 32) return SNP_ERR_BAD_LENGTH;
    }
    return SNP_ERR_BAD_LENGTH;
}

int64_t snp_frame_max_encoded_length(int64_t n)
{
    if (n < 0) return -1;
    const int64_t chunks = (n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE;
    return SNP_STREAM_HEADER_LEN + chunks * SNP_CHUNK_HEADER_LEN + n;   // a chunk never grows (type 0x01 fallback)
}

// ---- batch, device pointers --------------------------------------------------------------------------------

snp_status snp_compress_batch(snp_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                              uint32_t nblocks, uint8_t* out, const uint64_t* out_off, uint32_t* out_len,
                              int32_t* status)
{
    if (!c || (nblocks && (!in || !in_off || !in_len || !out || !out_off || !out_len || !status))) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    return c->launch_compress(in, in_off, in_len, nblocks, out, out_off, out_len, status, 1) ? SNP_OK : SNP_ERR_DEVICE;
}

snp_status snp_decompress_batch(snp_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                                uint32_t nblocks, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap,
                                uint32_t* out_len, int32_t* status)
{
    if (!c || (nblocks && (!in || !in_off || !in_len || !out || !out_off || !out_cap || !out_len || !status)))
        return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    return c->launch_decompress(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, nullptr) ? SNP_OK
                                                                                                              : SNP_ERR_DEVICE;
}

snp_status snp_concat_batch(snp_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t nblocks,
                            uint8_t* out, const uint64_t* dst_off)
{
    if (!c || (nblocks && (!in || !in_off || !in_len || !out || !dst_off))) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    if (nblocks == 0) return SNP_OK;
    return c->check(snp_launch_gather(in, in_off, in_len, out, dst_off, nblocks, c->stream), "concat launch") ? SNP_OK : SNP_ERR_DEVICE;
}

snp_status snp_crc32c_batch(snp_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len,
                            uint32_t nblocks, int masked, uint32_t* out_crc)
{
    if (!c || (nblocks && (!in || !in_off || !in_len || !out_crc))) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    return c->check(snp_launch_crc32c(in, in_off, in_len, nblocks, (masked ? 1 : 0) | c->crc_bits(), out_crc, nullptr, nullptr, c->stream),
                    "crc32c launch") ? SNP_OK : SNP_ERR_DEVICE;
}

// workspace layout for snp_frame_encode_device (all sub-arrays 16-byte aligned)
struct FrameWork {
    u64 *in_off, *comp_off, *dst_off, *total;
    u32 *in_len, *comp_len, *payload, *crc;
    i32* status;
    u8 *type, *comp;
    u64 bytes;
};
static FrameWork frame_work_layout(void* base, u64 n)
{
    const u64 nc = (n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE;
    u8* p = static_cast<u8*>(base);
    u64 o = 0;
    FrameWork w{};
    auto take = [&](u64 bytes) { u8* r = p ? p + o : nullptr; o += align_up(bytes, 16); return r; };
    w.in_off = reinterpret_cast<u64*>(take(nc * 8));
    w.comp_off = reinterpret_cast<u64*>(take(nc * 8));
    w.dst_off = reinterpret_cast<u64*>(take((nc + 1) * 8));
    w.total = reinterpret_cast<u64*>(take(8));
    w.in_len = reinterpret_cast<u32*>(take(nc * 4));
    w.comp_len = reinterpret_cast<u32*>(take(nc * 4));
    w.payload = reinterpret_cast<u32*>(take(nc * 4));
    w.crc = reinterpret_cast<u32*>(take(nc * 4));
    w.status = reinterpret_cast<i32*>(take(nc * 4));
    w.type = take(nc);
    w.comp = take(nc * kCompStride);
    w.bytes = o;
    return w;
}

uint64_t snp_frame_encode_workspace(uint64_t n) { return frame_work_layout(nullptr, n).bytes; }

// host_in != nullptr: the raw stream is still in host memory; it is uploaded into d_in in slices that overlap the compressor
static snp_status frame_encode_impl(snp_ctx* c, const uint8_t* d_in, const uint8_t* host_in, uint64_t n, uint8_t* d_out,
                                    uint64_t cap, uint64_t* d_written, void* d_work)
{
    if (cap < SNP_STREAM_HEADER_LEN) return SNP_ERR_OUTPUT_TOO_SMALL;
    const u32 nc = static_cast<u32>((n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE);
    hipStream_t s = c->stream;
    if (nc == 0)
        return c->check(snp_launch_frame_header_only(d_out, d_written, s), "frame header") ? SNP_OK : SNP_ERR_DEVICE;
    const FrameWork w = frame_work_layout(d_work, n);
    bool ok = c->check(snp_launch_frame_chunks(n, nc, kCompStride, w.in_off, w.in_len, w.comp_off, s), "frame chunks");
    // CompressBlock: TryCompress(chunk) = varint + one fragment  (SnappyStreamCompressor.cs:206)
    if (host_in) ok = ok && c->upload_and_compress(host_in, n, nc, w.in_off, w.in_len, w.comp, w.comp_off, w.comp_len, w.status, 1);
    else ok = ok && c->launch_compress(d_in, w.in_off, w.in_len, nc, w.comp, w.comp_off, w.comp_len, w.status, 1);
    // masked CRC-32C of the RAW chunk  (:243-245,258-260)
    ok = ok && c->check(snp_launch_crc32c(d_in, w.in_off, w.in_len, nc, 1 | c->crc_bits(), w.crc, nullptr, nullptr, s), "frame crc");
    ok = ok && c->check(snp_launch_frame_plan(w.in_len, w.comp_len, nc, w.type, w.payload, w.dst_off, d_written, s),
                        "frame plan");
    ok = ok && c->check(snp_launch_frame_emit(d_in, w.in_off, w.comp, w.comp_off, w.type, w.payload, w.crc, w.dst_off,
                                              d_out, cap, nc, s), "frame emit");
    return ok ? SNP_OK : SNP_ERR_DEVICE;
}

snp_status snp_frame_encode_device(snp_ctx* c, const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap,
                                   uint64_t* d_written, void* d_work)
{
    if (!c || !d_out || !d_written || !d_work || (n && !d_in)) return SNP_ERR_BAD_ARG;
    if (n > 0xffffffffull * SNP_BLOCK_SIZE) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    return frame_encode_impl(c, d_in, nullptr, n, d_out, cap, d_written, d_work);
}

snp_status snp_frame_decode_chunks_device(snp_ctx* c, const uint8_t* d_in, const uint8_t* chunk_type,
                                          const uint64_t* body_off, const uint32_t* body_len,
                                          const uint32_t* chunk_crc, uint32_t nchunks, uint8_t* d_out,
                                          const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len,
                                          int32_t* status)
{
    if (!c || (nchunks && (!d_in || !chunk_type || !body_off || !body_len || !chunk_crc || !d_out || !out_off ||
                           !out_cap || !out_len || !status)))
        return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    bool ok = c->launch_decompress(d_in, body_off, body_len, nchunks, d_out, out_off, out_cap, out_len, status, chunk_type);
    // CRC over the produced bytes, compared with the chunk's stored masked CRC  (SnappyStreamDecompressor.cs:117-131)
    ok = ok && c->check(snp_launch_crc32c(d_out, out_off, out_len, nchunks, 1 | c->crc_bits(), nullptr, chunk_crc, status, s),
                        "frame crc verify");
    return ok ? SNP_OK : SNP_ERR_DEVICE;
}

// ---- framed stream without a chunk table: header walk on the device --------------------------------
// workspace: 64-byte header {total, tail status, chunks} ; body_off, out_off (u64) ; body_len, crc, out_cap, out_len (u32) ;
// status (i32) ; type (u8)
uint64_t snp_frame_decode_workspace(uint32_t max_chunks)
{
    return 64 + align_up(static_cast<u64>(max_chunks) * (8 * 2 + 4 * 5 + 1), 16) + 16;
}

snp_status snp_frame_decode_device(snp_ctx* c, const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap,
                                   uint32_t max_chunks, void* d_work, uint64_t* d_result)
{
    if (!c || !d_work || !d_result || (n && !d_in) || (cap && !d_out)) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    u64* hdr = static_cast<u64*>(d_work);
    u64* body_off = hdr + 8;
    u64* out_off = body_off + max_chunks;
    u32* body_len = reinterpret_cast<u32*>(out_off + max_chunks);
    u32* crc = body_len + max_chunks;
    u32* out_cap = crc + max_chunks;
    u32* out_len = out_cap + max_chunks;
    i32* status = reinterpret_cast<i32*>(out_len + max_chunks);
    u8* type = reinterpret_cast<u8*>(status + max_chunks);
    bool ok;
    if (c->frame_scan == 1) {
        ok = c->check(snp_launch_frame_scan(d_in, n, cap, max_chunks, type, body_off, body_len, crc, out_off, out_cap, hdr, s), "frame scan");
    } else {
        // per-span candidate tables live in context scratch (a few MB per 10 GiB of stream; grows on first use only)
        if (!c->ensure(c->scan, snp_frame_scan_workspace(n), "hipMalloc(frame scan)")) return SNP_ERR_DEVICE;
        ok = c->check(snp_launch_frame_scan_spans(d_in, n, cap, max_chunks, type, body_off, body_len, crc, out_off, out_cap, hdr,
                                                  c->scan.p, s), "frame scan (spans)");
    }
    if (ok && max_chunks && n) {                     // n == 0: no chunk, nothing to launch
        const snp_status st = snp_frame_decode_chunks_device(c, d_in, type, body_off, body_len, crc, max_chunks, d_out, out_off,
                                                             out_cap, out_len, status);
        if (st != SNP_OK) return st;
    }
    ok = ok && c->check(snp_launch_frame_result(status, hdr, d_result, s), "frame result");
    return ok ? SNP_OK : SNP_ERR_DEVICE;
}

// ---- single-buffer, host pointers --------------------------------------------------------------------------

static bool ranges_overlap(const u8* a, size_t an, const u8* b, size_t bn)
{
    return an && bn && a < b + bn && b < a + an;
}

}  // extern "C"

namespace {
// The input of a host-pointer call: one span (snp_try_compress / snp_try_decompress) or the segments of a ReadOnlySequence
// (snp_try_*_segments).  Segments go up one after the other into ONE device buffer: the managed side never flattens them.
struct HostSpans {
    const uint8_t* const* ptr;
    const size_t* len;
    uint32_t count;
    bool valid(size_t* total) const
    {
        size_t t = 0;
        for (uint32_t i = 0; i < count; ++i) {
            if (len[i] && !ptr[i]) return false;
            if (len[i] > 0xffffffffull || t + len[i] > 0xffffffffull) { *total = ~size_t{0}; return true; }
            t += len[i];
        }
        *total = t;
        return true;
    }
    bool upload(snp_ctx* c, void* dev) const
    {
        size_t at = 0;
        for (uint32_t i = 0; i < count; ++i) {
            if (len[i] && !c->h2d(static_cast<u8*>(dev) + at, ptr[i], len[i], "H2D input segment")) return false;
            at += len[i];
        }
        return true;
    }
    void head(uint8_t* dst, size_t want) const                            // the first `want` bytes (the varint preamble)
    {
        size_t got = 0;
        for (uint32_t i = 0; i < count && got < want; ++i)
            for (size_t k = 0; k < len[i] && got < want; ++k) dst[got++] = ptr[i][k];
    }
};

snp_status compress_spans(snp_ctx* c, const HostSpans& in_spans, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    const uint8_t* in = in_spans.count == 1 ? in_spans.ptr[0] : nullptr;
    *written = 0;
    if (n > 0xffffffffull) return SNP_ERR_BAD_ARG;                       // SnappyCompressor.cs:88-91
    if (cap == 0) return SNP_ERR_OUTPUT_TOO_SMALL;                        // Snappy.cs:57-62
    for (uint32_t i = 0; i < in_spans.count; ++i)
        if (ranges_overlap(in_spans.ptr[i], in_spans.len[i], out, cap)) return SNP_ERR_OVERLAP;   // SnappyCompressor.cs:27-30
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;

    u8 hdr[SNP_VARINT_MAX];                                               // VarIntEncoding.TryWrite  :34-37
    u32 hb = 0;
    for (u32 v = static_cast<u32>(n);;) {
        if (v < 128) { hdr[hb++] = static_cast<u8>(v); break; }
        hdr[hb++] = static_cast<u8>(v | 0x80);
        v >>= 7;
    }
    if (cap < hb) return SNP_ERR_OUTPUT_TOO_SMALL;
    const u32 nf = static_cast<u32>((n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE);
    if (nf == 0) { memcpy(out, hdr, hb); *written = hb; return SNP_OK; }

    hipStream_t s = c->stream;
    // meta: in_off, comp_off, dst_off (u64) ; in_len, comp_len (u32) ; status (i32)
    const u64 meta_bytes = static_cast<u64>(nf) * (8 * 3 + 4 * 3);
    if (!c->ensure(c->in, n, "hipMalloc(in)") || !c->ensure(c->work, nf * kCompStride, "hipMalloc(work)") ||
        !c->ensure(c->meta, meta_bytes, "hipMalloc(meta)"))
        return SNP_ERR_DEVICE;
    u64* d_in_off = static_cast<u64*>(c->meta.p);
    u64* d_comp_off = d_in_off + nf;
    u64* d_dst_off = d_comp_off + nf;
    u32* d_in_len = reinterpret_cast<u32*>(d_dst_off + nf);
    u32* d_comp_len = d_in_len + nf;
    i32* d_status = reinterpret_cast<i32*>(d_comp_len + nf);

    bool ok = c->check(snp_launch_frame_chunks(n, nf, kCompStride, d_in_off, d_in_len, d_comp_off, s), "fragment table");
    if (in)
        ok = ok && c->upload_and_compress(in, n, nf, d_in_off, d_in_len, static_cast<u8*>(c->work.p), d_comp_off, d_comp_len, d_status, 0);
    else
        ok = ok && in_spans.upload(c, c->in.p) &&
             c->launch_compress(static_cast<const u8*>(c->in.p), d_in_off, d_in_len, nf, static_cast<u8*>(c->work.p), d_comp_off, d_comp_len, d_status, 0);
    std::vector<u32> comp_len(nf);
    ok = ok && c->check(hipMemcpyAsync(comp_len.data(), d_comp_len, nf * 4ull, hipMemcpyDeviceToHost, s), "D2H lengths");
    ok = ok && c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;

    std::vector<u64> dst_off(nf);
    u64 total = 0;
    for (u32 f = 0; f < nf; ++f) { dst_off[f] = total; total += comp_len[f]; }
    if (cap - hb < total) return SNP_ERR_OUTPUT_TOO_SMALL;                 // SnappyCompressor.cs:63-68
    if (!c->ensure(c->out, total, "hipMalloc(out)")) return SNP_ERR_DEVICE;
    ok = c->check(hipMemcpyAsync(d_dst_off, dst_off.data(), nf * 8ull, hipMemcpyHostToDevice, s), "H2D offsets");
    ok = ok && c->check(snp_launch_gather(static_cast<const u8*>(c->work.p), d_comp_off, d_comp_len,
                                          static_cast<u8*>(c->out.p), d_dst_off, nf, s), "gather");
    ok = ok && c->d2h(out + hb, c->out.p, total, "D2H output");
    ok = ok && c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    memcpy(out, hdr, hb);
    *written = hb + total;
    return SNP_OK;
}

snp_status decompress_spans(snp_ctx* c, const HostSpans& in_spans, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    *written = 0;
    if (n > 0x7fffffffull) return SNP_ERR_BAD_ARG;                        // the reference's spans are int-length
    uint8_t in[8] = {0};                                                  // the preamble is read on the host
    in_spans.head(in, n < 5 ? n : 5);
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    const u32 cap32 = cap > 0x7fffffffull ? 0x7fffffffu : static_cast<u32>(cap);
    if (!c->ensure(c->in, n + 16, "hipMalloc(in)") || !c->ensure(c->out, static_cast<size_t>(cap32) + 16, "hipMalloc(out)") ||
        !c->ensure(c->meta, 64, "hipMalloc(meta)"))
        return SNP_ERR_DEVICE;
    struct Meta { u64 in_off, out_off; u32 in_len, out_cap, out_len; i32 status; } h{0, 0, static_cast<u32>(n), cap32, 0, 0};
    u8* m = static_cast<u8*>(c->meta.p);
    bool ok = c->check(hipMemcpyAsync(m, &h, sizeof(h), hipMemcpyHostToDevice, s), "H2D meta");

    // A large block: one wavefront per 64 KiB output fragment, fragment starts from the tag index (tag_index.hip).
    // Taken only for a clean preamble that fits the output; any fragment that does not come back OK (foreign streams
    // whose copies cross fragments, malformed data) sends the whole block to the single-wavefront decoder below.
    u32 expected = 0, hb = 0, shift = 0;
    bool clean = false;
    for (u32 i = 0; i < 5 && i < n; ++i) {                                // VarIntEncoding.Read.cs:38-79
        const u32 ch = in[i], val = ch & 0x7fu;
        if (val & ~(0xffffffffu >> shift)) break;
        expected |= val << shift;
        shift += 7;
        hb = i + 1;
        if (ch < 128) { clean = true; break; }
    }
    const bool large = clean && c->par_min && expected >= c->par_min && expected <= cap32 && n > hb;
    const u32 nf = large ? (expected + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE : 0;
    // fragment table: in_off, out_off (u64) ; in_len, out_cap, skip, out_len (u32) ; status (i32)
    if (large && (!c->ensure(c->work, snp_tag_index_workspace_bytes(static_cast<u32>(n), hb), "hipMalloc(tag index)") ||
                  !c->ensure(c->fragtab, static_cast<size_t>(nf) * (8 * 2 + 4 * 5), "hipMalloc(fragment table)")))
        return SNP_ERR_DEVICE;
    // The upload.  One large stream from one host buffer goes up in slices on the copy stream, and the tag index's per-chunk pass (most of its
    // time, and independent chunk by chunk) runs on each slice as it lands.
    bool indexed = false;
    if (ok && n && large && in_spans.count == 1 && n >= (8u << 20) && !snp_tag_index_look_back_only(static_cast<u32>(n), expected) && c->copy_stream_ready()) {
        const u8* const host_in = in_spans.ptr[0];
        const size_t slice = n / 8 > (4u << 20) ? (n / 8 + 4095) / 4096 * 4096 : (4u << 20);
        ok = c->check(snp_launch_tag_index_begin(static_cast<u64*>(c->work.p), static_cast<u32>(n), hb, 0, s), "tag index") &&
             c->check(hipEventRecord(c->copy_ev[0], s), "event") && c->check(hipStreamWaitEvent(c->copy_stream, c->copy_ev[0], 0), "wait");
        u32 done_chunks = 0, k = 0;
        for (size_t off = 0; off < n && ok; off += slice, ++k) {
            const size_t len = n - off < slice ? n - off : slice;
            hipEvent_t ev = c->copy_ev[k & 1];
            const u32 ready = snp_tag_index_chunks_ready(static_cast<u32>(n), hb, off + len);
            ok = c->check(hipMemcpyAsync(static_cast<u8*>(c->in.p) + off, host_in + off, len, hipMemcpyHostToDevice, c->copy_stream), "H2D input") &&
                 c->check(hipEventRecord(ev, c->copy_stream), "event") && c->check(hipStreamWaitEvent(s, ev, 0), "wait") &&
                 c->check(snp_launch_tag_index_chunks(static_cast<const u8*>(c->in.p), static_cast<u32>(n), hb, static_cast<u64*>(c->work.p),
                                                      done_chunks, ready - done_chunks, s), "tag index");
            done_chunks = ready;
        }
        if (!ok) (void)hipStreamSynchronize(c->copy_stream);              // (the caller may free its buffer at once: no copy from it may be in flight)
        indexed = ok;
    } else if (n) {
        ok = ok && in_spans.upload(c, c->in.p);
    }
    {
        if (ok && large) {
            const u32 nent = snp_tag_index_entries(static_cast<u32>(n), hb);
            u64* f_in_off = static_cast<u64*>(c->fragtab.p);
            u64* f_out_off = f_in_off + nf;
            u32* f_in_len = reinterpret_cast<u32*>(f_out_off + nf);
            u32* f_out_cap = f_in_len + nf;
            u32* f_skip = f_out_cap + nf;
            u32* f_out_len = f_skip + nf;
            i32* f_status = reinterpret_cast<i32*>(f_out_len + nf);
            ok = indexed ? c->check(snp_launch_tag_index_finish(static_cast<const u8*>(c->in.p), static_cast<u32>(n), hb, expected,
                                                                static_cast<u64*>(c->work.p), f_in_off, f_in_len, f_out_off, f_out_cap, f_skip, s),
                                    "tag index")
                         : c->check(snp_launch_tag_index(static_cast<const u8*>(c->in.p), static_cast<u32>(n), hb, expected,
                                                         static_cast<u64*>(c->work.p), f_in_off, f_in_len, f_out_off, f_out_cap, f_skip, s),
                                    "tag index");
            ok = ok && c->check(snp_launch_decompress(static_cast<const u8*>(c->in.p), f_in_off, f_in_len, nf,
                                                      static_cast<u8*>(c->out.p), f_out_off, f_out_cap, f_out_len, f_status,
                                                      nullptr, c->fenced | ((c->dec_lds / 256) << 8), s, f_skip),
                                "decompress fragments");
            std::vector<i32> st(nf);
            u32 looked_back = 0;
            ok = ok && c->check(hipMemcpyAsync(st.data(), f_status, nf * 4ull, hipMemcpyDeviceToHost, s), "D2H status");
            ok = ok && c->check(hipMemcpyAsync(&looked_back, static_cast<const u8*>(c->work.p) + snp_tag_index_fallback_offset(static_cast<u32>(n), hb), 4,
                                               hipMemcpyDeviceToHost, s), "D2H tag-index flag");
            ok = ok && c->check(hipStreamSynchronize(s), "sync");
            if (!ok) return SNP_ERR_DEVICE;
            bool all_ok = true;
            for (u32 f = 0; f < nf; ++f) all_ok = all_ok && st[f] == SNP_OK;
            ++c->counters[all_ok ? 0 : 1];
            if (looked_back) ++c->counters[6];
            if (!all_ok && SNP_GETENV("SNAPPIER_HIP_DEBUG")) {
                std::vector<u64> ent(nent), fo(nf);
                std::vector<u32> sk(nf), il(nf);
                (void)hipMemcpy(ent.data(), c->work.p, nent * 8ull, hipMemcpyDeviceToHost);
                (void)hipMemcpy(fo.data(), f_in_off, nf * 8ull, hipMemcpyDeviceToHost);
                (void)hipMemcpy(sk.data(), f_skip, nf * 4ull, hipMemcpyDeviceToHost);
                (void)hipMemcpy(il.data(), f_in_len, nf * 4ull, hipMemcpyDeviceToHost);
                fprintf(stderr, "[snappier] fragment decode fell back: n=%zu hb=%u expected=%u nent=%u last=(ip %u, op %u)\n", n, hb,
                        expected, nent, static_cast<u32>(ent[nent - 1]), static_cast<u32>(ent[nent - 1] >> 32) & 0x7fffffffu);
                for (u32 i = 0; i < nent && i < 12; ++i)
                    fprintf(stderr, "   entry %u: ip %u op %u\n", i, static_cast<u32>(ent[i]), static_cast<u32>(ent[i] >> 32) & 0x7fffffffu);
                u32 shown = 0;
                for (u32 f = 0; f < nf && shown < 8; ++f)
                    if (st[f] != SNP_OK) { fprintf(stderr, "   fragment %u: status %d in_off %llu in_len %u skip %u\n", f, st[f], (unsigned long long)fo[f], il[f], sk[f]); ++shown; }
            }
            if (all_ok) {
                ok = c->d2h(out, c->out.p, expected, "D2H output") &&
                     c->check(hipStreamSynchronize(s), "sync");
                if (!ok) return SNP_ERR_DEVICE;
                *written = expected;
                return SNP_OK;
            }
        }
    }
    ok = ok && c->check(snp_launch_decompress(static_cast<const u8*>(c->in.p), reinterpret_cast<u64*>(m + offsetof(Meta, in_off)),
                                              reinterpret_cast<u32*>(m + offsetof(Meta, in_len)), 1,
                                              static_cast<u8*>(c->out.p), reinterpret_cast<u64*>(m + offsetof(Meta, out_off)),
                                              reinterpret_cast<u32*>(m + offsetof(Meta, out_cap)),
                                              reinterpret_cast<u32*>(m + offsetof(Meta, out_len)),
                                              reinterpret_cast<i32*>(m + offsetof(Meta, status)), nullptr, c->fenced, s, nullptr),
                        "decompress");
    ok = ok && c->check(hipMemcpyAsync(&h, m, sizeof(h), hipMemcpyDeviceToHost, s), "D2H meta");
    ok = ok && c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    if (h.status != SNP_OK) return static_cast<snp_status>(h.status);
    if (h.out_len) {
        ok = c->d2h(out, c->out.p, h.out_len, "D2H output") &&
             c->check(hipStreamSynchronize(s), "sync");
        if (!ok) return SNP_ERR_DEVICE;
    }
    *written = h.out_len;
    return SNP_OK;
}
}  // namespace

extern "C" {

snp_status snp_try_compress(snp_ctx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    if (!c || !written || (n && !in) || (cap && !out)) return SNP_ERR_BAD_ARG;
    const HostSpans one{&in, &n, 1};
    return compress_spans(c, one, n, out, cap, written);
}

snp_status snp_try_decompress(snp_ctx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    if (!c || !written || (n && !in) || (cap && !out)) return SNP_ERR_BAD_ARG;
    const HostSpans one{&in, &n, 1};
    return decompress_spans(c, one, n, out, cap, written);
}

snp_status snp_try_compress_segments(snp_ctx* c, const uint8_t* const* seg, const size_t* seg_len, uint32_t nseg, uint8_t* out, size_t cap,
                                     size_t* written)
{
    if (!c || !written || (nseg && (!seg || !seg_len)) || (cap && !out)) return SNP_ERR_BAD_ARG;
    *written = 0;
    const HostSpans spans{seg, seg_len, nseg};
    size_t n = 0;
    if (!spans.valid(&n)) return SNP_ERR_BAD_ARG;
    if (n == ~size_t{0}) return SNP_ERR_BAD_ARG;                          // >= 2^32 bytes in all  SnappyCompressor.cs:88-91
    return compress_spans(c, spans, n, out, cap, written);
}

snp_status snp_try_decompress_segments(snp_ctx* c, const uint8_t* const* seg, const size_t* seg_len, uint32_t nseg, uint8_t* out, size_t cap,
                                       size_t* written)
{
    if (!c || !written || (nseg && (!seg || !seg_len)) || (cap && !out)) return SNP_ERR_BAD_ARG;
    *written = 0;
    const HostSpans spans{seg, seg_len, nseg};
    size_t n = 0;
    if (!spans.valid(&n) || n == ~size_t{0}) return SNP_ERR_BAD_ARG;
    return decompress_spans(c, spans, n, out, cap, written);
}

snp_status snp_crc32c(snp_ctx* c, const uint8_t* in, size_t n, int masked, uint32_t* out_crc)
{
    if (!c || !out_crc || (n && !in)) return SNP_ERR_BAD_ARG;
    if (n > 0xffffffffull) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    if (!c->ensure(c->in, n + 16, "hipMalloc(in)") || !c->ensure(c->meta, 64, "hipMalloc(meta)")) return SNP_ERR_DEVICE;
    struct Meta { u64 off; u32 len, crc; } h{0, static_cast<u32>(n), 0};
    u8* m = static_cast<u8*>(c->meta.p);
    bool ok = c->check(hipMemcpyAsync(m, &h, sizeof(h), hipMemcpyHostToDevice, s), "H2D meta");
    if (n) ok = ok && c->h2d(c->in.p, in, n, "H2D input");
    ok = ok && c->check(snp_launch_crc32c(static_cast<const u8*>(c->in.p), reinterpret_cast<u64*>(m), reinterpret_cast<u32*>(m + 8),
                                          1, (masked ? 1 : 0) | c->crc_bits(), reinterpret_cast<u32*>(m + 12), nullptr, nullptr, s), "crc32c");
    ok = ok && c->check(hipMemcpyAsync(&h, m, sizeof(h), hipMemcpyDeviceToHost, s), "D2H meta");
    ok = ok && c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    *out_crc = h.crc;
    return SNP_OK;
}

snp_status snp_frame_encode(snp_ctx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    if (!c || !written || (n && !in) || (cap && !out)) return SNP_ERR_BAD_ARG;
    *written = 0;
    if (ranges_overlap(in, n, out, cap)) return SNP_ERR_OVERLAP;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    const u64 max_out = static_cast<u64>(snp_frame_max_encoded_length(static_cast<int64_t>(n)));
    const u64 wbytes = snp_frame_encode_workspace(n);
    if (!c->ensure(c->in, n + 16, "hipMalloc(in)") || !c->ensure(c->out, max_out + 16, "hipMalloc(out)") ||
        !c->ensure(c->work, wbytes + 16, "hipMalloc(work)") || !c->ensure(c->meta, 64, "hipMalloc(meta)"))
        return SNP_ERR_DEVICE;
    bool ok = true;
    if (n > 0xffffffffull * SNP_BLOCK_SIZE) return SNP_ERR_BAD_ARG;
    snp_status st = frame_encode_impl(c, static_cast<const u8*>(c->in.p), n ? in : nullptr, n, static_cast<u8*>(c->out.p), max_out,
                                      static_cast<u64*>(c->meta.p), c->work.p);
    if (st != SNP_OK) return st;
    u64 total = 0;
    ok = c->check(hipMemcpyAsync(&total, c->meta.p, 8, hipMemcpyDeviceToHost, s), "D2H total") &&
         c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    if (total > cap) return SNP_ERR_OUTPUT_TOO_SMALL;
    ok = c->d2h(out, c->out.p, total, "D2H output") &&
         c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    *written = total;
    return SNP_OK;
}

// Host-side walk over chunk headers only (SnappyStreamDecompressor.ReadChunkHeader  :215-254): 4 bytes per chunk.
struct ChunkScan {
    std::vector<u8> type;
    std::vector<u64> body_off, out_off;
    std::vector<u32> body_len, crc, out_cap;
    u64 total = 0;
    snp_status tail = SNP_OK;   // error met after the chunks listed above (they are still decoded and checked first)
};
static inline u64 snp_max_expansion(u64 body_bytes) { return (body_bytes / 3 + 1) * 64; }
static void scan_chunks(const u8* in, size_t n, ChunkScan& cs)
{
    size_t ip = 0;
    while (ip < n) {
        if (n - ip < 4) { cs.tail = SNP_ERR_TRUNCATED_STREAM; return; }
        const u32 type = in[ip];
        const u32 size = in[ip + 1] | (in[ip + 2] << 8) | (static_cast<u32>(in[ip + 3]) << 16);   // :64-65
        ip += 4;
        if (n - ip < size) { cs.tail = SNP_ERR_TRUNCATED_STREAM; return; }
        if (type == 0x00 || type == 0x01) {
            if (size < 4) { cs.tail = SNP_ERR_TRUNCATED_STREAM; return; }
            u32 crc;
            memcpy(&crc, in + ip, 4);                                    // ReadChunkCrc  :260-289
            u32 dec = size - 4, hb = 0;
            if (type == 0x00 && snp_get_uncompressed_length(in + ip + 4, size - 4, &dec, &hb) != SNP_OK) {
                cs.tail = SNP_ERR_BAD_LENGTH;
                return;
            }
            if (dec > 0x7fffffffu) { cs.tail = SNP_ERR_BAD_LENGTH; return; }
            // No tag expands more than 3 bytes -> 64 (a copy-2 of length 64): a chunk that declares more than its body can
            // possibly produce is "Incomplete Snappy block." whatever its tags say -- and must not size any allocation.
            if (type == 0x00 && dec > snp_max_expansion(size - 4 - hb)) { cs.tail = SNP_ERR_INCOMPLETE; return; }
            cs.type.push_back(static_cast<u8>(type));
            cs.body_off.push_back(ip + 4);
            cs.body_len.push_back(size - 4);
            cs.crc.push_back(crc);
            cs.out_off.push_back(cs.total);
            cs.out_cap.push_back(dec);
            cs.total += dec;
        } else if (type < 0x80) {                                        // :182-185
            cs.tail = SNP_ERR_CHUNK_TYPE;
            return;
        }                                                                // 0x80..0xff skipped unvalidated  :187-196
        ip += size;
    }
}

snp_status snp_frame_decoded_length(const uint8_t* in, size_t n, uint64_t* out_len)
{
    if (!out_len || (n && !in)) return SNP_ERR_BAD_ARG;
    ChunkScan cs;
    scan_chunks(in, n, cs);
    *out_len = cs.total;
    return cs.tail;
}

snp_status snp_frame_decode(snp_ctx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    if (!c || !written || (n && !in) || (cap && !out)) return SNP_ERR_BAD_ARG;
    *written = 0;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    ChunkScan cs;
    scan_chunks(in, n, cs);
    const u32 nc = static_cast<u32>(cs.type.size());
    if (cs.total > cap) return SNP_ERR_OUTPUT_TOO_SMALL;
    if (nc == 0) return cs.tail;
    hipStream_t s = c->stream;
    // meta: body_off, out_off (u64) ; body_len, crc, out_cap, out_len (u32) ; status (i32) ; type (u8)
    const u64 meta_bytes = static_cast<u64>(nc) * (8 * 2 + 4 * 5 + 1) + 64;
    if (!c->ensure(c->in, n + 16, "hipMalloc(in)") || !c->ensure(c->out, cs.total + 16, "hipMalloc(out)") ||
        !c->ensure(c->meta, meta_bytes, "hipMalloc(meta)"))
        return SNP_ERR_DEVICE;
    u64* d_body_off = static_cast<u64*>(c->meta.p);
    u64* d_out_off = d_body_off + nc;
    u32* d_body_len = reinterpret_cast<u32*>(d_out_off + nc);
    u32* d_crc = d_body_len + nc;
    u32* d_out_cap = d_crc + nc;
    u32* d_out_len = d_out_cap + nc;
    i32* d_status = reinterpret_cast<i32*>(d_out_len + nc);
    u8* d_type = reinterpret_cast<u8*>(d_status + nc);
    bool ok = c->h2d(c->in.p, in, n, "H2D input");
    ok = ok && c->check(hipMemcpyAsync(d_body_off, cs.body_off.data(), nc * 8ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_out_off, cs.out_off.data(), nc * 8ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_body_len, cs.body_len.data(), nc * 4ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_crc, cs.crc.data(), nc * 4ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_out_cap, cs.out_cap.data(), nc * 4ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_type, cs.type.data(), nc, hipMemcpyHostToDevice, s), "H2D meta");
    if (!ok) return SNP_ERR_DEVICE;
    snp_status st = snp_frame_decode_chunks_device(c, static_cast<const u8*>(c->in.p), d_type, d_body_off, d_body_len,
                                                   d_crc, nc, static_cast<u8*>(c->out.p), d_out_off, d_out_cap, d_out_len,
                                                   d_status);
    if (st != SNP_OK) return st;
    std::vector<i32> status(nc);
    ok = c->check(hipMemcpyAsync(status.data(), d_status, nc * 4ull, hipMemcpyDeviceToHost, s), "D2H status") &&
         c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    for (u32 i = 0; i < nc; ++i)                                          // first failing chunk in stream order wins,
        if (status[i] != SNP_OK) return static_cast<snp_status>(status[i]);   // as the sequential reference would throw
    if (cs.tail != SNP_OK) return cs.tail;
    if (cs.total) {
        ok = c->d2h(out, c->out.p, cs.total, "D2H output") &&
             c->check(hipStreamSynchronize(s), "sync");
        if (!ok) return SNP_ERR_DEVICE;
    }
    *written = cs.total;
    return SNP_OK;
}

}  // extern "C"
