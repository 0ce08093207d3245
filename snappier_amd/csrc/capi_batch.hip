// capi_batch.hip -- the device-pointer batch entry points (snp_compress_batch, snp_decompress_batch, snp_crc32c_batch, snp_concat_batch) and the
// launch POLICY behind them: which decoder / compressor layout a batch gets (by its size, by what the context's previous batch looked like, or
// as pinned through snp_ctx_set_option).  These calls only enqueue kernels on the context's stream and may be captured into a hipGraph.
#include "capi_internal.h"

#ifndef SNP_W_DUAL_LDS_PER_CU
#define SNP_W_DUAL_LDS_PER_CU 4u   // dual form: persistent workgroups of the LDS form per CU (LDS holds four 35 KiB tables)
#endif

// One launch sequence of the decompressor over nblocks blocks (decompress_small.hip, then decompress.hip).
bool snp_ctx::launch_decompress(const u8* d_in, const u64* in_off, const u32* in_len, u32 nblocks, u8* d_out, const u64* out_off,
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

bool snp_ctx::hint_ready()
{
    if (!hint && hipHostMalloc(reinterpret_cast<void**>(&hint), 68 * 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); hint = nullptr; return false; }
    if (!hint_ev && hipEventCreateWithFlags(&hint_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); hint_ev = nullptr; return false; }
    return true;
}

u32 snp_ctx::persistent_waves()                               // one chip-full of 64-thread workgroups at 8 wavefronts per SIMD
{
    if (!n_waves) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
        n_waves = static_cast<u32>(cus) * 32u;
    }
    return n_waves;
}

// One launch of the compressor over nblocks fragments, picking the layout (see compress_lanes.hip).
bool snp_ctx::launch_compress(const u8* d_in, const u64* in_off, const u32* in_len, u32 nblocks, u8* d_out, const u64* out_off,
                     u32* out_len, i32* status, int emit_varint)
{
    // Measured on MI355X (profiles/r06z_compress_by_batch.jsonl): the per-wavefront kernel with its table in LDS (4 fragments in flight per CU) runs at the
    // same 34-35 GB/s at any batch size; the lane kernel (HBM tables) needs its 31-37 ms whatever the count up to ~20 000 fragments and overtakes
    // everything else from 32 768 on.  In between the per-wavefront kernel runs in its DUAL form (below); its global-slot form alone (layout 4) keeps
    // 12 wavefronts per CU, few enough that their 32 KiB slots stay cache resident.
    const bool win = compress_mode == 3 || compress_mode == 4 || compress_mode == 5 || (compress_mode == 0 && nblocks < win_max);
    if (win) {
        // From win_dual_min fragments on BOTH forms at once, on two streams, drawing fragments from one ticket counter (compress_win.hip, dual form:
        // 45-51 GB/s from 6 144 fragments up against 34.5 / 42-45 for either alone).  A context whose side stream cannot be created (or a first use under capture) keeps the single form.
        if ((compress_mode == 5 || (compress_mode == 0 && nblocks >= win_dual_min)) && win_np == 1 && (side_state > 0 || (!stream_is_capturing() && side_stream_ready()))) {
            const u32 per_cu = persistent_waves() / 32u;                 // (= CUs)
            u32 slots = win_gslots ? win_gslots : per_cu * 10u;
            if (!win_gslots && nblocks / 4u * 3u < slots) slots = nblocks / 4u * 3u;   // small batches: three quarters of the fragments' worth (3 072 fragments: 42.7 GB/s with 1 536-2 304 slots, 35.7 LDS form alone)
            if (nblocks < slots) slots = nblocks;
            if (slots == 0) slots = 1;
            if (!ensure(win_tables, snp_compress_win_table_bytes(slots), "hipMalloc(window tables)") || !ensure(small, 256, "hipMalloc(scalars)")) return false;
            return check(snp_launch_compress_win_dual(d_in, in_off, in_len, nblocks, d_out, out_off, out_len, status, variant, emit_varint, stream, side_stream,
                                                      side_ev[0], side_ev[1], static_cast<uint16_t*>(win_tables.p), slots, per_cu * SNP_W_DUAL_LDS_PER_CU,
                                                      static_cast<u32*>(small.p) + 16), "compress (windows, dual) launch");
        }
        const bool gtab = compress_mode == 4 || (compress_mode == 0 && nblocks >= win_gtab_min);
        uint16_t* tabs = nullptr;
        u32 slots = 0;
        if (gtab) {
            slots = win_gslots ? win_gslots : persistent_waves() / 32u * 12u;
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
    snp_lane_tuning tune = lane_tune;                                   // the caller's SNP_OPT_COMPRESS_LANE_* settings, plus what the previous batch taught:
    tune.hint = (chint_small ? 32 : 0) | (chint_mid ? 256 : 0) | (chint_tiny << 9);   // bits 9-14: slot size of the launch with the input in LDS, which then goes first (compress_lanes.hip, SMALL)
    for (u32 first = 0; first < nblocks; first += kSlice) {
        const u32 cnt = nblocks - first < kSlice ? nblocks - first : kSlice;
        if (!check(snp_launch_compress_lanes(d_in, in_off + first, in_len + first, cnt, d_out, out_off + first,
                                             out_len + first, status + first, variant, emit_varint, &tp,
                                             static_cast<u32*>(small.p), stream, &tune),
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

bool snp_ctx::side_stream_ready()
{
    if (side_state) return side_state > 0;
    side_state = -1;
    RelaxedCaptureMode relaxed;
    if (hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); side_stream = nullptr; return false; }
    for (auto& e : side_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    side_state = 1;
    return true;
}

bool snp_ctx::chint_ready()
{
    if (!chint && hipHostMalloc(reinterpret_cast<void**>(&chint), 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); chint = nullptr; return false; }
    if (!chint_ev && hipEventCreateWithFlags(&chint_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); chint_ev = nullptr; return false; }
    return true;
}

extern "C" {

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

}  // extern "C"
