// capi_pool.hip -- the lane compressor's hash-table workspace: ONE per device (TablePool, capi_internal.h), borrowed by every context on it per
// launch sequence; built by the first large compress call or snp_ctx_reserve_compress, placed by PieceSearch (piece_search.h: why pieces).
// The reference pools one table per compressor (Snappier/Internal/HashTable.cs:22-55).
#include "capi_internal.h"
#include "piece_search.h"

using snp_piece_search::PieceSearch;   // piece_search.h: why the workspace is made of pieces, and how they are chosen

void TablePool::drain()
{
    // (an event wait of this thread: legal in relaxed mode while ANOTHER thread captures in global mode -- ADVICE r5)
    RelaxedCaptureMode relaxed;
    if (used && last_use) (void)hipEventSynchronize(last_use);
}

void TablePool::drop_workspace()                         // callers hold mu
{
    drain();
    RelaxedCaptureMode relaxed;
    auto gone = [&](void* q) { if (!q) return; if (pinned) retired.push_back(q); else (void)hipFree(q); };
    gone(plain);
    for (void* q : pieces) gone(q);
    plain = nullptr;
    plain_cap = 0;
    pieces.clear();
    tp = snp_table_pieces{};
}

void TablePool::destroy()                                // the device's last context is gone
{
    pinned = false;
    drop_workspace();
    for (void* q : retired) (void)hipFree(q);
    retired.clear();
    if (last_use) (void)hipEventDestroy(last_use);
    last_use = nullptr;
    used = false;
}

namespace {
std::mutex g_pools_mu;
}
// Pools are keyed by the real device ordinal; an ordinal beyond the table has no pool (snp_ctx_create answers SNP_ERR_DEVICE) -- two devices never
// share a workspace.
TablePool* snp_pool_of(int device)
{
    static TablePool* pools[kSnpMaxDevices] = {};
    if (device < 0 || device >= kSnpMaxDevices) return nullptr;
    std::lock_guard<std::mutex> g(g_pools_mu);
    if (!pools[device]) pools[device] = new TablePool();
    return pools[device];
}

bool snp_ctx::borrow_tables(u32 nblocks, bool thorough)
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

void snp_ctx::return_tables()
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
bool snp_ctx::build_tables(TablePool& P, u32 nblocks, bool thorough, bool capturing)
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
    // (allocations, probe launches and event waits of THIS thread: relaxed mode keeps them from invalidating a hipStreamCaptureModeGlobal capture
    //  another thread of the process has in progress on its own stream -- the pool is shared across contexts, so that interaction is real: ADVICE r5)
    RelaxedCaptureMode relaxed;
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
