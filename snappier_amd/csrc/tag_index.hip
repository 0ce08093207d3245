// tag_index.hip -- finds where the 64 KiB output fragments of ONE large Snappy block begin in its compressed stream,
// so that the block can be decoded by one wavefront per fragment instead of one wavefront for the whole block
// (SnappyDecompressor.DecompressAllTags, SnappyDecompressor.cs:184-347, is a single serial tag walk; a compressor that
// follows SnappyCompressor.cs:34-80 restarts its table every 65 536 input bytes, so no copy of fragment f reads output
// of fragment f-1 -- the fragments are independent once their starting tags are known).
//
// A Snappy stream carries no index: the start of tag k+1 is only known after tag k is decoded.  But "the tag that would
// start at byte j" can be decoded for EVERY byte j at once, giving next[j] (start of the following tag) and len[j]
// (output bytes the tag produces).  Walking a chain of next-pointers is pointer jumping:
//   1. one workgroup per 16 KiB chunk of the stream builds (len, next) for its 16 384 positions in LDS and runs 12
//      rounds of in-place pointer doubling, restricted to jumps that stay inside a 4 KiB sub-chunk -- afterwards
//      entry j holds (output bytes, first tag start at or after the end of j's sub-chunk) for a walk that enters at j;
//   2. the only serial step is a decoupled look-back: workgroup k waits for the true entry point of its chunk from
//      workgroup k-1 (one 64-bit word: position | output offset), follows it through its four sub-chunks (four LDS
//      reads), records the four entry points, and publishes the entry of chunk k+1.  Workgroups take their chunk
//      number from a ticket counter, so a workgroup only ever waits for one that started earlier.
// Round 5: step 2 as written above is a chain of ~0.84 us hand-offs, one per 16 KiB of stream -- 15.6 ms per GiB of output, twelve times the
// fragment decode itself.  It is now the FALLBACK.  The default takes the chain off the workgroups:
//   2a. k_tag_cand: workgroup k, with the same table in LDS, follows the walks that enter its chunk at its first 128 bytes: after 16 KiB they
//       have merged, and their (usually one) landing point in chunk k + 1 is a CANDIDATE entry of that chunk, handed to the neighbour only (no
//       chain: every workgroup publishes before it waits).  For each of its own <= 8 candidates it records what step 2 would have recorded:
//       the sub-chunk entry points and the exit, output bytes counted from the entry.
//   2b. k_tag_scan: ONE workgroup.  A chunk is a function  row -> (row of the next chunk | a position further on | end, output bytes); functions
//       compose: every thread evaluates its run of chunks for each row, thread 0 chains the runs, every thread replays its run and writes the entries.
//   2c. k_tag_fix: a landing that is no candidate (a literal longer than a chunk -- incompressible fragments -- lands in the middle of a chunk whose
//       own walks started in its body) stops the scan; one workgroup follows the true chain from there, giving each landing its row (from the tag's
//       own bytes when it leaves its chunk by itself, from the chunk's table otherwise) until the chain is on a candidate again; the scan runs
//       again (its run functions are cached: only runs that stopped at the landing are evaluated anew).  Same kernel, same workgroup, one pass per
//       incompressible region; on anything irregular, or when the passes would cost more than it (one per 417 chunks), the look-back kernel runs.
// The entry table (one entry per 4 KiB of compressed data) is searched per fragment by k_fragment_starts; the
// fragment decoder (k_decompress<.., FRAG = true>) then parses at most 4 KiB of tags before its fragment begins.
// Anything that is not a well-formed stream ending exactly at (n, declared length) marks the table irregular and the
// caller falls back to the single-wavefront decoder, which owns the error semantics.
#include "snp_device.h"

namespace {

constexpr u32 kChunk = 16384;
constexpr u32 kThreads = 1024;                    // one workgroup per CU (128 KiB of LDS): sixteen wavefronts hide the LDS latency of the pointer doubling
constexpr u32 kSub = 4096;
constexpr u32 kSubs = kChunk / kSub;
constexpr u32 kRounds = 12;                       // a 4 KiB sub-chunk holds at most 2048 tags: 2^11 hops
constexpr u64 kValid = 1ull << 63;
constexpr u32 kBadIp = 0xffffffffu;               // entry: the stream is irregular from here on
constexpr u32 kFar = 0xfffffffeu;                 // table: next-pointer not representable / tag truncated

__device__ __forceinline__ u64 pack(u32 sum, u32 next) { return (static_cast<u64>(sum) << 32) | next; }

// Steps 1a + 1b for the chunk at stream offset `base`: T[j] = (output bytes, first tag start at or after the end of j's sub-chunk).
__device__ __forceinline__ void build_table(u64* T, u8* raw, const u8* __restrict__ src, const u32 n, const u64 base)
{
    // ---- 1a. the tag that would start at every position ----------------------------------------------------------
    // the chunk's bytes (+ 8: a tag's trailer may reach into the next chunk; zeros past the end of the stream) come in once, 16 per thread, and
    // every position reads its 8 from LDS (16 384 overlapping 8-byte gathers from global memory were a third of this function)
    static_assert(kThreads * 16 == kChunk, "one 16-byte piece per thread");
    {
        const u64 p = base + threadIdx.x * 16u;
        snp_u128_unaligned v = {{0, 0, 0, 0}};
        if (p + 16 <= n) v = *reinterpret_cast<const snp_u128_unaligned*>(src + p);
        else
            for (u32 i = 0; i < 16 && p + i < n; ++i) reinterpret_cast<u8*>(&v)[i] = src[p + i];
        *reinterpret_cast<snp_u128_unaligned*>(raw + threadIdx.x * 16u) = v;
        if (threadIdx.x < 8) raw[kChunk + threadIdx.x] = base + kChunk + threadIdx.x < n ? src[base + kChunk + threadIdx.x] : u8{0};
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < kChunk; j += kThreads) {
        const u64 pos = base + j;
        u64 e = pack(0, kFar);
        if (pos < n) {
            const u64 q = ld64u(raw + j);
            const u32 c = static_cast<u32>(q) & 0xffu;
            const u32 type = c & 3u;
            const u32 hi6 = c >> 2;
            const u32 extra = type == 0 ? (hi6 >= 60 ? hi6 - 59 : 0) : (type == 3 ? 4 : type);   // Constants.cs:42-76
            const u32 b1234 = static_cast<u32>(q >> 8);
            const u32 trailer = extra >= 4 ? b1234 : (b1234 & ((1u << (8 * extra)) - 1u));
            u64 len;                                                      // output bytes
            if (type == 0) len = hi6 >= 60 ? static_cast<u64>(trailer) + 1 : hi6 + 1;
            else if (type == 1) len = (hi6 & 7u) + 4;
            else len = hi6 + 1;
            const u64 next = static_cast<u64>(j) + 1 + extra + (type == 0 ? len : 0);   // relative to the chunk
            if (pos + 1 + extra <= n && next < kFar && len <= 0x7fffffffull)
                e = pack(static_cast<u32>(len), static_cast<u32>(next));
        }
        T[j] = e;
    }
    __syncthreads();
    // ---- 1b. pointer doubling inside each 4 KiB sub-chunk (in place: any value a reader sees is a valid jump) -----
    for (u32 r = 0; r < kRounds; ++r) {
#pragma unroll 4
        for (u32 j = threadIdx.x; j < kChunk; j += kThreads) {
            const u64 e = T[j];
            const u32 nx = static_cast<u32>(e);
            if (nx < kChunk && (nx / kSub) == (j / kSub) && base + nx < n) {   // the end of the stream is a terminal
                const u64 e2 = T[nx];
                const u64 sum = (e >> 32) + (e2 >> 32);
                T[j] = pack(sum > 0x7fffffffull ? 0x80000000u : static_cast<u32>(sum), static_cast<u32>(e2));
            }
        }
        __syncthreads();
    }
}

// Step 2's walk for one entry point: from (ip, op) through the chunk's sub-chunks; rec_ip / rec_op[sc] = the entry point of sub-chunk sc (sc >= 1)
// as the look-back records it, [kSubs] = the entry of the next chunk.
__device__ __forceinline__ void walk_chunk(const u64* T, const u32 n, const u64 base, u32 ip, u32 op, u32* rec_ip, u32* rec_op)
{
    for (u32 sc = 0; sc < kSubs; ++sc) {
        rec_ip[sc] = ip;
        rec_op[sc] = op;
        const u64 sub_end = base + static_cast<u64>(sc + 1) * kSub;
        if (ip == kBadIp || ip >= n || ip >= sub_end) continue;           // finished, irregular, or a literal jumps over this sub-chunk
        const u64 e = T[ip - base];
        const u32 nx = static_cast<u32>(e);
        const u64 sum = static_cast<u64>(op) + (e >> 32);
        if (nx >= kFar || base + nx > n || sum > 0x7fffffffull) { ip = kBadIp; continue; }
        ip = static_cast<u32>(base + nx);
        op = static_cast<u32>(sum);
    }
    rec_ip[kSubs] = ip;
    rec_op[kSubs] = op;
}

__global__ __launch_bounds__(kThreads) void k_tag_index(const u8* __restrict__ src, u32 n, u32 hb, u32 nchunks,
                                                  u64* __restrict__ entries, u32* __restrict__ ticket, const u32* __restrict__ wanted)
{
    __shared__ u64 T[kChunk];
    __shared__ __attribute__((aligned(16))) u8 s_raw[kChunk + 16];
    __shared__ u32 s_chunk;
    if (__hip_atomic_load(wanted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // (the fallback: k_tag_scan found every entry among the candidates)
    if (threadIdx.x == 0) s_chunk = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 k = s_chunk;
    if (k >= nchunks) return;
    const u64 base = hb + static_cast<u64>(k) * kChunk;               // stream offset of this chunk

    build_table(T, s_raw, src, n, base);
    // ---- 2. look-back: the true entry of this chunk, through its sub-chunks, to the entry of the next chunk -------
    if (threadIdx.x == 0) {
        u64 ent;
        if (k == 0) {
            ent = kValid | pack(0, hb);                                  // the first tag follows the varint preamble
        } else {
            // The hand-off IS this one 8-byte word (valid bit | output offset | stream position): a relaxed agent-scope load
            // (L2-served) and store are enough -- no payload behind a flag, so no acquire/release and no cache invalidation per
            // poll (MI355X_MICROARCH.md, hand-off price list: acquire polling costs 2-3x per hop).  Measured: 4.6 -> 0.84 us per hop
            // (k_tag_index over a 0.29 GiB stream: 86 -> 15.7 ms).
            while ((ent = __hip_atomic_load(&entries[static_cast<u64>(k) * kSubs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0)
                __builtin_amdgcn_s_sleep(1);
        }
        u32 ip = static_cast<u32>(ent);                                   // stream offset of a tag start (or n, or kBadIp)
        u32 op = static_cast<u32>(ent >> 32) & 0x7fffffffu;               // output bytes produced before it
        for (u32 sc = 0; sc < kSubs; ++sc) {
            if (sc) entries[static_cast<u64>(k) * kSubs + sc] = kValid | pack(op, ip);
            else if (k == 0) entries[0] = ent;
            const u64 sub_end = base + static_cast<u64>(sc + 1) * kSub;
            if (ip == kBadIp || ip >= n || ip >= sub_end) continue;       // finished, irregular, or a literal jumps over this sub-chunk
            const u64 e = T[ip - base];
            const u32 nx = static_cast<u32>(e);
            const u64 sum = static_cast<u64>(op) + (e >> 32);
            if (nx >= kFar || base + nx > n || sum > 0x7fffffffull) { ip = kBadIp; continue; }
            ip = static_cast<u32>(base + nx);
            op = static_cast<u32>(sum);
        }
        __hip_atomic_store(&entries[static_cast<u64>(k + 1) * kSubs], kValid | pack(op, ip), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- steps 2a / 2b: candidate entries per chunk, then one serial pass over the candidate tables ------------------------------------------
constexpr u32 kMaxCand = 8;                       // distinct candidate entries a chunk may have (more: the fallback)
constexpr u32 kProbe = 128;                       // walks from the chunk's first kProbe bytes define the next chunk's candidates
constexpr u32 kWide = 0xffu;

struct CandTable {                                // what step 2 would record for each candidate entry of one chunk; output bytes from the entry
    u32 ncand;                                    // kWide: too many distinct landings
    u32 key[kMaxCand];                            // stream position of the candidate entry
    u32 ip[kMaxCand][kSubs];                      // [c][sc - 1] for sc = 1..3: sub-chunk entry points; [c][kSubs - 1]: the next chunk's entry
    u32 op[kMaxCand][kSubs];
    u32 nxt[kMaxCand];                            // which candidate of the NEXT chunk that entry is (its row there), kDone, or kFail
};
constexpr u32 kDone = kMaxCand;                   // the stream ended (entry = (n, op) from here on)
constexpr u32 kFail = kMaxCand + 1;               // irregular from here on: the look-back kernel runs (and marks the table irregular)
constexpr u32 kByPos = kMaxCand + 2;              // the next entry is the POSITION ip[c][kSubs - 1]: a landing beyond the next chunk, or in it but not
                                                  // among its candidates -- the scan looks the position up when it gets there (k_tag_fix adds the row)
struct CandHandoff {                              // chunk k - 1 -> chunk k: the candidates, then the flag (release / acquire)
    u32 ready;
    u32 ncand;
    u32 key[kMaxCand];
};

__global__ __launch_bounds__(kThreads) void k_tag_cand(const u8* __restrict__ src, u32 n, u32 hb, u32 nchunks, CandTable* __restrict__ tables,
                                                 CandHandoff* __restrict__ hand, u32* __restrict__ ticket)
{
    __shared__ u64 T[kChunk];
    __shared__ __attribute__((aligned(16))) u8 s_raw[kChunk + 16];
    __shared__ u32 s_chunk;
    __shared__ u32 s_land[kProbe];
    __shared__ u32 s_ncand;
    __shared__ u32 s_key[kMaxCand];
    __shared__ u32 s_next_ncand;
    __shared__ u32 s_next_key[kMaxCand];
    if (threadIdx.x == 0) s_chunk = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 k = s_chunk;
    if (k >= nchunks) return;
    const u64 base = hb + static_cast<u64>(k) * kChunk;
    const u64 end = base + kChunk;
    build_table(T, s_raw, src, n, base);
    // where the walks that enter at the chunk's first bytes leave it: the next chunk's candidate entries
    if (threadIdx.x < kProbe) {
        u32 rip[kSubs + 1], rop[kSubs + 1];
        const u64 start = base + threadIdx.x;
        u32 land = kBadIp;
        if (start < n) {
            walk_chunk(T, n, base, static_cast<u32>(start), 0u, rip, rop);
            land = rip[kSubs];
        }
        s_land[threadIdx.x] = (land != kBadIp && land >= end && land < end + kChunk && land < n) ? land : kBadIp;   // (beyond the next chunk: it passes through)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        CandHandoff* const h = hand + k + 1;
        u32 cnt = 0;
        u32 keys[kMaxCand];
        for (u32 i = 0; i < kProbe && cnt != kWide; ++i) {
            const u32 v = s_land[i];
            if (v == kBadIp) continue;
            bool seen = false;
#pragma unroll
            for (u32 c = 0; c < kMaxCand; ++c) seen = seen || (c < cnt && keys[c] == v);
            if (seen) continue;
            if (cnt == kMaxCand) { cnt = kWide; break; }
#pragma unroll
            for (u32 c = 0; c < kMaxCand; ++c)
                if (c == cnt) keys[c] = v;
            ++cnt;
        }
#pragma unroll
        for (u32 c = 0; c < kMaxCand; ++c) {
            h->key[c] = keys[c];
            s_next_key[c] = keys[c];
        }
        s_next_ncand = cnt;
        h->ncand = cnt;
        __hip_atomic_store(&h->ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // this chunk's own candidates, from the chunk before (which took its ticket earlier and publishes before it waits: no chain)
        if (k == 0) {
            s_ncand = 1;
            s_key[0] = hb;
        } else {
            const CandHandoff* const m = hand + k;
            while (__hip_atomic_load(&m->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
            const u32 nc = m->ncand;
            s_ncand = nc;
            for (u32 c = 0; c < kMaxCand && nc != kWide && c < nc; ++c) s_key[c] = m->key[c];
        }
    }
    __syncthreads();
    const u32 nc = s_ncand;
    CandTable* const t = tables + k;
    if (threadIdx.x == 0) t->ncand = nc;
    if (nc != kWide && threadIdx.x < nc) {
        u32 rip[kSubs + 1], rop[kSubs + 1];
        const u32 key = s_key[threadIdx.x];
        walk_chunk(T, n, base, key, 0u, rip, rop);
        t->key[threadIdx.x] = key;
        for (u32 sc = 1; sc <= kSubs; ++sc) {
            t->ip[threadIdx.x][sc - 1] = rip[sc];
            t->op[threadIdx.x][sc - 1] = rop[sc];
        }
        const u32 out = rip[kSubs];
        u32 nx = out == kBadIp ? kFail : out == n ? kDone : kByPos;
        if (nx == kByPos && s_next_ncand != kWide)
            for (u32 c = 0; c < kMaxCand; ++c)
                if (c < s_next_ncand && s_next_key[c] == out) nx = c;
        t->nxt[threadIdx.x] = nx;
    }
}

// ---- step 2b: the scan --------------------------------------------------------------------------------------------------------------------
// The state between two chunks: a ROW of the next chunk's table, a POSITION further on (looked up when its chunk comes), the END of the
// stream, FAILED (irregular), or PENDING (a position that is no candidate: k_tag_fix adds its row, the scan runs again).
enum : u32 { kStRow = 0, kStPos = 1, kStEnd = 2, kStFail = 3, kStPend = 4 };
struct ScanState {
    u32 kind, v;                                  // v: row (kStRow) or stream position (kStPos)
    u64 op;                                       // output bytes before the entry
};
struct ScanCtl {                                  // in the workspace's control words
    u32 cand_ticket, look_back_ticket, fallback;  // (fallback: the look-back kernel is wanted)
    u32 complete;                                 // the scan reached the end of the stream: the entries are final
    u32 pending, pend_chunk, pend_ip, pend_op;    // a landing that needs a row (k_tag_fix)
    u32 cached;                                   // the run functions of the first pass are in the workspace (k_tag_fix only ADDS rows: they stay valid,
};                                                //  except where a run stopped at a pending landing)
struct RunCache {                                 // [thread][row]
    u8 kind[kThreads][kMaxCand];
    u32 v[kThreads][kMaxCand];
    u64 op[kThreads][kMaxCand];
};

// One chunk.  entries != nullptr: the replay -- writes the chunk's entries and files the pending request.
__device__ __forceinline__ void scan_step(ScanState& s, const u32 k, const CandTable* __restrict__ tables, const u32 n, const u32 hb,
                                          u64* __restrict__ entries, ScanCtl* __restrict__ ctl)
{
    if (s.kind == kStFail || s.kind == kStPend) return;
    if (s.op > 0x7fffffffull) { s.kind = kStFail; return; }
    u64* const e = entries ? entries + static_cast<u64>(k) * kSubs : nullptr;
    const u32 op = static_cast<u32>(s.op);
    if (s.kind == kStEnd) {
        if (e) for (u32 sc = 0; sc < kSubs; ++sc) e[sc] = kValid | pack(op, n);
        return;
    }
    const u64 end = hb + static_cast<u64>(k + 1) * kChunk;
    const CandTable& t = tables[k];
    if (s.kind == kStPos) {
        if (s.v >= end) {                                                // a literal jumps over this chunk
            if (e) for (u32 sc = 0; sc < kSubs; ++sc) e[sc] = kValid | pack(op, s.v);
            return;
        }
        u32 row = kMaxCand;
        if (t.ncand != kWide)
            for (u32 c = 0; c < kMaxCand; ++c)
                if (c < t.ncand && t.key[c] == s.v) row = c;
        if (row == kMaxCand) {
            if (ctl) { ctl->pend_chunk = k; ctl->pend_ip = s.v; ctl->pend_op = op; ctl->pending = 1; }
            s.kind = kStPend;
            return;
        }
        s.kind = kStRow;
        s.v = row;
    }
    if (t.ncand == kWide || s.v >= t.ncand) { s.kind = kStFail; return; }   // (cannot happen: a row comes from nxt or from the lookup above)
    const u32 row = s.v;
    if (e) e[0] = kValid | pack(op, t.key[row]);
    for (u32 sc = 1; sc < kSubs; ++sc) {
        const u64 sum = s.op + t.op[row][sc - 1];
        if (t.ip[row][sc - 1] == kBadIp || sum > 0x7fffffffull) { s.kind = kStFail; return; }
        if (e) e[sc] = kValid | pack(static_cast<u32>(sum), t.ip[row][sc - 1]);
    }
    s.op += t.op[row][kSubs - 1];
    const u32 nx = t.nxt[row];
    if (nx < kMaxCand) { s.kind = kStRow; s.v = nx; }
    else if (nx == kDone) s.kind = kStEnd;
    else if (nx == kByPos) { s.kind = kStPos; s.v = t.ip[row][kSubs - 1]; }
    else s.kind = kStFail;
}

// One workgroup.  A run of chunks is a function on <= 8 rows: every thread evaluates its run for each row; thread 0 chains the runs (a run entered
// by POSITION is evaluated then and there: chunks a literal jumps over cost no memory access); every thread replays its run from its true
// entry state, writing the entries.  `last`: no k_tag_fix follows -- a pending landing means the look-back kernel.
struct ScanLds {                                  // (carved out of the workgroup's LDS pool: the table of k_tag_fix lives there between scans)
    u8 r_kind[kThreads][kMaxCand];
    u32 r_v[kThreads][kMaxCand];
    u64 r_op[kThreads][kMaxCand];
    u8 in_kind[kThreads];
    u32 in_v[kThreads];
    u64 in_op[kThreads];
};
__device__ __forceinline__ void scan_pass(const CandTable* __restrict__ tables, u32 n, u32 hb, u32 nchunks, u64* __restrict__ entries,
                                          ScanCtl* __restrict__ ctl, RunCache* __restrict__ cache, const bool last, ScanLds& L, u32& s_final)
{
    auto& r_kind = L.r_kind;
    auto& r_v = L.r_v;
    auto& r_op = L.r_op;
    auto& in_kind = L.in_kind;
    auto& in_v = L.in_v;
    auto& in_op = L.in_op;
    const u32 per = (nchunks + kThreads - 1) / kThreads;
    const u32 k0 = min(threadIdx.x * per, nchunks), k1 = min(k0 + per, nchunks);
    const bool cached = __hip_atomic_load(&ctl->cached, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (u32 r = 0; r < kMaxCand; ++r) {
        ScanState s{kStRow, r, 0};
        if (cached && cache->kind[threadIdx.x][r] != kStPend) {          // (a later pass: only what stopped at a pending landing is evaluated again)
            s.kind = cache->kind[threadIdx.x][r];
            s.v = cache->v[threadIdx.x][r];
            s.op = cache->op[threadIdx.x][r];
        } else {
            if (k0 < k1 && (tables[k0].ncand == kWide || r >= tables[k0].ncand)) s.kind = kStFail;   // (no such row: never selected)
            for (u32 k = k0; k < k1 && s.kind != kStFail && s.kind != kStPend; ++k) scan_step(s, k, tables, n, hb, nullptr, nullptr);
            cache->kind[threadIdx.x][r] = static_cast<u8>(s.kind);
            cache->v[threadIdx.x][r] = s.v;
            cache->op[threadIdx.x][r] = s.op;
        }
        r_kind[threadIdx.x][r] = static_cast<u8>(s.kind);
        r_v[threadIdx.x][r] = s.v;
        r_op[threadIdx.x][r] = s.op;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ScanState s{kStRow, 0, 0};                                       // chunk 0 has one candidate: the first tag after the preamble
        for (u32 t = 0; t < kThreads; ++t) {
            in_kind[t] = static_cast<u8>(s.kind);
            in_v[t] = s.v;
            in_op[t] = s.op;
            const u32 a0 = min(t * per, nchunks), a1 = min(a0 + per, nchunks);
            if (s.kind == kStRow && a0 < a1) {
                const u32 r = s.v;
                s.kind = r_kind[t][r];
                s.v = r_v[t][r];
                s.op += r_op[t][r];
            } else if (s.kind == kStPos) {
                for (u32 k = a0; k < a1; ++k) scan_step(s, k, tables, n, hb, nullptr, nullptr);
            }
        }
        s_final = s.kind;
    }
    __syncthreads();
    ScanState s{in_kind[threadIdx.x], in_v[threadIdx.x], in_op[threadIdx.x]};
    for (u32 k = k0; k < k1; ++k) scan_step(s, k, tables, n, hb, entries, ctl);
    if (k0 < k1 && k1 == nchunks && s.kind == kStEnd) entries[static_cast<u64>(nchunks) * kSubs] = kValid | pack(static_cast<u32>(s.op), n);
    if (threadIdx.x == 0) {
        ctl->cached = 1;
        if (s_final == kStEnd) ctl->complete = 1;
        else if (s_final != kStPend || last) ctl->fallback = 1;          // irregular (or the stream does not end with its last chunk); or out of passes
    }
    __syncthreads();
}

// ---- step 2c: a landing that was no candidate gets its row ---------------------------------------------------------------------------------------
// One workgroup follows the true chain from the pending landing: a tag that leaves its chunk by itself (a literal longer than what is left of the
// chunk: the incompressible fragments) gets its row from its own bytes; anything else from the chunk's table, built here; until the chain lands on
// a candidate again (it has merged with the walks the candidates came from) or the budget of this pass is spent.
constexpr u32 kFixBudget = 4096;                  // landings one pass may give a row
constexpr u32 kFixPasses = 512;                   // passes at most (each ends where the chain rejoins the candidates: one per incompressible region)
constexpr u32 kChunksPerPass = 417;               // a pass costs ~0.35 ms, the look-back kernel 0.84 us per chunk: more passes than chunks / 417 and it is cheaper
struct FixLds {
    u64 T[kChunk];
    __attribute__((aligned(16))) u8 raw[kChunk + 16];
};
struct FixVars { u32 k, ip, go, far, next, len; };
__device__ __forceinline__ void fix_pass(const u8* __restrict__ src, u32 n, u32 hb, CandTable* __restrict__ tables, ScanCtl* __restrict__ ctl,
                                         FixLds& F, FixVars& V)
{
    u64* const T = F.T;
    u8* const s_raw = F.raw;
    u32 &s_k = V.k, &s_ip = V.ip, &s_go = V.go, &s_far = V.far, &s_next = V.next, &s_len = V.len;
    if (threadIdx.x == 0) { s_k = ctl->pend_chunk; s_ip = ctl->pend_ip; s_go = 1; }
    __syncthreads();
    for (u32 it = 0; it < kFixBudget; ++it) {
        if (!s_go) break;
        const u32 k = s_k, ip = s_ip;
        const u64 base = hb + static_cast<u64>(k) * kChunk, end = base + kChunk;
        // the tag at ip, from its own bytes: does it leave the chunk by itself?
        if (threadIdx.x == 0) {
            u64 q = 0;
            for (u32 i = 0; i < 8 && ip + i < n; ++i) q |= static_cast<u64>(src[ip + i]) << (8 * i);
            const u32 c = static_cast<u32>(q) & 0xffu, type = c & 3u, hi6 = c >> 2;
            const u32 extra = type == 0 ? (hi6 >= 60 ? hi6 - 59 : 0) : (type == 3 ? 4 : type);
            const u32 b1234 = static_cast<u32>(q >> 8);
            const u32 trailer = extra >= 4 ? b1234 : (b1234 & ((1u << (8 * extra)) - 1u));
            const u64 len = type == 0 ? (hi6 >= 60 ? static_cast<u64>(trailer) + 1 : hi6 + 1) : 0;
            const u64 next = static_cast<u64>(ip) + 1 + extra + len;
            s_far = type == 0 && static_cast<u64>(ip) + 1 + extra <= n && next >= end && next <= n && len <= 0x7fffffffull;
            s_next = static_cast<u32>(next);
            s_len = static_cast<u32>(len);
        }
        __syncthreads();
        u32 rip[kSubs + 1], rop[kSubs + 1];
        if (s_far) {                                                     // walk_chunk's record for a single tag that leaves the chunk
            for (u32 sc = 0; sc <= kSubs; ++sc) {
                const bool after = sc == kSubs || base + static_cast<u64>(sc) * kSub > ip;
                rip[sc] = after ? s_next : ip;
                rop[sc] = after ? s_len : 0u;
            }
        } else {
            build_table(T, s_raw, src, n, base);
            walk_chunk(T, n, base, ip, 0u, rip, rop);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            CandTable& t = tables[k];
            u32 r = t.ncand == kWide ? 0u : t.ncand;
            if (r >= kMaxCand) {
                ctl->fallback = 1;                                       // no room for another row
                s_go = 0;
            } else {
                t.key[r] = ip;
                for (u32 sc = 1; sc <= kSubs; ++sc) {
                    t.ip[r][sc - 1] = rip[sc];
                    t.op[r][sc - 1] = rop[sc];
                }
                const u32 out = rip[kSubs];
                u32 nx = out == kBadIp ? kFail : out == n ? kDone : kByPos;
                bool joined = nx != kByPos;                              // (the end, or irregular: nothing more to add)
                u32 ko = 0;
                if (nx == kByPos) {
                    ko = static_cast<u32>((out - hb) / kChunk);
                    const CandTable& o = tables[ko];
                    if (o.ncand != kWide)
                        for (u32 c = 0; c < kMaxCand; ++c)
                            if (c < o.ncand && o.key[c] == out) { joined = true; if (ko == k + 1) nx = c; }
                }
                t.nxt[r] = nx;
                t.ncand = r + 1;
                if (joined) s_go = 0;
                else { s_k = ko; s_ip = out; }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) ctl->pending = 0;
    __syncthreads();
}

// One workgroup: scan; while a landing is pending: give it (and what follows it, up to the rejoin) rows, scan again.
__global__ __launch_bounds__(kThreads) void k_tag_scan(const u8* __restrict__ src, CandTable* __restrict__ tables, u32 n, u32 hb, u32 nchunks,
                                                      u64* __restrict__ entries, ScanCtl* __restrict__ ctl, RunCache* __restrict__ cache)
{
    __shared__ __attribute__((aligned(16))) u8 pool[sizeof(FixLds) > sizeof(ScanLds) ? sizeof(FixLds) : sizeof(ScanLds)];
    __shared__ FixVars vars;
    __shared__ u32 s_final, s_more;
    if (__hip_atomic_load(&ctl->fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // (the caller asked for the look-back pass)
    for (u32 pass = 0;; ++pass) {
        scan_pass(tables, n, hb, nchunks, entries, ctl, cache, pass >= min(kFixPasses, max(1u, nchunks / kChunksPerPass)), *reinterpret_cast<ScanLds*>(pool), s_final);
        if (threadIdx.x == 0) s_more = ctl->complete == 0 && ctl->fallback == 0 && ctl->pending != 0;
        __syncthreads();
        if (!s_more) break;
        fix_pass(src, n, hb, tables, ctl, *reinterpret_cast<FixLds*>(pool), vars);
    }
}

// One thread per output fragment: the last table entry at or before the fragment's first output byte.
__global__ __launch_bounds__(256) void k_fragment_starts(const u64* __restrict__ scanned, const u64* __restrict__ looked_back,
                                                        const u32* __restrict__ fallback, u32 nent, u32 n, u32 expected,
                                                        u32 nfrag, u64* __restrict__ in_off, u32* __restrict__ in_len,
                                                        u64* __restrict__ out_off, u32* __restrict__ out_cap,
                                                        u32* __restrict__ skip)
{
    const u32 f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfrag) return;
    const u64* const entries = *fallback ? looked_back : scanned;
    const u32 target = f * SNP_BLOCK_SIZE;
    const u64 last = entries[nent - 1];
    const bool good = static_cast<u32>(last) == n && (static_cast<u32>(last >> 32) & 0x7fffffffu) == expected;
    out_off[f] = target;
    out_cap[f] = expected - target < SNP_BLOCK_SIZE ? expected - target : SNP_BLOCK_SIZE;
    if (!good) {                              // the fragment decoder then reports "incomplete" and the caller falls back
        in_off[f] = 0;
        in_len[f] = 0;
        skip[f] = 0;
        return;
    }
    u32 lo = 0, hi = nent - 1;                // entries are ordered by output offset; entries[0] has offset 0
    while (lo < hi) {
        const u32 mid = (lo + hi + 1) / 2;
        if ((static_cast<u32>(entries[mid] >> 32) & 0x7fffffffu) <= target) lo = mid;
        else hi = mid - 1;
    }
    const u64 e = entries[lo];
    const u32 ip = static_cast<u32>(e);
    in_off[f] = ip;
    in_len[f] = n - ip;
    skip[f] = target - (static_cast<u32>(e >> 32) & 0x7fffffffu);
}

}  // namespace

extern "C" u32 snp_tag_index_entries(u32 n, u32 hb) { return ((n - hb + kChunk - 1) / kChunk) * kSubs + 1; }

// Workspace layout (all zeroed by the launch): scanned entries | look-back entries | tickets and flag | hand-offs | candidate tables.
static size_t ws_entries(u32 nent) { return (static_cast<size_t>(nent) * 8 + 63) / 64 * 64; }
extern "C" size_t snp_tag_index_workspace_bytes(u32 n, u32 hb)
{
    const u32 nent = snp_tag_index_entries(n, hb);
    const u32 nchunks = (nent - 1) / kSubs;
    return 2 * ws_entries(nent) + 64 + (static_cast<size_t>(nchunks) + 2) * sizeof(CandHandoff) + static_cast<size_t>(nchunks) * sizeof(CandTable) + 128 + sizeof(RunCache);
}

extern "C" size_t snp_tag_index_fallback_offset(u32 n, u32 hb) { return 2 * ws_entries(snp_tag_index_entries(n, hb)) + 8; }   // (ctl[2])

// The workspace holds snp_tag_index_workspace_bytes(n, hb) bytes; its first snp_tag_index_entries words are the entry table when the fallback flag is 0
// (the debug dump in capi_host.hip reads those).  Three steps, so that a caller who uploads the stream in slices can index what has arrived:
//   snp_launch_tag_index_begin   zeroes the control words;
//   snp_launch_tag_index_chunks  k_tag_cand for chunks [first, first + count), IN ORDER (the tickets number the chunks across launches); chunk k
//                                needs stream bytes [hb + k * 16 KiB, hb + (k + 1) * 16 KiB + 8) on the device (snp_tag_index_chunks_ready);
//   snp_launch_tag_index_finish  the scan, the look-back fallback, the fragment table.
namespace {
struct TagIndexLayout {
    u32 nent, nchunks;
    u64 *scanned, *looked_back;
    u32* ctl;                                                            // [0] k_tag_cand's ticket, [1] k_tag_index's, [2] the fallback flag
    CandHandoff* hand;
    CandTable* tables;
    RunCache* cache;
};
TagIndexLayout tag_index_layout(u64* work, u32 n, u32 hb)
{
    TagIndexLayout L;
    L.nent = snp_tag_index_entries(n, hb);
    L.nchunks = (L.nent - 1) / kSubs;
    u8* const w = reinterpret_cast<u8*>(work);
    L.scanned = work;
    L.looked_back = reinterpret_cast<u64*>(w + ws_entries(L.nent));
    L.ctl = reinterpret_cast<u32*>(w + 2 * ws_entries(L.nent));
    L.hand = reinterpret_cast<CandHandoff*>(L.ctl + 16);
    L.tables = reinterpret_cast<CandTable*>(reinterpret_cast<u8*>(L.hand) + (static_cast<size_t>(L.nchunks) + 2) * sizeof(CandHandoff));
    L.cache = reinterpret_cast<RunCache*>((reinterpret_cast<uintptr_t>(L.tables + L.nchunks) + 63) / 64 * 64);
    return L;
}
}  // namespace

// How many chunks of the stream are complete on the device once its first `uploaded` bytes are (of n).
extern "C" u32 snp_tag_index_chunks_ready(u32 n, u32 hb, u64 uploaded)
{
    const u32 nchunks = (snp_tag_index_entries(n, hb) - 1) / kSubs;
    if (uploaded >= n) return nchunks;
    if (uploaded < static_cast<u64>(hb) + 8) return 0;
    const u64 k = (uploaded - hb - 8) / kChunk;
    return k < nchunks ? static_cast<u32>(k) : nchunks;
}

// look_back_only: skip the candidate pass (snp_launch_tag_index_chunks / the scan do nothing) -- for streams that are mostly literals longer than a
// chunk (hardly compressed data), where the candidate pass would fail anyway and only add its 4.7 ms per GiB.
extern "C" hipError_t snp_launch_tag_index_begin(u64* work, u32 n, u32 hb, int look_back_only, hipStream_t stream)
{
    const TagIndexLayout L = tag_index_layout(work, n, hb);
    // (the candidate tables are written before they are read: only what precedes them needs zeroing)
    hipError_t e = hipMemsetAsync(work, 0, reinterpret_cast<u8*>(L.tables) - reinterpret_cast<u8*>(work), stream);
    if (e == hipSuccess && look_back_only) e = hipMemsetAsync(L.ctl + 2, 1, 4, stream);   // (any non-zero value is "wanted")
    return e;
}

extern "C" hipError_t snp_launch_tag_index_chunks(const u8* src, u32 n, u32 hb, u64* work, u32 first, u32 count, hipStream_t stream)
{
    (void)first;                                                         // (in order: the ticket counter IS the chunk number)
    if (count == 0) return hipSuccess;
    const TagIndexLayout L = tag_index_layout(work, n, hb);
    hipLaunchKernelGGL(k_tag_cand, dim3(count), dim3(kThreads), 0, stream, src, n, hb, L.nchunks, L.tables, L.hand, L.ctl);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_tag_index_finish(const u8* src, u32 n, u32 hb, u32 expected, u64* work, u64* in_off, u32* in_len,
                                                  u64* out_off, u32* out_cap, u32* skip, hipStream_t stream)
{
    const TagIndexLayout L = tag_index_layout(work, n, hb);
    const u32 nfrag = (expected + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE;
    hipLaunchKernelGGL(k_tag_scan, dim3(1), dim3(kThreads), 0, stream, src, L.tables, n, hb, L.nchunks, L.scanned, reinterpret_cast<ScanCtl*>(L.ctl), L.cache);
    hipLaunchKernelGGL(k_tag_index, dim3(L.nchunks), dim3(kThreads), 0, stream, src, n, hb, L.nchunks, L.looked_back, L.ctl + 1, L.ctl + 2);
    hipLaunchKernelGGL(k_fragment_starts, dim3((nfrag + 255) / 256), dim3(256), 0, stream, L.scanned, L.looked_back, L.ctl + 2, L.nent, n, expected,
                       nfrag, in_off, in_len, out_off, out_cap, skip);
    return hipGetLastError();
}

extern "C" int snp_tag_index_look_back_only(u32 n, u32 expected) { return static_cast<u64>(n) * 100 >= static_cast<u64>(expected) * 85; }

extern "C" hipError_t snp_launch_tag_index(const u8* src, u32 n, u32 hb, u32 expected, u64* work, u64* in_off, u32* in_len,
                                           u64* out_off, u32* out_cap, u32* skip, hipStream_t stream)
{
    const int lbo = snp_tag_index_look_back_only(n, expected);
    hipError_t e = snp_launch_tag_index_begin(work, n, hb, lbo, stream);
    if (e != hipSuccess) return e;
    if (!lbo) e = snp_launch_tag_index_chunks(src, n, hb, work, 0, (snp_tag_index_entries(n, hb) - 1) / kSubs, stream);
    if (e != hipSuccess) return e;
    return snp_launch_tag_index_finish(src, n, hb, expected, work, in_off, in_len, out_off, out_cap, skip, stream);
}
