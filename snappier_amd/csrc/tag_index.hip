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
//   2b. k_tag_scan: ONE workgroup walks the candidate tables in stream order (64 chunks' tables staged in LDS at a time): the true entry of
//       chunk k selects a row, the row gives the entry of chunk k + 1.  An entry that is no candidate (a long literal that lands in the middle
//       of a chunk and has not merged, more than 8 distinct landings) raises a flag, and the look-back kernel of step 2 runs after all.
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
constexpr u32 kFail = kMaxCand + 1;               // not a candidate of the next chunk, irregular, or it jumps over the next chunk: the look-back kernel runs
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
        u32 nx = kFail;
        if (out == n) nx = kDone;
        else if (out != kBadIp && s_next_ncand != kWide)
            for (u32 c = 0; c < kMaxCand; ++c)
                if (c < s_next_ncand && s_next_key[c] == out) nx = c;
        t->nxt[threadIdx.x] = nx;
    }
}

// One workgroup: the true entries out of the candidate tables.  A chunk is a function  row -> (row of the next chunk, output bytes)  on at most
// 8 rows (+ two absorbing states); functions compose, so: every thread composes its run of chunks for all 8 rows, thread 0 chains the kThreads
// run functions, every thread replays its run from its true row and writes the entries.
__global__ __launch_bounds__(kThreads) void k_tag_scan(const CandTable* __restrict__ tables, u32 n, u32 hb, u32 nchunks, u64* __restrict__ entries,
                                                      u32* __restrict__ fallback)
{
    __shared__ u8 s_nxt[kThreads][kMaxCand];
    __shared__ u64 s_sum[kThreads][kMaxCand];
    __shared__ u32 s_row[kThreads];
    __shared__ u64 s_op[kThreads];
    __shared__ u32 s_fail;
    if (__hip_atomic_load(fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // (the caller asked for the look-back pass)
    const u32 per = (nchunks + kThreads - 1) / kThreads;
    const u32 k0 = min(threadIdx.x * per, nchunks), k1 = min(k0 + per, nchunks);
    if (threadIdx.x == 0) s_fail = 0;
    for (u32 r = 0; r < kMaxCand; ++r) {
        u32 row = r;
        u64 sum = 0;
        for (u32 k = k0; k < k1 && row < kMaxCand; ++k) {
            const CandTable& t = tables[k];
            if (t.ncand == kWide || row >= t.ncand) { row = kFail; break; }
            sum += t.op[row][kSubs - 1];
            row = t.nxt[row];
        }
        s_nxt[threadIdx.x][r] = static_cast<u8>(row);
        s_sum[threadIdx.x][r] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 row = 0;                                                     // chunk 0 has one candidate: the first tag after the preamble
        u64 op = 0;
        for (u32 t = 0; t < kThreads; ++t) {
            s_row[t] = row;
            s_op[t] = op;
            if (row < kMaxCand) {
                op += s_sum[t][row];
                row = s_nxt[t][row];
            }
        }
    }
    __syncthreads();
    u32 row = s_row[threadIdx.x];
    u64 op = s_op[threadIdx.x];
    bool fail = false;
    for (u32 k = k0; k < k1; ++k) {
        u64* const e = entries + static_cast<u64>(k) * kSubs;
        if (row == kFail || op > 0x7fffffffull) { fail = true; break; }
        if (row == kDone) {
            for (u32 sc = 0; sc < kSubs; ++sc) e[sc] = kValid | pack(static_cast<u32>(op), n);
            continue;
        }
        const CandTable& t = tables[k];
        if (t.ncand == kWide || row >= t.ncand) { fail = true; break; }
        e[0] = kValid | pack(static_cast<u32>(op), t.key[row]);
        bool bad = false;
        for (u32 sc = 1; sc < kSubs; ++sc) {
            bad = bad || t.ip[row][sc - 1] == kBadIp || op + t.op[row][sc - 1] > 0x7fffffffull;
            e[sc] = kValid | pack(static_cast<u32>(op + t.op[row][sc - 1]), t.ip[row][sc - 1]);
        }
        if (bad) { fail = true; break; }
        op += t.op[row][kSubs - 1];
        row = t.nxt[row];
    }
    if (k1 == nchunks && k0 < k1 && !fail) {                            // (the thread that owns the last chunk: the entry after it)
        if (row == kDone) entries[static_cast<u64>(nchunks) * kSubs] = kValid | pack(static_cast<u32>(op), n);
        else fail = true;                                                // the stream does not end with its last chunk: irregular
    }
    if (fail) atomicOr(&s_fail, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && s_fail) *fallback = 1;
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
    return 2 * ws_entries(nent) + 64 + (static_cast<size_t>(nchunks) + 2) * sizeof(CandHandoff) + static_cast<size_t>(nchunks) * sizeof(CandTable) + 64;
}

extern "C" size_t snp_tag_index_fallback_offset(u32 n, u32 hb) { return 2 * ws_entries(snp_tag_index_entries(n, hb)) + 8; }   // (ctl[2])

// The workspace holds snp_tag_index_workspace_bytes(n, hb) bytes; its first snp_tag_index_entries words are the entry table when the fallback flag is 0
// (the debug dump in capi.hip reads those).  Three steps, so that a caller who uploads the stream in slices can index what has arrived:
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
    hipLaunchKernelGGL(k_tag_scan, dim3(1), dim3(kThreads), 0, stream, L.tables, n, hb, L.nchunks, L.scanned, L.ctl + 2);
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
