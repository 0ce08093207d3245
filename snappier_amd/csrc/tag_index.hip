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
// The entry table (one entry per 4 KiB of compressed data) is searched per fragment by k_fragment_starts; the
// fragment decoder (k_decompress<.., FRAG = true>) then parses at most 4 KiB of tags before its fragment begins.
// Anything that is not a well-formed stream ending exactly at (n, declared length) marks the table irregular and the
// caller falls back to the single-wavefront decoder, which owns the error semantics.
#include "snp_device.h"

namespace {

constexpr u32 kChunk = 16384;
constexpr u32 kSub = 4096;
constexpr u32 kSubs = kChunk / kSub;
constexpr u32 kRounds = 12;                       // a 4 KiB sub-chunk holds at most 2048 tags: 2^11 hops
constexpr u64 kValid = 1ull << 63;
constexpr u32 kBadIp = 0xffffffffu;               // entry: the stream is irregular from here on
constexpr u32 kFar = 0xfffffffeu;                 // table: next-pointer not representable / tag truncated

__device__ __forceinline__ u64 pack(u32 sum, u32 next) { return (static_cast<u64>(sum) << 32) | next; }

__global__ __launch_bounds__(256) void k_tag_index(const u8* __restrict__ src, u32 n, u32 hb, u32 nchunks,
                                                  u64* __restrict__ entries, u32* __restrict__ ticket)
{
    __shared__ u64 T[kChunk];
    __shared__ u32 s_chunk;
    if (threadIdx.x == 0) s_chunk = atomicAdd(ticket, 1u);
    __syncthreads();
    const u32 k = s_chunk;
    if (k >= nchunks) return;
    const u64 base = hb + static_cast<u64>(k) * kChunk;               // stream offset of this chunk

    // ---- 1a. the tag that would start at every position ----------------------------------------------------------
    for (u32 j = threadIdx.x; j < kChunk; j += 256) {
        const u64 pos = base + j;
        u64 e = pack(0, kFar);
        if (pos < n) {
            u64 q = 0;
            if (pos + 8 <= n) q = ld64u(src + pos);
            else
                for (u32 i = 0; pos + i < n; ++i) q |= static_cast<u64>(src[pos + i]) << (8 * i);
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
        for (u32 j = threadIdx.x; j < kChunk; j += 256) {
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

// One thread per output fragment: the last table entry at or before the fragment's first output byte.
__global__ __launch_bounds__(256) void k_fragment_starts(const u64* __restrict__ entries, u32 nent, u32 n, u32 expected,
                                                        u32 nfrag, u64* __restrict__ in_off, u32* __restrict__ in_len,
                                                        u64* __restrict__ out_off, u32* __restrict__ out_cap,
                                                        u32* __restrict__ skip)
{
    const u32 f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfrag) return;
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

// entries: snp_tag_index_entries(n, hb) u64 words followed by one u32 ticket; zeroed here.
extern "C" hipError_t snp_launch_tag_index(const u8* src, u32 n, u32 hb, u32 expected, u64* entries, u64* in_off, u32* in_len,
                                           u64* out_off, u32* out_cap, u32* skip, hipStream_t stream)
{
    const u32 nent = snp_tag_index_entries(n, hb);
    const u32 nchunks = (nent - 1) / kSubs;
    const u32 nfrag = (expected + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE;
    hipError_t e = hipMemsetAsync(entries, 0, static_cast<size_t>(nent) * 8 + 8, stream);
    if (e != hipSuccess) return e;
    u32* ticket = reinterpret_cast<u32*>(entries + nent);
    hipLaunchKernelGGL(k_tag_index, dim3(nchunks), dim3(256), 0, stream, src, n, hb, nchunks, entries, ticket);
    hipLaunchKernelGGL(k_fragment_starts, dim3((nfrag + 255) / 256), dim3(256), 0, stream, entries, nent, n, expected, nfrag,
                       in_off, in_len, out_off, out_cap, skip);
    return hipGetLastError();
}
