// compress_lanes.hip -- Snappy fragment compression, one <= 64 KiB fragment per LANE (gfx950), bit-exact with
// SnappyCompressor.CompressFragment (Snappier/Internal/SnappyCompressor.cs:174-415) for both TableEntry hashes.
//
// Why a second layout.  The reference parse is a serial chain: probe -> table lookup -> candidate compare -> insert,
// ~14 000 table accesses per html-like fragment.  The wave-per-fragment kernel (compress_win.hip) keeps the table in LDS,
// which caps a CU at 4-5 fragments in flight (32 KiB each of 160 KiB of LDS), and a lone wavefront is bound by
// instruction issue (one instruction per ~6 cycles).  For LARGE batches the parallelism that matters is across
// fragments, so here every lane runs the serial parse of its own fragment: 64 fragments per wavefront, 163 840
// independent memory streams in flight.  The hash table of each fragment lives in an HBM workspace (u32 entries: position +
// 16 check bits; stride = CalculateTableSize of the batch's longest fragment, 64 KiB at 64 KiB; zeroed by the owning
// wavefront at kernel start -- HashTable.cs:52,57-71);
// LDS holds only the 4 x 256-entry table for the SNP_HASH_CRC32C hash (the CRC step is GF(2)-linear, so it factors
// over the four input bytes; gfx950 has no CRC instruction).  The kernel is bound by the rate at which HBM serves
// random 4-byte read-modify-writes to the tables (scripts/microbench_random_table.hip), not by
// instruction issue.  Batches below 16 384 fragments use compress_win.hip (one wavefront per fragment is better there).
#include <cstdlib>

#include "snp_device.h"

#ifndef SNP_CL_FLAT
#define SNP_CL_FLAT 1     // 1: flat per-lane state machine (default); 0: the reference's nested loops, verbatim
#endif
#ifndef SNP_CL_SLOTS
#define SNP_CL_SLOTS 1    // probes of one lane's scan issued together (same-process A/B, scripts/ab_compress_opts.py: 1 is 1.5 % faster than 2; 3 and 4 cost bandwidth)
#endif

namespace {

constexpr u32 kDefaultSlots = SNP_CL_SLOTS;

constexpr u32 crc_step32(u32 x)
{
    for (int k = 0; k < 32; ++k) x = (x >> 1) ^ ((x & 1u) ? 0x82F63B78u : 0u);
    return x;
}

// Table entry = position (low 16 bits, what the reference stores) | 16 check bits of the 4 bytes at that position.
// A probe whose check bits differ from the entry's cannot match, so the candidate's bytes (a random 64-byte-sector
// read) are fetched only when the check bits agree; position 0 (the zero-initialised state) is compared against the
// fragment's first four bytes kept in a register.  Results are unchanged: "check bits differ" implies "bytes differ".
__device__ __forceinline__ u32 check_bits(u32 bytes) { return (bytes * 0x9E3779B1u) & 0xffff0000u; }

// Probe + insert as ONE memory request.  `cand = table[h]; table[h] = ip` (SnappyCompressor.cs:329-333) IS an exchange: one
// global_atomic_swap (executed at L2) instead of a load and a store, the store being a read-for-ownership of the same sector.
// Round 3, same-process A/B on one workspace: 123.0 -> 112.8 ms per 10 GiB of html-like blocks, mixed corpus 174.0 -> 161.1
// (profiles/r03a_compress_option_ab.json, r03b_compress_option_ab_*.json).  Measured with it and rejected: non-temporal table loads /
// stores (148.3 ms: the two halves of a read-modify-write want the sector to stay in L2 in between) and a 16-byte register window
// over the input (one input load per ~9 probes: 121.2 ms alone, 112.4 with the exchange -- input requests hit L1 / L2 and are nearly
// free at the table-rate bound); both were removed again.
__device__ __forceinline__ u32 table_swap(u32* p, u32 v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct LaneCtx {
    const u8* src;
    u8* dst;
    u32* table;
    u32 first4;     // ld32(src + 0)
    u32 n;
    u32 mask;
    u32 hmask;      // Lmap(mask): H_crc(b) = Lmap(b) ^ Lmap(mask)
};

template <int VARIANT>
__device__ __forceinline__ u32 lane_hash(const LaneCtx& c, u32 bytes, const u16 (*lut)[256])
{
    u32 hash;
    if constexpr (VARIANT == SNP_HASH_CRC32C) {
        // Sse42.Crc32(bytes, mask) = Lmap(bytes ^ mask), Lmap linear: XOR of one table entry per input byte
        hash = lut[0][bytes & 0xffu] ^ lut[1][(bytes >> 8) & 0xffu] ^ lut[2][(bytes >> 16) & 0xffu] ^ lut[3][bytes >> 24] ^ c.hmask;
    } else {
        hash = (0x1e35a7bdu * bytes) >> 17;                            // HashTable.cs:121-122
    }
    return (hash & c.mask) >> 1;                                       // :125 (byte offset -> entry index)
}

// EmitLiteral  SnappyCompressor.cs:418-464
__device__ __forceinline__ u32 lane_emit_literal(const LaneCtx& c, u32 op, u32 s, u32 len)
{
    u8* o = c.dst + op;
    const u32 k = len - 1;
    u32 hdr;
    if (k < 60) { o[0] = static_cast<u8>(k << 2); hdr = 1; }
    else if (k < 256) { o[0] = static_cast<u8>(60u << 2); o[1] = static_cast<u8>(k); hdr = 2; }
    else { o[0] = static_cast<u8>(61u << 2); o[1] = static_cast<u8>(k); o[2] = static_cast<u8>(k >> 8); hdr = 3; }   // k < 65536
    const u8* src = c.src + s;
    u8* d = o + hdr;
    u32 i = 0;
    for (; i + 16 <= len; i += 16)
        *reinterpret_cast<snp_u128_unaligned*>(d + i) = *reinterpret_cast<const snp_u128_unaligned*>(src + i);
    for (; i + 4 <= len; i += 4) st32u(d + i, ld32u(src + i));
    for (; i < len; ++i) d[i] = src[i];
    return op + hdr + len;
}

// EmitLiteral (:418-464) with the first 16 bytes of the literal already in registers (`first`): a literal of <= 16 bytes
// costs no load at all.  `cap` > 0: the lane's output area holds MaxCompressedLength bytes, so a 16-byte store that
// overshoots the literal is harmless (the next tag overwrites the excess) whenever it still ends inside the area.
__device__ __forceinline__ u32 lane_emit_literal16(const LaneCtx& c, u32 op, u32 s, u32 len, const snp_u128_unaligned& first, u32 cap, bool merge)
{
    u8* o = c.dst + op;
    const u32 k = len - 1;
    if (merge && len <= 15 && op + 16 <= cap) {                        // tag + body in ONE 16-byte store (the common case)
        snp_u128_unaligned w;
        w.v[0] = (first.v[0] << 8) | (k << 2);
        w.v[1] = __builtin_amdgcn_alignbit(first.v[1], first.v[0], 24);
        w.v[2] = __builtin_amdgcn_alignbit(first.v[2], first.v[1], 24);
        w.v[3] = __builtin_amdgcn_alignbit(first.v[3], first.v[2], 24);
        *reinterpret_cast<snp_u128_unaligned*>(o) = w;
        return op + 1 + len;
    }
    u32 hdr;
    if (k < 60) { o[0] = static_cast<u8>(k << 2); hdr = 1; }
    else if (k < 256) { o[0] = static_cast<u8>(60u << 2); o[1] = static_cast<u8>(k); hdr = 2; }
    else { o[0] = static_cast<u8>(61u << 2); o[1] = static_cast<u8>(k); o[2] = static_cast<u8>(k >> 8); hdr = 3; }   // k < 65536
    u8* d = o + hdr;
    if (len >= 16 || op + hdr + 16 <= cap) {
        *reinterpret_cast<snp_u128_unaligned*>(d) = first;
    } else {                                                           // exact-length stores cut out of the registers
        u32 w[4] = {first.v[0], first.v[1], first.v[2], first.v[3]};
        u32 at = 0, wi = 0;
        if (len & 8) { st32u(d, w[0]); st32u(d + 4, w[1]); at = 8; wi = 2; }
        if (len & 4) { st32u(d + at, w[wi]); at += 4; ++wi; }
        u32 rest = w[wi & 3];
        if (len & 2) { d[at] = static_cast<u8>(rest); d[at + 1] = static_cast<u8>(rest >> 8); rest >>= 16; at += 2; }
        if (len & 1) d[at] = static_cast<u8>(rest);
    }
    if (len > 16) {
        const u8* src = c.src + s;
        u32 i = 16;
        for (; i + 16 <= len; i += 16)
            *reinterpret_cast<snp_u128_unaligned*>(d + i) = *reinterpret_cast<const snp_u128_unaligned*>(src + i);
        if (i < len)                                                   // the tail, as one 16-byte piece ending at len
            *reinterpret_cast<snp_u128_unaligned*>(d + len - 16) = *reinterpret_cast<const snp_u128_unaligned*>(src + len - 16);
    }
    return op + hdr + len;
}

// EmitCopyAtMost64*  SnappyCompressor.cs:467-505
// `cap` > 0: the tag is written as ONE 4-byte store (the reference does the same, :467-505 "writes 4 B blind"); the byte
// or two beyond the tag are overwritten by whatever is emitted next and still lie inside MaxCompressedLength.
__device__ __forceinline__ u32 lane_emit_copy64(u8* dst, u32 op, u32 off, u32 len, u32 cap)
{
    u8* o = dst + op;
    const bool one = len < 12 && off < 2048;
    const u32 w = one ? (1u | ((len - 4) << 2) | ((off >> 8) << 5) | ((off & 0xffu) << 8)) : (2u | ((len - 1) << 2) | (off << 8));
    if (op + 4 <= cap) {
        st32u(o, w);
    } else {
        o[0] = static_cast<u8>(w);
        o[1] = static_cast<u8>(w >> 8);
        if (!one) o[2] = static_cast<u8>(w >> 16);
    }
    return op + (one ? 2u : 3u);
}

// EmitCopyLenLessThan12 / EmitCopyLenGreaterThanOrEqualTo12  SnappyCompressor.cs:507-543
__device__ __forceinline__ u32 lane_emit_copy(u8* dst, u32 op, u32 off, u32 len, u32 cap = 0)
{
    while (len >= 68) { op = lane_emit_copy64(dst, op, off, 64, cap); len -= 64; }
    if (len > 64) { op = lane_emit_copy64(dst, op, off, 60, cap); len -= 60; }
    return lane_emit_copy64(dst, op, off, len, cap);
}

// FindMatchLength  SnappyCompressor.cs:562-688: bytes s1[k] == s2[k] for k < result, s2 + result <= n
__device__ __forceinline__ u32 lane_find_match_length(const u8* src, u32 s1, u32 s2, u32 n)
{
    u32 matched = 0;
    while (s2 + matched + 8 <= n) {
        const u64 x = ld64u(src + s1 + matched) ^ ld64u(src + s2 + matched);
        if (x) return matched + (static_cast<u32>(__builtin_ctzll(x)) >> 3);
        matched += 8;
    }
    while (s2 + matched < n && src[s1 + matched] == src[s2 + matched]) ++matched;
    return matched;
}

// ---- output staging (option bit 4) -------------------------------------------------------------------------------
// A lane emits its compressed stream in pieces of 2..17 bytes (a copy tag, a short literal): thousands of partial-sector
// stores per fragment, each one a write request to L2.  Staged, the pieces go into a per-lane LDS buffer (unaligned LDS
// stores) and leave as whole 64-byte runs: four 16-byte global stores per 64 bytes of output.  The buffer holds the
// bytes [flushed, op) of the lane's output at offset 0.. ; it is drained before anything is written around it.
constexpr u32 kStageStride = 100;      // bytes of LDS per lane: 96 usable (63 pending + 16 + 4 blind), odd dword stride

struct OutStage {
    u8* lds;        // this lane's buffer
    u32 flushed;    // output bytes [0, flushed) are in global memory
};

__device__ __forceinline__ void stage_flush64(const LaneCtx& c, OutStage& st)
{
    u8* g = c.dst + st.flushed;
    // (non-temporal: nobody reads the compressed bytes back in this kernel, and L2 has better uses -- interleaved launches 103.23 -> 102.78 ms)
    typedef u32 v4u __attribute__((ext_vector_type(4), aligned(1)));
#pragma unroll
    for (u32 i = 0; i < 64; i += 16) {
        const snp_u128_unaligned q = *reinterpret_cast<const snp_u128_unaligned*>(st.lds + i);
        const v4u v = {q.v[0], q.v[1], q.v[2], q.v[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(g + i));
    }
    const snp_u128_unaligned t0 = *reinterpret_cast<const snp_u128_unaligned*>(st.lds + 64);
    const snp_u128_unaligned t1 = *reinterpret_cast<const snp_u128_unaligned*>(st.lds + 80);
    *reinterpret_cast<snp_u128_unaligned*>(st.lds) = t0;
    *reinterpret_cast<snp_u128_unaligned*>(st.lds + 16) = t1;
    st.flushed += 64;
}

// Write out whatever is staged (exact bytes), e.g. before a long literal is copied directly.
__device__ __forceinline__ void stage_drain(const LaneCtx& c, OutStage& st, u32 op)
{
    const u32 fill = op - st.flushed;
    u8* g = c.dst + st.flushed;
    u32 i = 0;
    for (; i + 16 <= fill; i += 16)
        *reinterpret_cast<snp_u128_unaligned*>(g + i) = *reinterpret_cast<const snp_u128_unaligned*>(st.lds + i);
    for (; i < fill; ++i) g[i] = st.lds[i];
    st.flushed = op;
}

// SMALL = true: the launch for batches whose LONGEST fragment is at most `small_max` bytes (256-byte blocks: a handful of probes per
// fragment).  Such a launch is bound by the texture path: every step of a lane is ~7 scattered vector-memory instructions at 40-57 cycles
// each (PMC, profiles/r03y_small_compress_pmc.txt: TA busy 80 %).  With the fragment's BYTES copied into LDS first (small_max + 16 per lane,
// dynamic), the input, candidate and literal reads of a step are LDS reads and only the table accesses and the output stay on the
// texture path.  Whether a batch qualifies is known only on the device (max_len), so the host launches this kernel in front of the
// general one whenever the previous batch was small, and each of the two returns at once if the batch is not its own.
// Launch options of k_compress_lanes (`opts`, chosen per launch from the batch size by snp_launch_compress_lanes; results never depend on them):
enum : int {
    kOptBlindLiterals = 1,    // short literals leave as one 16-byte store (inside MaxCompressedLength)
    kOptBlindLiteral16 = 2,   // ... also the 16-byte form of lane_emit_literal16
    kOptBlindCopies = 4,      // copy tags as one 4-byte store
    kOptShortExtend = 8,      // match extension: 16 + 16 bytes per trip only
    kOptStagedOutput = 16,    // output staged per lane in LDS, leaving in 64-byte non-temporal runs (launches of >= 32 768 fragments)
    kOptExchangeProbe = 64,   // probe + insert as ONE atomic exchange (one probe per trip only)
    kOptInputWindow = 128,    // the probe bytes out of a 16-byte register window over the input (launches of >= 131 072 fragments)
};

template <int VARIANT, u32 kSlots, bool SMALL>
__global__ __launch_bounds__(SNP_WAVE) void k_compress_lanes(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                            const u32* __restrict__ in_len, u32 nblocks,
                                                            u8* __restrict__ out, const u64* __restrict__ out_off,
                                                            u32* __restrict__ out_len, i32* __restrict__ status,
                                                            int emit_varint, const snp_table_pieces tp, int opts,
                                                            const u32* __restrict__ max_len, u32 small_max)
{
    __shared__ u16 lut[4][256];
    if (VARIANT == SNP_HASH_CRC32C) {
        for (u32 e = threadIdx.x; e < 1024; e += blockDim.x)
            lut[e >> 8][e & 255u] = static_cast<u16>(crc_step32((e & 255u) << (8 * (e >> 8))) & 0x7ffeu);
        __syncthreads();
    }
    __shared__ u8 s_out[SMALL ? 16 : SNP_WAVE * kStageStride];        // (SMALL: the stage lives behind the input slots in dynamic LDS, sized by the launch's lanes per wavefront)
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    // Table stride = CalculateTableSize of the LONGEST fragment of the batch (HashTable.cs:57-71; k_max_len ran before this
    // launch): a batch of 256-byte blocks keeps 1 KiB of table per fragment, not 64 KiB.  The wavefront's tables are one
    // contiguous run, zeroed here with coalesced 16-byte stores (HashTable.cs:52) instead of a memset of the worst case.
    const u32 maxlen = *max_len;
    if (SMALL ? maxlen > small_max : (small_max != 0 && maxlen <= small_max)) return;   // the twin launch's batch
    extern __shared__ __attribute__((aligned(16))) u8 s_dyn[];         // SMALL: blockDim x (small_max + 16) bytes of input
    const u32 tstride = maxlen > 16384 ? 16384u : maxlen < 256 ? 256u : (2u << (31u - __clz(maxlen - 1)));
    u32* wt;                                                            // this workgroup's run of tables (wave-uniform)
    {
        const u32 first = blockIdx.x * blockDim.x;
        const u32 mine = nblocks - first < blockDim.x ? nblocks - first : blockDim.x;
        const u32 piece = first / tp.piece_frags;                       // (piece_frags is a multiple of 64: no run straddles two pieces)
        wt = tp.p[piece] + static_cast<size_t>(first - piece * tp.piece_frags) * tstride;
        const size_t words = static_cast<size_t>(mine) * tstride;       // a multiple of 256
        if (tstride == 16384u) {
            // full-size tables: non-temporal -- 4 MiB of zeros per wavefront that the probes will fetch sector by sector much later need not pass
            // through L2 (ten interleaved launch pairs each 0.2-0.9 ms faster, mean 102.67 -> 102.20 ms).  Small tables are probed at once and
            // want their zeros in L2: 256-byte blocks 60.5 -> 58.6 GB/s with non-temporal stores.
            typedef u32 v4u __attribute__((ext_vector_type(4)));
            const v4u zero4 = {0, 0, 0, 0};
            for (size_t i = static_cast<size_t>(threadIdx.x) * 4; i < words; i += static_cast<size_t>(blockDim.x) * 4)
                __builtin_nontemporal_store(zero4, reinterpret_cast<v4u*>(wt + i));
        } else {
            for (size_t i = static_cast<size_t>(threadIdx.x) * 4; i < words; i += static_cast<size_t>(blockDim.x) * 4)
                *reinterpret_cast<uint4*>(wt + i) = make_uint4(0, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // other lanes' stores, before this lane probes its table
    }
    if (b >= nblocks) return;
    const bool staged = (opts & kOptStagedOutput) != 0;
    const bool t_swap = kSlots == 1 && (opts & kOptExchangeProbe) != 0;          // probe + insert as one atomic exchange (one probe per trip only)
    bool in_win = (opts & kOptInputWindow) != 0;                              // 16-byte register window over the input for the probe bytes (n >= 32, set below)
    OutStage stg{(SMALL ? s_dyn + blockDim.x * (small_max + 16u) : s_out) + threadIdx.x * kStageStride, 0};

    LaneCtx c;
    c.dst = out + out_off[b];
    c.n = in_len[b];
    c.table = wt + static_cast<size_t>(threadIdx.x) * tstride;
    const u32 n = c.n;
    if constexpr (SMALL) {                                              // the fragment's bytes into this lane's LDS slot (n <= small_max)
        const u8* g = in + in_off[b];
        u8* mine = s_dyn + threadIdx.x * (small_max + 16u);
        u32 i = 0;
        for (; i + 16 <= n; i += 16) {
            const snp_u128_unaligned q = *reinterpret_cast<const snp_u128_unaligned*>(g + i);
            *reinterpret_cast<uint4*>(mine + i) = make_uint4(q.v[0], q.v[1], q.v[2], q.v[3]);
        }
        for (; i + 4 <= n; i += 4) *reinterpret_cast<u32*>(mine + i) = ld32u(g + i);
        for (; i < n; ++i) mine[i] = g[i];
        c.src = mine;
    } else {
        c.src = in + in_off[b];
    }
    c.first4 = n >= 4 ? ld32u(c.src) : 0u;
    if (n > SNP_BLOCK_SIZE) { out_len[b] = 0; status[b] = SNP_ERR_BAD_ARG; return; }

    u32 op = 0;
    if (emit_varint) {                                                 // VarIntEncoding.TryWrite  VarIntEncoding.Write.cs:5-79
        u32 v = n;
        while (v >= 128) { c.dst[op++] = static_cast<u8>(v | 0x80u); v >>= 7; }
        c.dst[op++] = static_cast<u8>(v);
    }
    stg.flushed = op;                                                   // the preamble went straight to global memory

    u32 ip = 0;
#if SNP_CL_FLAT
    // Flat per-lane state machine: in every trip of the loop each lane does ONE step of its own parse -- a probe
    // (scan or the probe right after a copy) or a bounded piece of match extension -- so a lane never idles while its
    // neighbours finish a longer scan.  (With the reference's nested loops a wavefront pays the LONGEST scan of its
    // 64 fragments on every outer iteration.)
    enum : u32 { kScan = 0, kPost = 1, kExtend = 2, kDone = 3 };
    u32 mode = kDone;
    u32 next_emit = 0, skip = 32, cand = 0, base = 0, mlen = 0, limit = 0;
    snp_u128_unaligned win = {};                                       // option bit 7: input bytes [win_at, win_at + 16)
    u32 win_at = 0x80000000u;
    in_win = in_win && n >= 32;
    if (n >= 15) {                                                     // :190
        const u32 tsize = n > 16384 ? 16384u : n < 256 ? 256u : (2u << (31u - __clz(n - 1)));   // HashTable.cs:57-71
        c.mask = 2 * (tsize - 1);                                      // :181
        c.hmask = static_cast<u16>(crc_step32(c.mask) & 0x7ffeu);
        limit = n - 15;                                                // :192
        mode = kScan;                                                  // first outer iteration: next_emit = 0, ip = 1  :198-199
        ip = 1;
    }
    // Every trip of the loop is three dependent memory round trips, whatever the lanes are doing:
    //   trip 1: the input bytes of this lane's probes (one 8-byte load covers ip-1 / ip / ip+1..4)  |  the next 32 + 32
    //           bytes of a running match extension;
    //   trip 2: the table entries of the probes;
    //   trip 3: for the first probe whose entry can match (check bits agree): 16 bytes at the candidate, 16 bytes at
    //           the probe and the first 16 bytes of the pending literal -- so a hit is confirmed, its literal emitted
    //           from registers and a match shorter than 16 finished without another trip.
    // (The first version waited for each of these in turn plus a load per literal piece and four extension steps:
    // ~10 trips per loop trip, and the loop trip is what 64 lanes pay together.)
    const u32 lit_cap = (opts & kOptBlindLiterals) ? 32u + n + n / 6u : 0u;            // blind 16-byte literal stores stay inside MaxCompressedLength
    while (__any(mode != kDone)) {
        u32 cp_len = 0, cp_off = 0;                                     // the copy this trip ends with, emitted once below
        u32 lit_len = 0;                                                // the literal this trip ends with (staged mode: emitted below)
        snp_u128_unaligned lit16 = {};
        const bool post = mode == kPost;
        const bool scanning = mode == kScan || post;
        u32 p[kSlots], nx[kSlots], sk[kSlots], d[kSlots], h[kSlots], cv[kSlots];
        bool legal[kSlots];
        u64 w0 = 0;
        if (scanning) {
            u32 q = ip, sv = skip;
            bool ok = true;
#pragma unroll
            for (u32 k = 0; k < kSlots; ++k) {
                const u32 bb = post ? 0u : sv >> 5;                     // :319
                sv += bb;
                p[k] = q;
                nx[k] = q + bb;
                sk[k] = sv;
                ok = ok && (post ? k == 0 : nx[k] <= limit);            // :323
                legal[k] = ok;
                q = nx[k];
            }
            const u32 at = ip - (post ? 1u : 0u);                       // ip <= limit = n - 15: in bounds
            if (in_win) {
                // the 8 probe bytes come out of a 16-byte register window that is reloaded when the scan leaves it: a stride-1 scan
                // asks memory for input once per 9 probes instead of once per probe
                u32 o = at - win_at;
                if (o > 8u) {
                    win_at = min(at, n - 16u);                          // at <= limit = n - 15: the window is pulled back inside the fragment
                    win = *reinterpret_cast<const snp_u128_unaligned*>(c.src + win_at);
                    o = at - win_at;                                    // 0 or 1
                }
                const u32 dq = o >> 2, sh = (o & 3u) * 8u;
                const u32 a0 = dq == 0 ? win.v[0] : dq == 1 ? win.v[1] : win.v[2];
                const u32 a1 = dq == 0 ? win.v[1] : dq == 1 ? win.v[2] : win.v[3];
                const u32 a2 = dq == 0 ? win.v[2] : dq == 1 ? win.v[3] : 0u;
                const u64 lo = (static_cast<u64>(a1) << 32) | a0;
                w0 = sh ? (lo >> sh) | (static_cast<u64>(a2) << (64u - sh)) : lo;
            } else {
                w0 = ld64u(c.src + at);
            }
        } else {
#pragma unroll
            for (u32 k = 0; k < kSlots; ++k) { p[k] = nx[k] = sk[k] = 0; legal[k] = false; }
        }
        // ---- trip 1, extension side -------------------------------------------------------------------------------
        const bool ext = mode == kExtend;
        const bool ext_wide = ext && base + mlen + 32 <= n;
        snp_u128_unaligned xa0 = {}, xa1 = {}, xb0 = {}, xb1 = {};
        if (ext_wide) {
            const u8* a = c.src + cand + mlen;
            const u8* bq = c.src + base + mlen;
            xa0 = *reinterpret_cast<const snp_u128_unaligned*>(a);
            xb0 = *reinterpret_cast<const snp_u128_unaligned*>(bq);
            if (!(opts & kOptShortExtend)) {
                xa1 = *reinterpret_cast<const snp_u128_unaligned*>(a + 16);
                xb1 = *reinterpret_cast<const snp_u128_unaligned*>(bq + 16);
            }
        }
        // ---- trip 2: hashes and table entries ----------------------------------------------------------------------
        u32 hm1 = ~0u, vm1 = 0;                                         // the post-copy insert of ip - 1 (bucket, entry)
        bool swapped = false;
        if (scanning) {
            if (post) {                                                 // :393-394
                const u32 dm1 = static_cast<u32>(w0);
                hm1 = lane_hash<VARIANT>(c, dm1, lut);
                vm1 = (ip - 1) | check_bits(dm1);
                // full-size tables: non-temporal -- nothing reads the entry soon, and the probe that one day does is an atomic that goes to
                // L2 / memory anyway (same-workspace interleaved A/B 96.46 -> 96.13 ms, profiles/r04g_compress_insert_store_kinds.json;
                // an agent-scope (sc1) store: 96.41).  Small tables live in L2 and keep the plain store.
                if (tstride == 16384u) __builtin_nontemporal_store(vm1, &c.table[hm1]);
                else c.table[hm1] = vm1;
                d[0] = static_cast<u32>(w0 >> 8);
            } else {
                d[0] = static_cast<u32>(w0);
            }
#pragma unroll
            for (u32 k = 1; k < kSlots; ++k) {
                d[k] = 0;
                if (legal[k]) {
                    const u32 delta = p[k] - p[0];
                    d[k] = delta <= 4 ? static_cast<u32>(w0 >> (8 * delta)) : ld32u(c.src + p[k]);
                }
            }
#pragma unroll
            for (u32 k = 0; k < kSlots; ++k) h[k] = lane_hash<VARIANT>(c, d[k], lut);
            if (t_swap) {
                // one probe per trip and it is always inserted when legal (:333 / :397), so read + insert is one exchange; a probe into
                // the bucket the ip - 1 insert of this same trip just wrote takes that entry from the register instead
                swapped = legal[0] && h[0] != hm1;
                cv[0] = !legal[0] ? 0u : swapped ? table_swap(&c.table[h[0]], p[0] | check_bits(d[0])) : vm1;
            } else {
#pragma unroll
                for (u32 k = 0; k < kSlots; ++k) cv[k] = legal[k] ? c.table[h[k]] : 0u;   // :329 / :396  (position | check bits)
            }
        }
        // ---- extension: compare what trip 1 brought (while the table entries are in flight) -----------------------
        if (ext) {
            bool finished = false;
            if (ext_wide) {
                const u64 x0 = (static_cast<u64>(xa0.v[0] ^ xb0.v[0])) | (static_cast<u64>(xa0.v[1] ^ xb0.v[1]) << 32);
                const u64 x1 = (static_cast<u64>(xa0.v[2] ^ xb0.v[2])) | (static_cast<u64>(xa0.v[3] ^ xb0.v[3]) << 32);
                const u64 x2 = (static_cast<u64>(xa1.v[0] ^ xb1.v[0])) | (static_cast<u64>(xa1.v[1] ^ xb1.v[1]) << 32);
                const u64 x3 = (static_cast<u64>(xa1.v[2] ^ xb1.v[2])) | (static_cast<u64>(xa1.v[3] ^ xb1.v[3]) << 32);
                finished = true;
                if (x0) mlen += static_cast<u32>(__builtin_ctzll(x0)) >> 3;
                else if (x1) mlen += 8 + (static_cast<u32>(__builtin_ctzll(x1)) >> 3);
                else if (opts & kOptShortExtend) { mlen += 16; finished = false; }
                else if (x2) mlen += 16 + (static_cast<u32>(__builtin_ctzll(x2)) >> 3);
                else if (x3) mlen += 24 + (static_cast<u32>(__builtin_ctzll(x3)) >> 3);
                else { mlen += 32; finished = false; }
            } else {                                                    // the last < 32 bytes of the fragment  :562-688
                for (u32 k = 0; k < 4 && !finished; ++k) {
                    if (base + mlen + 8 <= n) {
                        const u64 x = ld64u(c.src + cand + mlen) ^ ld64u(c.src + base + mlen);
                        if (x) { mlen += static_cast<u32>(__builtin_ctzll(x)) >> 3; finished = true; }
                        else mlen += 8;
                    } else {
                        while (base + mlen < n && c.src[cand + mlen] == c.src[base + mlen]) ++mlen;
                        finished = true;
                    }
                }
            }
            if (finished) {
                ip = base + mlen;
                cp_len = mlen;                                         // :371-379
                cp_off = base - cand;
                mode = ip >= limit ? kDone : kPost;                    // :381-384
            }
        }
        // ---- trip 3: candidate / probe / literal bytes for the first probe that can match -------------------------
        if (scanning) {
            // a later probe of the group falling into the bucket of an earlier one sees that probe's entry
#pragma unroll
            for (u32 k = 1; k < kSlots; ++k)
#pragma unroll
                for (u32 i = 0; i < k; ++i)
                    if (legal[k] && h[i] == h[k]) cv[k] = p[i] | check_bits(d[i]);
            u32 kw = kSlots;                                            // first probe whose entry can match
#pragma unroll
            for (u32 k = kSlots; k-- > 0;) {
                const u32 pos = cv[k] & 0xffffu;
                const bool want = legal[k] && (pos == 0 ? c.first4 == d[k] : (cv[k] & 0xffff0000u) == check_bits(d[k]));
                if (want) kw = k;
            }
            u32 wp = 0, wpos = 0;
#pragma unroll
            for (u32 k = 0; k < kSlots; ++k)
                if (kw == k) { wp = p[k]; wpos = cv[k] & 0xffffu; }
            snp_u128_unaligned cb = {}, pb = {}, lb = {};
            if (kw < kSlots) {                                          // wpos < wp <= n - 16: all three loads in bounds
                cb = *reinterpret_cast<const snp_u128_unaligned*>(c.src + wpos);
                pb = *reinterpret_cast<const snp_u128_unaligned*>(c.src + wp);
                if (!post) lb = *reinterpret_cast<const snp_u128_unaligned*>(c.src + next_emit);
            }
            // in-order resolution: probes before kw missed, kw is decided by the bytes, probes after it did not happen
            bool ended = false;
#pragma unroll
            for (u32 k = 0; k < kSlots; ++k) {
                if (!ended && k <= kw) {
                    if (!legal[k]) {
                        if (!post) { ended = true; ip = next_emit; mode = kDone; }   // :323-327 -> emit_remainder
                    } else if (!swapped) {
                        c.table[h[k]] = p[k] | check_bits(d[k]);   // :333 / :397
                    }
                }
            }
            if (!ended) {
                bool hit = false;
                if (kw < kSlots) {
                    const u64 x0 = (static_cast<u64>(cb.v[0] ^ pb.v[0])) | (static_cast<u64>(cb.v[1] ^ pb.v[1]) << 32);
                    const u64 x1 = (static_cast<u64>(cb.v[2] ^ pb.v[2])) | (static_cast<u64>(cb.v[3] ^ pb.v[3]) << 32);
                    hit = static_cast<u32>(x0) == 0;                    // :334 / :398
                    if (hit) {
                        if (!post) {                                    // :347
                            if (staged) { lit_len = wp - next_emit; lit16 = lb; }
                            else op = lane_emit_literal16(c, op, next_emit, wp - next_emit, lb, lit_cap, (opts & kOptBlindLiteral16) != 0);
                        }
                        base = wp;
                        cand = wpos;
                        if (x0) mlen = static_cast<u32>(__builtin_ctzll(x0)) >> 3;
                        else if (x1) mlen = 8 + (static_cast<u32>(__builtin_ctzll(x1)) >> 3);
                        else mlen = 16;
                        if (mlen < 16) {
                            ip = base + mlen;
                            cp_len = mlen;                              // :371-379
                            cp_off = base - cand;
                            mode = ip >= limit ? kDone : kPost;        // :381-384
                        } else {
                            mode = kExtend;
                        }
                    }
                }
                if (!hit) {
                    if (post) {                                         // the probe after a copy missed: next outer iteration
                        next_emit = ip;
                        ++ip;
                        skip = 32;
                        mode = kScan;
                    } else {                                            // probes 0..min(kw, kSlots-1) missed  :339-340
                        const u32 last = kw < kSlots ? kw : kSlots - 1;
#pragma unroll
                        for (u32 k = 0; k < kSlots; ++k)
                            if (k == last) { ip = nx[k]; skip = sk[k]; }
                    }
                }
            }
        }
        if (!staged) {
            if (cp_len) op = lane_emit_copy(c.dst, op, cp_off, cp_len, (opts & kOptBlindCopies) ? lit_cap : 0u);   // after this trip's literal, if any
        } else {
            // literal (next_emit is still the literal's start: it only changes when a scan begins), then the copy
            if (lit_len) {
                if (lit_len <= 15) {                                    // tag + body: one 16-byte LDS store
                    snp_u128_unaligned w;
                    w.v[0] = (lit16.v[0] << 8) | ((lit_len - 1) << 2);
                    w.v[1] = __builtin_amdgcn_alignbit(lit16.v[1], lit16.v[0], 24);
                    w.v[2] = __builtin_amdgcn_alignbit(lit16.v[2], lit16.v[1], 24);
                    w.v[3] = __builtin_amdgcn_alignbit(lit16.v[3], lit16.v[2], 24);
                    *reinterpret_cast<snp_u128_unaligned*>(stg.lds + (op - stg.flushed)) = w;
                    op += 1 + lit_len;
                } else {                                                // long: straight to global memory, around the stage
                    stage_drain(c, stg, op);
                    op = lane_emit_literal16(c, op, next_emit, lit_len, lit16, 0u, false);
                    stg.flushed = op;
                }
            }
            u32 len = cp_len;
            while (len) {                                               // EmitCopy  :507-543, one tag per turn
                const u32 piece = len >= 68 ? 64u : len > 64 ? 60u : len;
                const bool one = piece < 12 && cp_off < 2048;
                const u32 w = one ? (1u | ((piece - 4) << 2) | ((cp_off >> 8) << 5) | ((cp_off & 0xffu) << 8))
                                  : (2u | ((piece - 1) << 2) | (cp_off << 8));
                st32u(stg.lds + (op - stg.flushed), w);
                op += one ? 2u : 3u;
                len -= piece;
                if (op - stg.flushed >= 64) stage_flush64(c, stg);
            }
            if (op - stg.flushed >= 64) stage_flush64(c, stg);
        }
    }
    if (staged) stage_drain(c, stg, op);
#else
    if (n >= 15) {                                                     // :190
        const u32 tsize = n > 16384 ? 16384u : n < 256 ? 256u : (2u << (31u - __clz(n - 1)));   // HashTable.cs:57-71
        c.mask = 2 * (tsize - 1);                                      // :181
        c.hmask = static_cast<u16>(crc_step32(c.mask) & 0x7ffeu);
        const u32 limit = n - 15;                                      // :192
        for (;;) {
            const u32 next_emit = ip;                                  // :198
            ++ip;
            u32 skip = 32;                                             // :227
            u32 cand;
            // scan (:230-341; the unrolled 16-probe section follows the same sequence as the generic loop)
            for (;;) {
                const u32 data = ld32u(c.src + ip);
                const u32 bb = skip >> 5;                              // :319
                skip += bb;
                const u32 nxt = ip + bb;
                if (nxt > limit) { ip = next_emit; goto emit_remainder; }   // :323-327
                const u32 h = lane_hash<VARIANT>(c, data, lut);
                cand = c.table[h] & 0xffffu;                           // :329
                c.table[h] = ip | check_bits(data);                    // :333
                if (ld32u(c.src + cand) == data) break;                // :334
                ip = nxt;
            }
            op = lane_emit_literal(c, op, next_emit, ip - next_emit);  // :347
            for (;;) {                                                 // emit_match  :358-398
                const u32 base = ip;
                const u32 matched = 4 + lane_find_match_length(c.src, cand + 4, ip + 4, n);
                ip += matched;
                op = lane_emit_copy(c.dst, op, base - cand, matched);  // :371-379
                if (ip >= limit) goto emit_remainder;                  // :381-384
                const u32 dm1 = ld32u(c.src + ip - 1);
                c.table[lane_hash<VARIANT>(c, dm1, lut)] = (ip - 1) | check_bits(dm1);   // :393-394
                const u32 data = ld32u(c.src + ip);
                const u32 h = lane_hash<VARIANT>(c, data, lut);
                cand = c.table[h] & 0xffffu;                           // :396
                c.table[h] = ip | check_bits(data);                    // :397
                if (ld32u(c.src + cand) != data) break;                // :398
            }
        }
    }
emit_remainder:
#endif
    if (ip < n) op = lane_emit_literal(c, op, ip, n - ip);             // :406-411
    out_len[b] = op;
    status[b] = SNP_OK;
}

}  // namespace

// ---- workspace placement probe ---------------------------------------------------------------------------------
// The kernel is bound by the rate at which HBM serves random 4-byte exchanges spread over the whole table workspace, and that rate
// depends on WHERE the driver placed the memory (regions of three kinds; the traffic wants to be spread over them: piece_search.h,
// DESIGN.md 4.3).  This probe is the measuring instrument of the search that picks the workspace's pieces (capi_pool.hip, ensure_tables):
// the table traffic of the compressor and nothing else -- every lane walks a chain of dependent exchanges through its own 64 KiB table --
// on a SET of candidate pieces.  A set smaller than the workspace is probed folded (several lanes per table), so that one or two
// pieces see the whole grid's concurrency: one 0.6 GiB piece alone 3.4-3.6 ms per 512 probes when it straddles kinds, 3.9 when it does
// not; two pieces 3.65-3.68 ms when they are of different kinds, 4.31-4.36 when of the same; sixteen pieces 3.8 / 4.0 / 4.75 ms spread over
// three / two / one kind.  (History: rounds 1-2 probed whole 10 GiB candidates with load + store pairs and saw two levels; with exchanges,
// four -- the shares 1, 3/4, 1/2 ... of one kind inside one contiguous allocation.)
namespace {
__global__ __launch_bounds__(SNP_WAVE) void k_probe_tables(const snp_table_pieces tp, u32 nblocks, u32 probes)
{
    const u32 g = blockIdx.x * SNP_WAVE + threadIdx.x;
    if (g >= nblocks) return;
    // a set of FEWER pieces than nblocks fragments need is probed folded: the whole grid's concurrency on the pieces it has
    u32* t = tp.p[(g / tp.piece_frags) % tp.n] + static_cast<size_t>(g % tp.piece_frags) * 16384u;
    u32 st = g * 2654435761u + 1u;
    for (u32 i = 0; i < probes; ++i) {
        const u32 h = (st * 0x1e35a7bdu) >> 18;
        const u32 v = table_swap(&t[h], i);                             // (what the kernel issues since round 3: see below)
        st = st * 1664525u + 1013904223u + v;
    }
    if (st == 0x12345678u) tp.p[0][0] = st;
}
}  // namespace

extern "C" hipError_t snp_probe_tables(const snp_table_pieces* tp, u32 nblocks, u32 probes, hipStream_t stream, float* ms)
{
    hipEvent_t a, b;
    hipError_t e = hipEventCreate(&a);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&b);
    if (e != hipSuccess) { (void)hipEventDestroy(a); return e; }
    const u32 grid = (nblocks + SNP_WAVE - 1) / SNP_WAVE;
    hipLaunchKernelGGL(k_probe_tables, dim3(grid), dim3(SNP_WAVE), 0, stream, *tp, nblocks, 64u);   // warm
    (void)hipEventRecord(a, stream);
    hipLaunchKernelGGL(k_probe_tables, dim3(grid), dim3(SNP_WAVE), 0, stream, *tp, nblocks, probes);
    (void)hipEventRecord(b, stream);
    e = hipEventSynchronize(b);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return e;
}

// test hook (include/snappier_hip_debug.h): FindMatchLength as this kernel's lanes compute it, driven with the reference's KATs
namespace {
__global__ void k_debug_lane_match_length(const u8* buf, u32 n, u32 s1, u32 s2, u32* out)
{
    if (threadIdx.x == 0) *out = lane_find_match_length(buf, s1, s2, n);
}
}  // namespace
extern "C" int snp_debug_lane_match_length(const u8* d_buf, u32 n, u32 s1, u32 s2, u32* d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(k_debug_lane_match_length, dim3(1), dim3(SNP_WAVE), 0, stream, d_buf, n, s1, s2, d_out);
    return static_cast<int>(hipGetLastError());
}

extern "C" size_t snp_compress_lanes_workspace(u32 nblocks) { return static_cast<size_t>(nblocks) * 16384u * sizeof(u32); }

// longest fragment of the batch, on the device (the host never sees the lengths of a device-resident batch)
__global__ __launch_bounds__(256) void k_max_len(const u32* __restrict__ in_len, u32 nblocks, u32* __restrict__ max_len)
{
    u32 m = 0;
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < nblocks; i += gridDim.x * 256) m = max(m, in_len[i]);
    for (int sh = 32; sh >= 1; sh >>= 1) m = max(m, static_cast<u32>(__shfl_xor(static_cast<int>(m), sh, 64)));
    if ((threadIdx.x & 63u) == 0) atomicMax(max_len, m);
}

extern "C" hipError_t snp_launch_compress_lanes(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                                const u64* out_off, u32* out_len, i32* status, int variant,
                                                int emit_varint, const snp_table_pieces* tables, u32* max_len, hipStream_t stream, const snp_lane_tuning* tune)
{
    if (nblocks == 0) return hipSuccess;
    hipError_t e = snp_zero_words_async(max_len, 1, stream);
    if (e != hipSuccess) return e;
    const u32 mgrid = (nblocks + 255) / 256 < 1024 ? (nblocks + 255) / 256 : 1024u;
    hipLaunchKernelGGL(k_max_len, dim3(mgrid), dim3(256), 0, stream, in_len, nblocks, max_len);
    // Fragments per wavefront: 64 when there are enough fragments to fill the chip that way; fewer (partially filled
    // wavefronts, more of them) for mid-sized batches, so that every CU gets several wavefronts to overlap latency.
    // (`tune`: the context's SNP_OPT_COMPRESS_LANE_* settings and its hint from the previous batch; the SNAPPIER_HIP_* variables below exist in
    //  variant builds only -- SNP_GETENV is a null pointer in the product library)
    const char* env = SNP_GETENV("SNAPPIER_HIP_LANES_PER_WAVE");
    const bool two_probes = (tune->hint & 256) != 0;                    // the context's hint: small fragments (capi_batch.hip)
    const u32 small_hint = ((static_cast<u32>(tune->hint) >> 9) & 63u) << 4;   // ... and, when they are small enough for it, the LDS slot size of the SMALL launch
    const u32 lanes_hint = tune->lanes_per_wave ? static_cast<u32>(tune->lanes_per_wave) : static_cast<u32>(tune->hint & 255);
    u32 per = env ? static_cast<u32>(atoi(env)) : lanes_hint ? lanes_hint : (nblocks >= 131072 ? 64u : nblocks >= 8192 ? 32u : 16u);   // measured: scripts/sweep_layouts.py
    if (per != 64 && per != 32 && per != 16 && per != 8) per = 64;
    const u32 grid = (nblocks + per - 1) / per;
    // Output-store options (bit 0: a short literal may overshoot with one 16-byte store, bit 1: tag + body of a literal in
    // one store, bit 2: a copy tag as one 4-byte store, bit 3: 16- instead of 32-byte extension trips, bit 4: output staged
    // in LDS and written as whole 64-byte runs, bit 6: probe + insert as one atomic exchange (table_swap, one-probe-per-trip launches); default
    // 1+2+4+16+64; + bit 7 from 131 072 fragments: the probe bytes out of a 16-byte register window over the input, reloaded when the scan leaves
    // it -- one input load per ~9 probes.  On a one-allocation workspace that measured -0.3 % and was removed; on the piece workspace, where
    // the tables are no longer the only thing the memory system is busy with, same-process A/B: html 99.45 -> 97.29 ms, mixed 146.0 -> 144.6,
    // low entropy 35.75 -> 35.58; mid-size batches lose 1-2 % and do not get it).  SNAPPIER_HIP_EXACT_LITERALS=1 = none (exact-length stores only);
    // SNAPPIER_HIP_CL_OPTS=<mask> picks a subset -- read per launch, so one process can A/B on the same workspace.
    const char* ex = SNP_GETENV("SNAPPIER_HIP_EXACT_LITERALS");
    const char* oe = SNP_GETENV("SNAPPIER_HIP_CL_OPTS");
    // (the LDS staging pays once the memory system is saturated: same-process A/B, 16 384 fragments 37.4 vs 35.5 ms, 65 536: 59.7 vs 61.1)
    const int opts = ((ex && ex[0] == '1') ? 0 : oe ? (atoi(oe) & 255) : tune->opts >= 0 ? (tune->opts & 255)
                      : ((nblocks >= 32768 ? 23 : 7) | 64 | ((nblocks >= 131072 && !two_probes) ? 128 : 0)));   // (kOptExchangeProbe acts in one-probe-per-trip launches only)
    // probes issued together per scan trip (SNAPPIER_HIP_CL_SLOTS=1|2, read per launch; default SNP_CL_SLOTS)
    const char* se = SNP_GETENV("SNAPPIER_HIP_CL_SLOTS");
    // (two probes per trip hide latency while the batch is too small to saturate memory: 10 % faster up to 65 536 fragments;
    // one probe is 1.5 % faster at 163 840)
    // (... and for batches of SMALL fragments at any size -- bit 8 of lanes_per_wave is the context's hint for them: they are latency-bound, not
    // request-bound: 256-byte blocks 40.7 GB/s with one exchange probe per trip, 46.5 with two speculative probes, profiles/r03p_small_compress_sweep.jsonl)
    const u32 slots = se ? static_cast<u32>(atoi(se)) : tune->probes > 0 ? static_cast<u32>(tune->probes) : ((nblocks >= 131072 && !two_probes) ? kDefaultSlots : 2u);
    // The context's hint: the longest fragment of the previous launch, rounded up to 16 bytes, when it lay in (80, 768] (bits 9-14 of
    // lanes_per_wave; measured, profiles/r03y_small_compress_lds_input_sweep.txt: 96 B 41.0 -> 42.8 GB/s, 128 B 42.2 -> 47.2, 256 B 48.1 -> 60.1-61.6,
    // 512 B 47.8 -> 60.7-62.6, 768 B 49.6 -> 51.3; 64 B and 1 KiB lose).  The launch with the input in LDS goes first; each of the two launches
    // checks max_len on the device and returns if the batch is the other one's.  SNAPPIER_HIP_CL_SMALL=0 never launches it, =<bytes> forces
    // its slot size; SNAPPIER_HIP_CL_SMALL_PER = its lanes per wavefront.
    const char* sm = SNP_GETENV("SNAPPIER_HIP_CL_SMALL");
    u32 small_max = sm ? (static_cast<u32>(atoi(sm)) + 15u) & ~15u : tune->small_bytes >= 0 ? (static_cast<u32>(tune->small_bytes) + 15u) & ~15u : small_hint;
    if (small_max > 2048) small_max = 0;
    if (small_max) {
        const char* sp = SNP_GETENV("SNAPPIER_HIP_CL_SMALL_PER");
        u32 sper = sp ? static_cast<u32>(atoi(sp)) : tune->small_lanes ? static_cast<u32>(tune->small_lanes) : 32u;
        if (sper != 64 && sper != 32 && sper != 16) sper = 32;
        const u32 sgrid = (nblocks + sper - 1) / sper;
        const u32 dyn = sper * (small_max + 16u + kStageStride);
        // (The twin general launch below returns early for batches that are this launch's: if this one cannot be launched -- an LDS request a
        //  part refuses -- small_max goes back to 0, so that the general launch takes the batch and no block is left unwritten.)
        bool small_ok = true;
#define SNP_LAUNCH_SMALL(V)                                                                                                                         \
    do {                                                                                                                                            \
        if (dyn > 48u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_compress_lanes<V, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(dyn)) != hipSuccess) { \
            small_ok = false;                                                                                                                       \
            break;                                                                                                                                  \
        }                                                                                                                                           \
        hipLaunchKernelGGL((k_compress_lanes<V, 2, true>), dim3(sgrid), dim3(sper), dyn, stream, in, in_off, in_len, nblocks, out, out_off, out_len, status, \
                           emit_varint, *tables, opts, max_len, small_max);                                                                    \
        small_ok = hipGetLastError() == hipSuccess;                                                                                                 \
    } while (0)
        if (variant == SNP_HASH_CRC32C) SNP_LAUNCH_SMALL(SNP_HASH_CRC32C); else SNP_LAUNCH_SMALL(SNP_HASH_MUL);
#undef SNP_LAUNCH_SMALL
        if (!small_ok) {
            (void)hipGetLastError();
            small_max = 0;
        }
    }
#define SNP_LAUNCH_CL(V, S)                                                                                          \
    hipLaunchKernelGGL((k_compress_lanes<V, S, false>), dim3(grid), dim3(per), 0, stream, in, in_off, in_len, nblocks, out,    \
                       out_off, out_len, status, emit_varint, *tables, opts, max_len, small_max)
    if (variant == SNP_HASH_CRC32C) { if (slots == 1) SNP_LAUNCH_CL(SNP_HASH_CRC32C, 1); else if (slots >= 4) SNP_LAUNCH_CL(SNP_HASH_CRC32C, 4); else if (slots == 3) SNP_LAUNCH_CL(SNP_HASH_CRC32C, 3); else SNP_LAUNCH_CL(SNP_HASH_CRC32C, 2); }
    else { if (slots == 1) SNP_LAUNCH_CL(SNP_HASH_MUL, 1); else if (slots >= 4) SNP_LAUNCH_CL(SNP_HASH_MUL, 4); else if (slots == 3) SNP_LAUNCH_CL(SNP_HASH_MUL, 3); else SNP_LAUNCH_CL(SNP_HASH_MUL, 2); }
#undef SNP_LAUNCH_CL
    return hipGetLastError();
}
