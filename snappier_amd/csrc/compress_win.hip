// compress_win.hip -- Snappy fragment compression, one <= 64 KiB fragment per wavefront, u16 hash table in LDS,
// MULTI-TOKEN speculative rounds (gfx950).  Bit-exact with SnappyCompressor.CompressFragment
// (Snappier/Internal/SnappyCompressor.cs:174-415) for both TableEntry hashes (HashTable.cs:91-126).
// tests/window_model.c is the executable CPU model of exactly this algorithm (checked against the oracle).
//
// The reference parse is a serial chain of probe EVENTS in position order.  Its state is
//     (S, pos, pend):  S = start of the current scan (the OUTER iteration's ip+1, :198-199), pos = next position to
//                      probe, pos == S-1 = the probe that follows a copy (:395-398), pend = "insert pos-1 first" (:393-394)
//     probe p legal iff p == S-1 or p + bb <= limit, bb = (32 + p - S) >> 5     (skip = 32 + distance, :227,319-323;
//                      the unrolled 16-probe section :230-313 obeys the same rule)
//     miss: pos = p + bb (S after the post-copy probe)      hit: literal, copy of len m, ip = p + m;
//                      ip >= limit -> remainder (:381-384), else S = ip+1, pos = ip, pend
// DENSE round (scans in their stride-1 zone, i.e. html/text-like data): all W = 64*NP consecutive positions from pos
// are speculated against the table as it stood BEFORE the round: 16 input bytes per position in registers, hash,
// candidate gather from LDS, 16 candidate bytes from the fragment, hit + match length (up to kCap bytes per lane,
// longer matches are finished by the whole wave only if the parse really visits them).  A scalar walk over ballot
// masks then follows the state machine through the window -- several tokens per round -- and yields the set of
// positions the serial parse writes to the table (probes and ip-1 inserts).  The speculation was exact iff those
// positions have pairwise distinct buckets: they publish to the table and read back; on a repeat the round is cut at
// the first position whose bucket an earlier one of the round already used (min-position-wins publish makes every
// later duplicate read a smaller position), the table is rolled back and the prefix below the cut is kept.
// SPARSE round (scan beyond its 33rd probe -- incompressible data -- and the fragment tail): lane j speculates probe
// slot j of the current scan (offsets D[], the skip heuristic's sequence), at most one token per round.
// Tokens (position, length, offset) queue in LDS and are turned into literal/copy tags 64 at a time, one per lane
// (EmitLiteral :418-464, EmitCopy* :467-543), so the emission cost is paid per 64 tokens, not per round.
// LDS per wavefront: 32 KiB table + 2 KiB CRC-step table + 1 KiB token ring  -> 4 fragments per CU.
#include <cstdlib>

#include "snp_device.h"

#ifndef SNP_W_CAP
#define SNP_W_CAP 32      // bytes of match length every window position resolves by itself (16 or 32: one or two 16-byte pieces)
#endif
#ifndef SNP_W_PROF
#define SNP_W_PROF 0
#endif
#ifndef SNP_W_PREFILTER
#define SNP_W_PREFILTER 1   // global-slot form: distinct buckets proven in LDS before the table is touched (0 = always read the table back)
#endif
#ifndef SNP_W_PREFILTER_BITS
#define SNP_W_PREFILTER_BITS 11    // 2 048 slots = 4 KiB per wavefront: 12 bits cost the dual form its LDS room (46.7 against 49.9 GB/s), 10 and 11 measure the same (profiles/r06e_ab_prefilter.jsonl)
#endif
#ifndef SNP_W_ASMWALK
#define SNP_W_ASMWALK 1   // NP == 1: the walk's common case as a hand-written scalar loop
#endif

namespace {

constexpr u32 kCap = SNP_W_CAP;
constexpr int kPieces = SNP_W_CAP / 16;
constexpr u32 kNone = 0xffffffffu;
constexpr u32 kDupMask = (1u << SNP_W_PREFILTER_BITS) - 1u;

// ---- compile-time tables -----------------------------------------------------------------------------------
struct ProbeTableW {
    u16 d[704];
    constexpr ProbeTableW() : d{}
    {
        u32 v = 0;
        for (int i = 0; i < 704; ++i) {
            d[i] = static_cast<u16>(v > 0xffffu ? 0xffffu : v);      // saturate: anything >= 65536 is illegal anyway
            v = v + 1 + (v >> 5);
        }
    }
};
__device__ const ProbeTableW g_probe_w{};

constexpr u32 crc_step32_w(u32 x)
{
    for (int k = 0; k < 32; ++k) x = (x >> 1) ^ ((x & 1u) ? 0x82F63B78u : 0u);
    return x;
}
// The CRC step is GF(2)-linear: step32(b) = XOR over the four bytes of LUT[k][byte k]   (HashTable.cs:109-112)
struct CrcLutW {
    u16 v[1024];
    constexpr CrcLutW() : v{}
    {
        for (int k = 0; k < 4; ++k)
            for (int b = 0; b < 256; ++b) v[k * 256 + b] = static_cast<u16>(crc_step32_w(static_cast<u32>(b) << (8 * k)) & 0xffffu);
    }
};
__device__ const CrcLutW g_crc_lut_w{};

__device__ __forceinline__ u64 bcast_first64(u64 v)
{
    return (static_cast<u64>(bcast_first(static_cast<u32>(v >> 32))) << 32) | bcast_first(static_cast<u32>(v));
}
__device__ __forceinline__ void lds_fence() { asm volatile("" ::: "memory"); }   // DS ops of one wave execute in order; compiler-only (inside compress_win_fragment: tab_fence<GTAB>)
__device__ __forceinline__ u32 log2_floor_w(u32 v) { return 31u - __clz(v); }

struct __attribute__((packed)) snp_u16_unaligned_w { u16 v; };

// HashTable.TableEntry (HashTable.cs:91-126) -> entry index.  hmask = step32(mask) & 0xffff.
template <int VARIANT>
__device__ __forceinline__ u32 bucket_of(u32 bytes, u32 mask, u32 hmask, const u16* lut)
{
    u32 hash;
    if constexpr (VARIANT == SNP_HASH_CRC32C) {
        hash = lut[bytes & 0xffu] ^ lut[256 + ((bytes >> 8) & 0xffu)] ^ lut[512 + ((bytes >> 16) & 0xffu)] ^ lut[768 + (bytes >> 24)] ^ hmask;
    } else {
        hash = (0x1e35a7bdu * bytes) >> 17;
    }
    return (hash & mask) >> 1;
}

// number of leading equal bytes of two 16-byte pieces (0..16)
__device__ __forceinline__ u32 common16(const snp_u128_unaligned& a, const snp_u128_unaligned& b)
{
    const u32 x0 = a.v[0] ^ b.v[0], x1 = a.v[1] ^ b.v[1], x2 = a.v[2] ^ b.v[2], x3 = a.v[3] ^ b.v[3];
    const u32 xd = x0 ? x0 : x1 ? x1 : x2 ? x2 : x3;
    const u32 base = x0 ? 0u : x1 ? 4u : x2 ? 8u : 12u;
    return xd ? base + (static_cast<u32>(__builtin_ctz(xd)) >> 3) : 16u;
}

// FindMatchLength (:562-688) by the whole wave: src[cand..] vs src[p..], the first `known` bytes are known equal.
// 4 bytes per lane per step; the fragment's last bytes are compared bytewise.  Returns the total match length.
__device__ __forceinline__ u32 wave_match_extend(const u8* src, u32 n, u32 p, u32 cand, u32 known, u32 lane)
{
    u32 base = known;
    for (;;) {
        const u32 o = base + 4 * lane;
        const u32 at = p + o;
        const u32 avail = at < n ? (n - at < 4 ? n - at : 4u) : 0u;
        u32 x = 0, y = 0;
        if (avail == 4) {
            x = ld32u(src + at);
            y = ld32u(src + cand + o);
        } else {
            for (u32 k = 0; k < avail; ++k) {
                x |= static_cast<u32>(src[at + k]) << (8 * k);
                y |= static_cast<u32>(src[cand + o + k]) << (8 * k);
            }
        }
        const u32 d = x ^ y;
        u32 eq = d ? static_cast<u32>(__builtin_ctz(d)) >> 3 : 4u;
        eq = eq < avail ? eq : avail;
        const u64 nf = ballot64(eq != 4);
        if (nf) {
            const u32 fl = static_cast<u32>(__builtin_ctzll(nf));
            return base + 4 * fl + read_lane(eq, fl);
        }
        base += 256;
    }
}

// One lane copies len (1..64) bytes, exact stores; reads 16 bytes at s when len < 16 (caller guarantees they exist).
__device__ __forceinline__ void lane_copy_w(u8* d, const u8* s, u32 len)
{
    const snp_u128_unaligned p0 = *reinterpret_cast<const snp_u128_unaligned*>(s);
    if (len >= 16) {
        snp_u128_unaligned p1 = p0;
        if (len > 16) p1 = *reinterpret_cast<const snp_u128_unaligned*>(s + len - 16);
        *reinterpret_cast<snp_u128_unaligned*>(d) = p0;
        if (len > 16) *reinterpret_cast<snp_u128_unaligned*>(d + len - 16) = p1;
        if (len > 32) {
            *reinterpret_cast<snp_u128_unaligned*>(d + 16) = *reinterpret_cast<const snp_u128_unaligned*>(s + 16);
            if (len > 48) *reinterpret_cast<snp_u128_unaligned*>(d + 32) = *reinterpret_cast<const snp_u128_unaligned*>(s + 32);
        }
    } else {
        const bool c8 = (len & 8u) != 0, c4 = (len & 4u) != 0, c2 = (len & 2u) != 0;
        const u32 a0 = c8 ? p0.v[2] : p0.v[0];
        const u32 a1 = c8 ? p0.v[3] : p0.v[1];
        const u32 b0 = c4 ? a1 : a0;
        const u32 c0 = c2 ? b0 >> 16 : b0;
        const u32 o4 = len & 8u, o2 = len & 12u, o1 = len & 14u;
        if (c8) reinterpret_cast<snp_u64_unaligned*>(d)->v = p0.v[0] | (static_cast<u64>(p0.v[1]) << 32);
        if (c4) st32u(d + o4, a0);
        if (c2) reinterpret_cast<snp_u16_unaligned_w*>(d + o2)->v = static_cast<u16>(b0);
        if (len & 1u) d[o1] = static_cast<u8>(c0);
    }
}

// Inclusive prefix sum across the 64 lanes with DPP row shifts / row broadcasts.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_or_zero_w(u32 v)
{
    return static_cast<u32>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ u32 wave_inclusive_scan_w(u32 x)
{
    u32 y = x + dpp_or_zero_w<0x111, 0xf>(x);
    y += dpp_or_zero_w<0x112, 0xf>(x);
    y += dpp_or_zero_w<0x113, 0xf>(x);
    y += dpp_or_zero_w<0x114, 0xf>(y);
    y += dpp_or_zero_w<0x118, 0xf>(y);
    y += dpp_or_zero_w<0x142, 0xa>(y);
    y += dpp_or_zero_w<0x143, 0xc>(y);
    return y;
}

// EmitLiteral of a long run by the whole wave (:418-464): tag (+ length bytes) by the first lanes, body 16 B per lane.
__device__ __forceinline__ u32 wave_emit_literal(u8* dst, u32 op, const u8* src, u32 s, u32 len, u32 lane)
{
    const u32 k = len - 1;
    u32 hdr;
    if (k < 60) {
        if (lane == 0) dst[op] = static_cast<u8>(k << 2);
        hdr = 1;
    } else {
        const u32 count = (log2_floor_w(k) >> 3) + 1;                   // :447
        if (lane == 0) dst[op] = static_cast<u8>((59 + count) << 2);    // :451
        if (lane >= 1 && lane <= count) dst[op + lane] = static_cast<u8>(k >> (8 * (lane - 1)));
        hdr = 1 + count;
    }
    wave_copy(dst + op + hdr, src + s, len, lane);
    return op + hdr + len;
}

// ---- bit masks over the window positions, NP words, wave-uniform ---------------------------------------------
// first set bit at or above o (o < 64*NP); 64*NP if there is none
template <int NP>
__device__ __forceinline__ u32 mask_first_from(const u64 (&w)[NP], u32 o)
{
    if constexpr (NP == 1) {
        const u64 x = w[0] >> o;
        return x ? o + static_cast<u32>(__builtin_ctzll(x)) : 64u;
    } else {
        u32 r = 64u * NP;
#pragma unroll
        for (int k = NP - 1; k >= 0; --k) {
            const u32 base = 64u * k;
            if (o < base + 64) {
                const u64 x = o > base ? (w[k] >> (o - base)) << (o - base) : w[k];
                if (x) r = base + static_cast<u32>(__builtin_ctzll(x));
            }
        }
        return r;
    }
}
template <int NP>
__device__ __forceinline__ void mask_set_bit(u64 (&w)[NP], u32 a)
{
    if constexpr (NP == 1) w[0] |= 1ull << a;
    else {
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if ((a >> 6) == static_cast<u32>(k)) w[k] |= 1ull << (a & 63u);
    }
}
template <int NP>
__device__ __forceinline__ bool mask_test(const u64 (&w)[NP], u32 a)
{
    if constexpr (NP == 1) return (w[0] >> a) & 1ull;
    else {
        bool r = false;
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if ((a >> 6) == static_cast<u32>(k)) r = (w[k] >> (a & 63u)) & 1ull;
        return r;
    }
}
// set bits a..b inclusive; nothing if a > b; bits >= 64*NP are dropped
template <int NP>
__device__ __forceinline__ void mask_set_range(u64 (&w)[NP], u32 a, u32 b)
{
    if (a > b) return;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const u32 base = 64u * k;
        if (a < base + 64 && b >= base) {
            const u32 lo = a > base ? a - base : 0u;
            const u32 hi = b - base < 63 ? b - base : 63u;
            w[k] |= (~0ull << lo) & (~0ull >> (63u - hi));
        }
    }
}

template <int NP>
__device__ __forceinline__ u32 read_half(const u32 (&v)[NP], u32 q)
{
    if constexpr (NP == 1) return read_lane(v[0], q);
    else {
        u32 r = 0;
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if ((q >> 6) == static_cast<u32>(k)) r = read_lane(v[k], q & 63u);
        return r;
    }
}

struct ParseState {
    u32 S, pos, kb;        // kb: pos == S + D[kb] whenever pos >= S
    bool pend, done;
};

// SNP_W_PROF=1: event counters (slots 0-7) ; =2: plus phase timers (slots 8-15, s_memtime with a drain at every mark)
#if SNP_W_PROF
__device__ unsigned long long g_wprof[16];
#define WPROF_ADD(k, v) do { wacc[k] += static_cast<unsigned long long>(v); } while (0)
#define WPROF_DECL unsigned long long wacc[16] = {0}; u64 wt_ = __builtin_readcyclecounter(); (void)wt_;
#define WPROF_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 16; ++k_) if (wacc[k_]) atomicAdd(&g_wprof[k_], wacc[k_]); } while (0)
#if SNP_W_PROF >= 2
#define WPROF_T(k)                                                                \
    do {                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               \
        const u64 now_ = __builtin_readcyclecounter();                            \
        wacc[k] += now_ - wt_;                                                    \
        wt_ = now_;                                                               \
    } while (0)
#else
#define WPROF_T(k)
#endif
#else
#define WPROF_ADD(k, v)
#define WPROF_DECL
#define WPROF_FLUSH
#define WPROF_T(k)
#endif

// Where a wavefront's hash table lives.  GTAB = false: 32 KiB of LDS (the north-star layout; 4 fragments per CU, one wavefront per SIMD).
// GTAB = true: a 32 KiB slot of a global-memory workspace that stays in L2 / Infinity Cache (8 192 slots = 256 MiB) -- the wavefront then
// needs 3 KiB of LDS and a CU holds 32 of them: the lone wavefront's issue rate (one instruction per ~6 cycles) stops being the bound for
// batches of more than ~2 000 fragments.  Entries move with agent-scope (sc1) accesses, which bypass the CU's L1, and every point where
// the LDS form relies on DS operations executing in order drains vmcnt instead: a store is acknowledged by L2 before a lane reads it back.
template <bool GTAB>
struct WinTable {
    u16* p;
    __device__ __forceinline__ u32 get(u32 i) const
    {
        if constexpr (GTAB) return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return p[i];
    }
    __device__ __forceinline__ void set(u32 i, u32 v) const
    {
        if constexpr (GTAB) __hip_atomic_store(p + i, static_cast<u16>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else p[i] = static_cast<u16>(v);
    }
    __device__ __forceinline__ void zero8(u32 i) const { *reinterpret_cast<uint4*>(p + i) = make_uint4(0, 0, 0, 0); }
};
template <bool GTAB>
__device__ __forceinline__ void tab_fence()
{
    if constexpr (GTAB) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("" ::: "memory");
}

template <int VARIANT, int NP, bool GTAB>
__device__ __forceinline__ void compress_win_fragment(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                      const u32* __restrict__ in_len, u32 nblocks,
                                                      u8* __restrict__ out, const u64* __restrict__ out_off,
                                                      u32* __restrict__ out_len, i32* __restrict__ status,
                                                      int emit_varint, const u32 b, u16* gtab)
{
    constexpr u32 W = 64u * NP;
    __shared__ u16 table_lds[GTAB ? 8 : 16384];                         // HashTable.cs:17-18
    __shared__ u16 lut[VARIANT == SNP_HASH_CRC32C ? 1024 : 8];
    __shared__ u64 ring[128];                                           // tokens: position | length << 16 | offset << 32
    __shared__ u16 dupf[(GTAB && SNP_W_PREFILTER) ? (1u << SNP_W_PREFILTER_BITS) : 8];         // global-slot form: the distinct-bucket prefilter (see the publish step)
    const WinTable<GTAB> table{GTAB ? gtab : table_lds};
#define lds_fence tab_fence<GTAB>

    const u32 lane = lane_id();
    const u8* src = in + in_off[b];
    const u32 n = bcast_first(in_len[b]);
    u8* dst = out + out_off[b];
    if (n > SNP_BLOCK_SIZE) {
        if (lane == 0) { out_len[b] = 0; status[b] = SNP_ERR_BAD_ARG; }
        return;
    }

    u32 op = 0;
    if (emit_varint) {                                                  // VarIntEncoding.TryWrite  VarIntEncoding.Write.cs:5-79
        const u32 hb = n < (1u << 7) ? 1 : n < (1u << 14) ? 2 : 3;
        if (lane < hb) dst[lane] = static_cast<u8>((n >> (7 * lane)) | (lane + 1 < hb ? 0x80u : 0u));
        op = hb;
    }

    u32 eprev = 0;            // input position up to which output has been produced
    if (n >= 15) {                                                      // Constants.InputMarginBytes  :190
        const u32 tsize = n > 16384 ? 16384u : n < 256 ? 256u : (2u << log2_floor_w(n - 1));   // HashTable.cs:57-71
        const u32 mask = 2 * (tsize - 1);                               // :181
        for (u32 i = lane * 8; i < tsize; i += 64 * 8) table.zero8(i);
        u32 hmask = 0;
        if constexpr (VARIANT == SNP_HASH_CRC32C) {
            for (u32 i = lane; i < 1024; i += 64) lut[i] = g_crc_lut_w.v[i];
            hmask = crc_step32_w(mask) & 0xffffu;
        }
        lds_fence();
        const u32 limit = n - 15;                                       // :192

        ParseState st{1u, 1u, 0u, false, false};
        u32 head = 0, cnt = 0;                                          // token ring
        snp_u128_unaligned X[NP][kPieces];                              // the window's input bytes: kCap per position
        u32 xn_base = kNone;                                            // window start X was prefetched for
        WPROF_DECL

        // ---- emission of `nb` queued tokens, one per lane ------------------------------------------------------
        auto emit_batch = [&](u32 nb) {
            const bool act = lane < nb;
            const u64 tk = act ? ring[(head + lane) & 127u] : 0ull;
            const u32 t = static_cast<u32>(tk) & 0xffffu, len = static_cast<u32>(tk >> 16) & 0xffffu, off = static_cast<u32>(tk >> 32) & 0xffffu;
            const u32 endp = t + len;
            u32 prev = static_cast<u32>(__shfl_up(static_cast<int>(endp), 1));
            if (lane == 0) prev = eprev;
            const u32 ll = act ? t - prev : 0u;                         // literal before this copy (may be empty)
            const u32 k = ll - 1;
            const u32 lh = ll == 0 ? 0u : k < 60 ? 1u : k < 256 ? 2u : 3u;
            // EmitCopy in closed form (:507-543): q tags of 64, optionally one of 60, then the final 4..64-byte tag
            const u32 q = len >= 68 ? (len - 4) >> 6 : 0u;
            u32 r = len - (q << 6);
            const bool c60 = r > 64;
            if (c60) r -= 60;
            const bool one = r < 12 && off < 2048;
            const u32 csz = 3 * q + (c60 ? 3u : 0u) + (one ? 2u : 3u);
            const u32 sz = act ? lh + ll + csz : 0u;
            const u32 incl = wave_inclusive_scan_w(sz);
            const u32 total = read_lane(incl, 63);
            u8* o = dst + op + (incl - sz);
            // literal
            const u64 big = ballot64(act && ll > 64);
            if (act && ll) {
                if (k < 60) o[0] = static_cast<u8>(k << 2);
                else if (k < 256) { o[0] = static_cast<u8>(60u << 2); o[1] = static_cast<u8>(k); }
                else { o[0] = static_cast<u8>(61u << 2); o[1] = static_cast<u8>(k); o[2] = static_cast<u8>(k >> 8); }
                if (ll <= 64) {
                    if (prev + 16 <= n) lane_copy_w(o + lh, src + prev, ll);
                    else for (u32 i = 0; i < ll; ++i) o[lh + i] = src[prev + i];
                }
            }
            // copy tags
            if (act) {
                u8* c = o + lh + ll;
                for (u32 i = 0; i < q; ++i) { c[0] = static_cast<u8>(2u | (63u << 2)); c[1] = static_cast<u8>(off); c[2] = static_cast<u8>(off >> 8); c += 3; }
                if (c60) { c[0] = static_cast<u8>(2u | (59u << 2)); c[1] = static_cast<u8>(off); c[2] = static_cast<u8>(off >> 8); c += 3; }
                if (one) { c[0] = static_cast<u8>(1u | ((r - 4) << 2) | ((off >> 8) << 5)); c[1] = static_cast<u8>(off); }
                else { c[0] = static_cast<u8>(2u | ((r - 1) << 2)); c[1] = static_cast<u8>(off); c[2] = static_cast<u8>(off >> 8); }
            }
            // literals longer than 64 bytes: whole-wave copies
            u64 bg = big;
            while (bg) {
                const u32 f = static_cast<u32>(__builtin_ctzll(bg));
                bg &= bg - 1;
                const u32 f_o = read_lane(incl - sz, f) + read_lane(lh, f);
                wave_copy(dst + op + f_o, src + read_lane(prev, f), read_lane(ll, f), lane);
            }
            op += total;
            eprev = read_lane(endp, nb - 1);
            head = (head + nb) & 127u;
            cnt -= nb;
        };

        while (!st.done) {
            st.S = bcast_first(st.S); st.pos = bcast_first(st.pos); st.kb = bcast_first(st.kb);
            st.pend = bcast_first(st.pend ? 1u : 0u) != 0;
            op = bcast_first(op); eprev = bcast_first(eprev); head = bcast_first(head); cnt = bcast_first(cnt);
            xn_base = bcast_first(xn_base);
            const u32 w = st.pos - (st.pend ? 1u : 0u);
            const bool zone1 = (st.pos + 1 == st.S) || (st.pos - st.S <= 32);
            u32 cut0 = 0;
            if (w + kCap + 1 <= n) { cut0 = n - kCap - w; if (cut0 > W) cut0 = W; }   // positions p with p + kCap + 1 <= n
            WPROF_T(15);
            if (zone1 && cut0 >= 16) {
                // ================================ dense round ================================================
                WPROF_ADD(0, 1);
                u32 pp[NP], h[NP], c[NP], m[NP];
                bool valid[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const u32 q = 64u * k + lane;
                    pp[k] = w + q;
                    valid[k] = q < cut0;
                }
                if (xn_base != w) {                                     // not prefetched by the previous round
#pragma unroll
                    for (int k = 0; k < NP; ++k)
#pragma unroll
                        for (int j = 0; j < kPieces; ++j) {
                            if (valid[k]) X[k][j] = *reinterpret_cast<const snp_u128_unaligned*>(src + pp[k] + 16 * j);
                            else X[k][j].v[0] = X[k][j].v[1] = X[k][j].v[2] = X[k][j].v[3] = 0;
                        }
                }
                WPROF_T(8);                                             // window load
#pragma unroll
                for (int k = 0; k < NP; ++k) h[k] = bucket_of<VARIANT>(X[k][0].v[0], mask, hmask, lut);
                lds_fence();
#pragma unroll
                for (int k = 0; k < NP; ++k) c[k] = table.get(h[k]);
                WPROF_T(9);                                             // hash + table gather
                u64 HITM[NP], UNRES[NP];
                {
                    snp_u128_unaligned E[NP][kPieces];
#pragma unroll
                    for (int k = 0; k < NP; ++k)
#pragma unroll
                        for (int j = 0; j < kPieces; ++j)
                            if (valid[k]) E[k][j] = *reinterpret_cast<const snp_u128_unaligned*>(src + c[k] + 16 * j);
                            else E[k][j].v[0] = E[k][j].v[1] = E[k][j].v[2] = E[k][j].v[3] = ~0u;
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        u32 mm = common16(X[k][0], E[k][0]);
#pragma unroll
                        for (int j = 1; j < kPieces; ++j)
                            if (mm == 16u * j) mm += common16(X[k][j], E[k][j]);
                        m[k] = mm;
                        HITM[k] = ballot64(valid[k] && mm >= 4);
                        UNRES[k] = ballot64(valid[k] && mm == kCap);
                    }
                }
                WPROF_T(10);                                            // candidate gather + compare

                // ---- scalar walk over [0, cut): which positions does the serial parse write (PUB), which of them are
                //      copies (TOKM), where does it stand afterwards (e) -----------------------------------------------
                u64 PUB[NP], TOKM[NP];
                ParseState e;
                auto walk = [&](u32 cut) {
                    u64 SK[NP];                                         // interiors of the copies: positions never visited
#pragma unroll
                    for (int k = 0; k < NP; ++k) { SK[k] = 0; TOKM[k] = 0; }
                    e = st;
                    e.pend = false;
                    u32 o = st.pos - w;                                 // 0, or 1 behind the pending insert
                    u32 srel = st.S - w;                                // scan start relative to the window (may wrap below 0: only srel + 32 is used)
                    u32 end;
                    for (;;) {
                        if constexpr (NP == 1 && SNP_W_ASMWALK) {
                            // The common tokens -- resolved length < 60, copy ends inside the window -- in a hand-written scalar loop
                            // (a lone wavefront issues one instruction every ~6 cycles: the walk is priced per instruction).
                            // Anything else (no further hit, unresolved or long match, copy crossing the cut) falls out to the
                            // generic code below, which handles one event and comes back.
                            u64 x_;
                            u32 t_, z_, ml_, ip_, a_, b_;
                            const u32 cut_u = bcast_first(cut), cm1 = cut_u - 1;
                            const u64 hit_u = bcast_first64(HITM[0]), unres_u = bcast_first64(UNRES[0]);
                            o = bcast_first(o); srel = bcast_first(srel);
                            TOKM[0] = bcast_first64(TOKM[0]); SK[0] = bcast_first64(SK[0]);
                            asm volatile(
                                "1:\n\t"
                                "s_lshr_b64 %[x], %[hit], %[o]\n\t"
                                "s_cmp_eq_u64 %[x], 0\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                "s_ff1_i32_b64 %[t], %[x]\n\t"
                                "s_add_u32 %[t], %[t], %[o]\n\t"
                                "s_add_u32 %[z], %[srel], 32\n\t"
                                "s_min_u32 %[z], %[z], %[cm1]\n\t"
                                "s_cmp_gt_u32 %[t], %[z]\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                "s_bitcmp1_b64 %[unres], %[t]\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                "v_readlane_b32 %[ml], %[vm], %[t]\n\t"
                                "s_nop 0\n\t"
                                "s_cmp_gt_u32 %[ml], 59\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                "s_add_u32 %[ip], %[t], %[ml]\n\t"
                                "s_cmp_ge_u32 %[ip], %[cut]\n\t"
                                "s_cbranch_scc1 2f\n\t"
                                "s_bitset1_b64 %[tok], %[t]\n\t"
                                "s_sub_u32 %[a], %[ml], 2\n\t"
                                "s_add_u32 %[b], %[t], 1\n\t"
                                "s_bfm_b64 %[x], %[a], %[b]\n\t"
                                "s_or_b64 %[sk], %[sk], %[x]\n\t"
                                "s_add_u32 %[srel], %[ip], 1\n\t"
                                "s_mov_b32 %[o], %[ip]\n\t"
                                "s_branch 1b\n\t"
                                "2:\n\t"
                                : [o] "+s"(o), [srel] "+s"(srel), [tok] "+s"(TOKM[0]), [sk] "+s"(SK[0]), [x] "=&s"(x_), [t] "=&s"(t_),
                                  [z] "=&s"(z_), [ml] "=&s"(ml_), [ip] "=&s"(ip_), [a] "=&s"(a_), [b] "=&s"(b_)
                                : [hit] "s"(hit_u), [unres] "s"(unres_u), [vm] "v"(m[0]), [cut] "s"(cut_u), [cm1] "s"(cm1)
                                : "scc");
                            e.S = w + srel;
                        }
                        const u32 t = mask_first_from<NP>(HITM, o);
                        const u32 zend = srel + 32;
                        const u32 lim = zend < cut - 1 ? zend : cut - 1;
                        if (t > lim) {                                  // no hit before the zone / window ends
                            end = lim;
                            if (lim == zend) { e.pos = w + srel + 34; e.kb = 33; }   // probe 33 lies two bytes further (:319-320)
                            else { e.pos = w + cut; e.kb = e.pos >= e.S ? e.pos - e.S : 0u; }
                            break;
                        }
                        mask_set_bit<NP>(TOKM, t);
                        u32 ml = read_half<NP>(m, t);
                        if (mask_test<NP>(UNRES, t)) {
                            WPROF_ADD(3, 1);
                            WPROF_T(12);
                            ml = wave_match_extend(src, n, w + t, read_half<NP>(c, t), ml, lane);
#pragma unroll
                            for (int k = 0; k < NP; ++k) {
                                if (64u * k + lane == t) m[k] = ml;
                                if ((t >> 6) == static_cast<u32>(k)) UNRES[k] &= ~(1ull << (t & 63u));
                            }
                            WPROF_T(11);                                // whole-wave match extension
                        }
                        const u32 ip = t + ml;
                        if (w + ip >= limit) { e.done = true; e.pos = w + ip; end = t; break; }   // :381-384 (no ip-1 insert)
                        mask_set_range<NP>(SK, t + 1, ip - 2);          // ip-1 is inserted (:393-394), ip is probed (:395-398)
                        e.S = w + ip + 1;
                        srel = ip + 1;
                        if (ip >= cut) { e.pos = w + ip; e.kb = 0; e.pend = ip - 1 >= cut; end = cut - 1; break; }
                        o = ip;
                    }
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        u64 r = 0;
                        if (end >= 64u * k) r = end - 64u * k >= 63 ? ~0ull : (2ull << (end - 64u * k)) - 1ull;
                        PUB[k] = r & ~SK[k];
                    }
                };
                walk(cut0);
                WPROF_T(12);                                            // walk
                // prefetch the next window while this one is published and queued (a cut below moves the window: reload then)
                xn_base = kNone;
                if (!e.done) {
                    const u32 wn = e.pos - (e.pend ? 1u : 0u);
                    const bool z1 = (e.pos + 1 == e.S) || (e.pos - e.S <= 32);
                    u32 cn = 0;
                    if (wn + kCap + 1 <= n) { cn = n - kCap - wn; if (cn > W) cn = W; }
                    if (z1 && cn >= 16) {
                        xn_base = wn;
#pragma unroll
                        for (int k = 0; k < NP; ++k)
#pragma unroll
                            for (int j = 0; j < kPieces; ++j) {
                                if (64u * k + lane < cn) X[k][j] = *reinterpret_cast<const snp_u128_unaligned*>(src + wn + 64u * k + lane + 16 * j);
                                else X[k][j].v[0] = X[k][j].v[1] = X[k][j].v[2] = X[k][j].v[3] = 0;
                            }
                    }
                }
                // ---- publish, read back: pairwise distinct buckets? ----------------------------------------------------
                bool pub[NP];
                u32 rb[NP];
                bool bad = false;
#pragma unroll
                for (int k = 0; k < NP; ++k) pub[k] = (PUB[k] >> lane) & 1ull;
                // Global-slot form (round 6): a PREFILTER in LDS first.  The publishing lanes drop their lane number into dupf[bucket mod 4096] and read it
                // back: when every lane reads its own number the low 12 bits of the buckets are pairwise distinct, hence the buckets are, and the table
                // stores below need no read-back -- one L2 round trip less in ~4 of 5 rounds (the read-back was 44 % of this form's clock together with the
                // cut path, profiles/r06_compress_win_phase_clock.jsonl).  A collision in dupf (a real repeated bucket, or two buckets that share their
                // low bits) takes the exact path through the table as before: the prefilter never decides a cut.
                bool exact_check = true;
                if constexpr (GTAB && SNP_W_PREFILTER) {
                    static_assert(NP == 1, "the global-slot form runs with one position per lane");
                    if (pub[0]) dupf[h[0] & kDupMask] = static_cast<u16>(lane);
                    asm volatile("" ::: "memory");                      // (DS operations of a wavefront execute in order)
                    const u32 rr = pub[0] ? dupf[h[0] & kDupMask] : lane;
                    exact_check = ballot64(rr != lane) != 0ull;
                }
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (pub[k]) table.set(h[k], pp[k]);
                if (exact_check) {
                    lds_fence();
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        rb[k] = pub[k] ? table.get(h[k]) : pp[k];
                        bad = bad || rb[k] != pp[k];
                    }
                }
                if (exact_check && ballot64(bad)) {
                    WPROF_ADD(1, 1);
                    // min-position-wins: afterwards every later duplicate reads a smaller position than its own
                    for (;;) {
                        bool again = false;
#pragma unroll
                        for (int k = 0; k < NP; ++k)
                            if (pub[k] && rb[k] > pp[k]) { table.set(h[k], pp[k]); again = true; }
                        if (!ballot64(again)) break;
                        lds_fence();
#pragma unroll
                        for (int k = 0; k < NP; ++k) rb[k] = pub[k] ? table.get(h[k]) : pp[k];
                    }
                    u32 q2 = kNone;
#pragma unroll
                    for (int k = NP - 1; k >= 0; --k) {
                        const u64 los = ballot64(pub[k] && rb[k] < pp[k]);
                        if (los) q2 = 64u * k + static_cast<u32>(__builtin_ctzll(los));
                    }
                    // roll the table back and keep the prefix below q2 (distinct buckets by minimality of q2).  No second walk: the
                    // events before q2 are those of the first one, and the state at q2 follows from the last token before it
                    // (tests/window_model.c checks this against a re-walk).
                    lds_fence();
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                        if (pub[k]) table.set(h[k], c[k]);
                    lds_fence();
                    u32 last = kNone;
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const u64 below = q2 >= 64u * (k + 1) ? ~0ull : q2 <= 64u * k ? 0ull : ((1ull << (q2 - 64u * k)) - 1ull);
                        PUB[k] &= below;
                        TOKM[k] &= below;
                        if (TOKM[k]) last = 64u * k + 63u - static_cast<u32>(__builtin_clzll(TOKM[k]));
                    }
                    e = st;
                    e.pend = false;
                    if (last == kNone) e.pos = w + q2;
                    else {
                        const u32 ip = last + read_half<NP>(m, last);
                        e.S = w + ip + 1;
                        if (q2 + 1 == ip) { e.pos = w + ip; e.pend = true; }   // q2 is that copy's ip-1 insert: still pending
                        else e.pos = w + q2;
                    }
                    e.kb = e.pos >= e.S ? e.pos - e.S : 0u;
                    xn_base = kNone;                                    // the prefetched window started somewhere else
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        pub[k] = (PUB[k] >> lane) & 1ull;
                        if (pub[k]) table.set(h[k], pp[k]);
                    }
                    lds_fence();
                }
                // ---- queue the tokens of the accepted prefix ------------------------------------------------------
                u32 pushed = 0;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const u64 tm = TOKM[k];
                    if ((tm >> lane) & 1ull) {
                        const u32 idx = cnt + pushed + static_cast<u32>(__builtin_popcountll(tm & lanes_below(lane)));
                        ring[(head + idx) & 127u] = static_cast<u64>(pp[k]) | (static_cast<u64>(m[k]) << 16) | (static_cast<u64>(pp[k] - c[k]) << 32);
                    }
                    pushed += static_cast<u32>(__builtin_popcountll(tm));
                }
                WPROF_ADD(2, pushed);
                cnt += pushed;
                st = e;
                if constexpr (GTAB && SNP_W_PREFILTER) asm volatile("" ::: "memory");   // (the table stores are waited for where the next round reads the table; the ring is LDS)
                else lds_fence();
                WPROF_T(13);                                            // publish / cut / queue
            } else {
                // ================================ sparse round ===============================================
                WPROF_ADD(4, 1);
                xn_base = kNone;
                const u32 sh = st.pend ? 1u : 0u;
                const bool postcopy = st.pos + 1 == st.S;
                const bool is_ins = st.pend && lane == 0;
                const u32 i = lane - sh;                                // probe slot (lanes >= sh)
                u32 kidx, p, nxt;
                bool legal;
                if (is_ins) { kidx = 0; p = st.pos - 1; nxt = st.pos; legal = true; }
                else if (postcopy) {
                    if (i == 0) { kidx = 0; p = st.S - 1; nxt = st.S; legal = true; }
                    else {
                        kidx = i - 1;
                        p = st.S + g_probe_w.d[kidx];
                        nxt = st.S + g_probe_w.d[kidx + 1];
                        legal = nxt <= limit;
                    }
                } else {
                    kidx = st.kb + i;
                    kidx = kidx < 702 ? kidx : 702u;
                    p = st.S + g_probe_w.d[kidx];
                    nxt = st.S + g_probe_w.d[kidx + 1];
                    legal = nxt <= limit;
                }
                const u32 d = legal ? ld32u(src + p) : 0u;
                const u32 h = bucket_of<VARIANT>(d, mask, hmask, lut);
                lds_fence();
                const u32 c = table.get(h);
                const u32 ev = (legal && !is_ins) ? ld32u(src + c) : ~d;
                const bool hitl = legal && !is_ins && ev == d;
                const u64 lmask = ballot64(legal);
                const u64 stop = ballot64(hitl) | ~lmask;
                const u32 first0 = stop ? static_cast<u32>(__builtin_ctzll(stop)) : 64u;
                const bool is_hit = first0 < 64 && ((lmask >> first0) & 1ull);
                const bool terminated = first0 < 64 && !is_hit;
                u32 last = is_hit ? first0 + 1 : first0;                // slots [0, last) are this round's events
                // publish / read back / cut
                bool pub = lane < last;
                if (pub) table.set(h, p);
                lds_fence();
                u32 rb = pub ? table.get(h) : p;
                bool cut = false;
                if (ballot64(rb != p)) {
                    WPROF_ADD(5, 1);
                    for (;;) {
                        const bool again = pub && rb > p;
                        if (again) table.set(h, p);
                        if (!ballot64(again)) break;
                        lds_fence();
                        rb = pub ? table.get(h) : p;
                    }
                    const u64 los = ballot64(pub && rb < p);
                    const u32 q2 = static_cast<u32>(__builtin_ctzll(los));   // los != 0: some lane lost to an earlier one
                    lds_fence();
                    if (pub) table.set(h, c);
                    lds_fence();
                    last = q2;
                    cut = true;
                    pub = lane < last;
                    if (pub) table.set(h, p);
                    lds_fence();
                }
                if (cut) {
                    // resume at slot `last` (>= 1): its position becomes pos
                    st.pend = false;
                    if (postcopy && last == sh) { /* the post-copy probe itself: pos stays S-1 */ }
                    else { st.pos = read_lane(p, last); st.kb = read_lane(kidx, last); }
                } else if (is_hit) {
                    const u32 t = read_lane(p, first0), cd = read_lane(c, first0);
                    const u32 ml = wave_match_extend(src, n, t, cd, 4, lane);
                    if (lane == 0) ring[(head + cnt) & 127u] = static_cast<u64>(t) | (static_cast<u64>(ml) << 16) | (static_cast<u64>(t - cd) << 32);
                    cnt += 1;
                    WPROF_ADD(2, 1);
                    const u32 ip = t + ml;
                    st.pend = false;
                    if (ip >= limit) { st.done = true; st.pos = ip; }
                    else { st.S = ip + 1; st.pos = ip; st.kb = 0; st.pend = true; }
                } else if (terminated) {
                    st.done = true;
                } else {
                    st.pend = false;
                    st.pos = read_lane(nxt, 63);
                    st.kb = read_lane(kidx, 63) + 1;
                }
                lds_fence();
            }
            if (cnt >= 64) { WPROF_T(15); emit_batch(64); WPROF_T(14); }
        }
        if (cnt) emit_batch(cnt);
        WPROF_FLUSH;
    }
    if (eprev < n) op = wave_emit_literal(dst, op, src, eprev, n - eprev, lane);   // emit_remainder  :406-411

    if (lane == 0) {
        out_len[b] = op;
        status[b] = SNP_OK;
    }
}
#undef lds_fence

template <int VARIANT, int NP>
__global__ __launch_bounds__(SNP_WAVE) void k_compress_win(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                          const u32* __restrict__ in_len, u32 nblocks,
                                                          u8* __restrict__ out, const u64* __restrict__ out_off,
                                                          u32* __restrict__ out_len, i32* __restrict__ status,
                                                          int emit_varint)
{
    if (blockIdx.x >= nblocks) return;
    compress_win_fragment<VARIANT, NP, false>(in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, blockIdx.x, nullptr);
}

// The same with the table in a global-memory slot (see WinTable): a persistent grid of at most `slots` wavefronts, wavefront w owns slot w
// of `tables` and takes fragments w, w + grid, w + 2 grid, ...  (waves_per_eu: 58 VGPRs allow eight wavefronts per SIMD)
template <int VARIANT, int NP>
__global__ __launch_bounds__(SNP_WAVE) void k_compress_win_g(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                            const u32* __restrict__ in_len, u32 nblocks,
                                                            u8* __restrict__ out, const u64* __restrict__ out_off,
                                                            u32* __restrict__ out_len, i32* __restrict__ status,
                                                            int emit_varint, u16* __restrict__ tables)
{
    u16* const mine = tables + static_cast<size_t>(blockIdx.x) * 16384u;
    for (u32 b = blockIdx.x; b < nblocks; b += gridDim.x) {
        compress_win_fragment<VARIANT, NP, true>(in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, b, mine);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the next fragment reuses the slot and the LDS arrays)
    }
}

// DUAL form (round 6): both table forms at once, as two kernels on two streams that draw fragments from ONE ticket counter.  The LDS form is capped
// by LDS at 4 wavefronts per CU (one per SIMD: 43 % of its cycles issuing, 54 % parked -- profiles/r06_compress_win_pmc.txt), which leaves the SIMDs'
// issue slots and the L2 mostly idle; a second population of global-slot wavefronts, FEW enough that their 32 KiB slots stay L2-resident (~10 per CU:
// 2 560 slots = 80 MiB over 8 x 4 MiB of L2 and the Infinity Cache; at 32 per CU the slots thrash L2: TCC hit rate 33 %), fills them.  The ticket
// balances the two populations whatever their relative speed (a static split loses 3-8 % to the slower side's tail).
template <int VARIANT>
__global__ __launch_bounds__(SNP_WAVE) void k_compress_win_q(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                            const u32* __restrict__ in_len, u32 nblocks,
                                                            u8* __restrict__ out, const u64* __restrict__ out_off,
                                                            u32* __restrict__ out_len, i32* __restrict__ status,
                                                            int emit_varint, u32* __restrict__ ticket)
{
    for (;;) {
        u32 b = 0;
        if (lane_id() == 0) b = atomicAdd(ticket, 1u);
        b = bcast_first(b);
        if (b >= nblocks) break;
        compress_win_fragment<VARIANT, 1, false>(in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, b, nullptr);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the next fragment reuses the LDS arrays)
    }
}
template <int VARIANT>
__global__ __launch_bounds__(SNP_WAVE) void k_compress_win_gq(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                             const u32* __restrict__ in_len, u32 nblocks,
                                                             u8* __restrict__ out, const u64* __restrict__ out_off,
                                                             u32* __restrict__ out_len, i32* __restrict__ status,
                                                             int emit_varint, u16* __restrict__ tables, u32* __restrict__ ticket)
{
    u16* const mine = tables + static_cast<size_t>(blockIdx.x) * 16384u;
    for (;;) {
        u32 b = 0;
        if (lane_id() == 0) b = atomicAdd(ticket, 1u);
        b = bcast_first(b);
        if (b >= nblocks) break;
        compress_win_fragment<VARIANT, 1, true>(in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, b, mine);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the next fragment reuses the slot and the LDS arrays)
    }
}

// test hook (include/snappier_hip_debug.h): FindMatchLength by the wave, as the kernel uses it (tests/test_gpu_parity.py runs the reference's KATs through it)
__global__ __launch_bounds__(SNP_WAVE) void k_debug_match_length(const u8* buf, u32 n, u32 p, u32 cand, u32 known, u32* out)
{
    const u32 r = wave_match_extend(buf, n, p, cand, known, lane_id());
    if (lane_id() == 0) *out = r;
}

}  // namespace

extern "C" int snp_debug_match_length(const u8* d_buf, u32 n, u32 p, u32 cand, u32 known, u32* d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(k_debug_match_length, dim3(1), dim3(SNP_WAVE), 0, stream, d_buf, n, p, cand, known, d_out);
    return static_cast<int>(hipGetLastError());
}

// The dual form: `lds_groups` persistent workgroups of the LDS form on `stream`, `slots` of the global-slot form on `side` (forked from and joined back
// into `stream` through the two events: the pattern stream capture understands), one ticket counter (zeroed here, on `stream`).
extern "C" hipError_t snp_launch_compress_win_dual(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                                   const u64* out_off, u32* out_len, i32* status, int variant, int emit_varint,
                                                   hipStream_t stream, hipStream_t side, hipEvent_t fork, hipEvent_t join,
                                                   u16* tables, u32 slots, u32 lds_groups, u32* ticket)
{
    if (nblocks == 0) return hipSuccess;
    hipError_t e = snp_zero_words_async(ticket, 1, stream);
    if (e != hipSuccess) return e;
    if ((e = hipEventRecord(fork, stream)) != hipSuccess || (e = hipStreamWaitEvent(side, fork, 0)) != hipSuccess) return e;
    const u32 g = nblocks < slots ? nblocks : slots;
    const u32 l = nblocks < lds_groups ? nblocks : lds_groups;
    // (leaving the last ~900 fragments to the faster LDS form, so that both populations run dry together, measured worse at five of six batch sizes:
    //  profiles/r06e_dual_reserve.txt)
    if (variant == SNP_HASH_CRC32C) {
        hipLaunchKernelGGL((k_compress_win_gq<SNP_HASH_CRC32C>), dim3(g), dim3(SNP_WAVE), 0, side, in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, tables, ticket);
        hipLaunchKernelGGL((k_compress_win_q<SNP_HASH_CRC32C>), dim3(l), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, ticket);
    } else {
        hipLaunchKernelGGL((k_compress_win_gq<SNP_HASH_MUL>), dim3(g), dim3(SNP_WAVE), 0, side, in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, tables, ticket);
        hipLaunchKernelGGL((k_compress_win_q<SNP_HASH_MUL>), dim3(l), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, ticket);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipEventRecord(join, side)) != hipSuccess) return e;
    return hipStreamWaitEvent(stream, join, 0);
}

#if SNP_W_PROF
extern "C" int snp_debug_read_wprof(unsigned long long* out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wprof), sizeof(g_wprof));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_wprof), z, sizeof(z));
    }
    return static_cast<int>(e);
}
#endif

extern "C" size_t snp_compress_win_table_bytes(u32 slots) { return static_cast<size_t>(slots) * 16384u * sizeof(u16); }

// tables == nullptr: the LDS-table kernel, one workgroup per fragment.  Otherwise `slots` wavefronts, each with a 32 KiB table slot in `tables`.
extern "C" hipError_t snp_launch_compress_win(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                              const u64* out_off, u32* out_len, i32* status, int variant,
                                              int emit_varint, int np, hipStream_t stream, u16* tables, u32 slots)
{
    if (nblocks == 0) return hipSuccess;
    if (tables && slots) {
        const u32 grid = nblocks < slots ? nblocks : slots;
        if (variant == SNP_HASH_CRC32C)
            hipLaunchKernelGGL((k_compress_win_g<SNP_HASH_CRC32C, 1>), dim3(grid), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, tables);
        else
            hipLaunchKernelGGL((k_compress_win_g<SNP_HASH_MUL, 1>), dim3(grid), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, out, out_off, out_len, status, emit_varint, tables);
        return hipGetLastError();
    }
#define SNP_LAUNCH_W(V, P)                                                                                            \
    hipLaunchKernelGGL((k_compress_win<V, P>), dim3(nblocks), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, \
                       out, out_off, out_len, status, emit_varint)
    if (variant == SNP_HASH_CRC32C) {
        if (np == 1) SNP_LAUNCH_W(SNP_HASH_CRC32C, 1); else SNP_LAUNCH_W(SNP_HASH_CRC32C, 2);
    } else {
        if (np == 1) SNP_LAUNCH_W(SNP_HASH_MUL, 1); else SNP_LAUNCH_W(SNP_HASH_MUL, 2);
    }
#undef SNP_LAUNCH_W
    return hipGetLastError();
}
