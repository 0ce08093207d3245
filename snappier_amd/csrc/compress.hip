// compress.hip -- Snappy fragment compression, one <= 64 KiB fragment per wavefront (gfx950), bit-exact with
// SnappyCompressor.CompressFragment (Snappier/Internal/SnappyCompressor.cs:174-415) for both TableEntry hashes
// (HashTable.cs:91-126).
//
// The reference parse is a serial greedy chain (probe -> table lookup -> candidate compare -> insert).  A wavefront
// runs it in ROUNDS of up to 64 speculative probes and then proves which prefix of the round the serial algorithm
// would really have executed (tests/wave_model.py is the executable model of exactly this file, checked against
// the oracle on CPU):
//   kind A (scan)        lane j   probes p = start + D[kbase+j]
//   kind B (after copy)  lane 0   inserts ip-1 only (:393-394); lane 1 probes ip (a hit = back-to-back copy,
//                        :395-398); lane 2+j probes ip+1+D[j] (the next outer iteration, :198-341)
//   D[] = probe offsets of the skip heuristic (:227,319-320): D[0]=0, D[k+1]=D[k]+1+(D[k]>>5); probe k is legal
//   iff start + D[k+1] <= limit (:323).  The unrolled 16-probe section (:230-313) follows the same sequence.
//   1. every lane loads its 4 bytes, hashes, gathers the pre-round candidate from the LDS table, loads the
//      candidate's 4 bytes; first0 = first lane with a (stale-table) hit or an illegal probe;
//   2. lanes 0..first0 ("R") publish their position to the table and read it back: if every lane sees its own
//      value the buckets are pairwise distinct and the stale view was exact -> done (fast path, >99 % on text);
//   3. otherwise a scalar loop over R redoes the lookups in registers (latest earlier lane in the same bucket is
//      the true candidate), the table is restored and R's survivors republish with max-position-wins.
//   Match extension is a 64-lane byte compare + ballot + ctz; literals and copy tags are emitted lane-parallel.
// LDS: the 16384 x u16 table = 32 KiB per wavefront, exactly 5 wavefronts per 160 KiB CU; no other LDS is used
// (the probe-offset table lives in four VGPRs).  The hash for SNP_HASH_CRC32C is table-free: the CRC step is
// GF(2)-linear, so bit i of it is parity(x & ROW[i]) (14 AND+popcount pairs; gfx950 has no CRC instruction).
#include <cstdlib>

#include "snp_device.h"

namespace {

// ---- compile-time tables ---------------------------------------------------------------------------------
struct ProbeTable {
    u16 d[640];
    constexpr ProbeTable() : d{}
    {
        u32 v = 0;
        for (int i = 0; i < 640; ++i) {
            d[i] = static_cast<u16>(v > 0xffffu ? 0xffffu : v);      // saturate: anything >= 65536 is illegal anyway
            v = v + 1 + (v >> 5);
        }
    }
};
__device__ const ProbeTable g_probe{};

constexpr u32 crc_step32(u32 x)
{
    for (int k = 0; k < 32; ++k) x = (x >> 1) ^ ((x & 1u) ? 0x82F63B78u : 0u);
    return x;
}
// ROW[i] such that bit i of crc_step32(x) == parity(x & ROW[i])
constexpr u32 crc_row(int i)
{
    u32 m = 0;
    for (int b = 0; b < 32; ++b)
        if ((crc_step32(1u << b) >> i) & 1u) m |= 1u << b;
    return m;
}
template <int I>
struct CrcRow { static constexpr u32 value = crc_row(I); };

template <int I>
__device__ __forceinline__ u32 crc_bits(u32 x)
{
    if constexpr (I > 14) return 0u;
    else return ((__popc(x & CrcRow<I>::value) & 1u) << I) | crc_bits<I + 1>(x);
}

// HashTable.TableEntry (HashTable.cs:91-126) -> entry index (byte offset / 2)
template <int VARIANT>
__device__ __forceinline__ u32 table_index(u32 bytes, u32 mask)
{
    u32 hash;
    if constexpr (VARIANT == SNP_HASH_CRC32C) hash = crc_bits<1>(bytes ^ mask);   // Sse42.Crc32(bytes, mask), bits 1..14
    else hash = (0x1e35a7bdu * bytes) >> 17;                                       // :121-122
    return (hash & mask) >> 1;
}

__device__ __forceinline__ u32 log2_floor(u32 v) { return 31u - __clz(v); }
__device__ __forceinline__ void lanes_sync() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ u32 bperm(u32 src_lane, u32 v)
{
    return static_cast<u32>(__builtin_amdgcn_ds_bpermute(static_cast<int>(src_lane << 2), static_cast<int>(v)));
}

// tuning knobs (compile-time; scripts/ablate_compress.sh builds variants to measure each one)
#ifndef SNP_C_NARROW
#define SNP_C_NARROW 16
#endif
#ifndef SNP_C_SMALLR
#define SNP_C_SMALLR 0
#endif
constexpr u32 kNarrow = SNP_C_NARROW;   // lanes speculated in the first round after a match (most hits are within a few probes)
constexpr u32 kSmallR = SNP_C_SMALLR;   // up to this many real lanes, bucket conflicts are found by comparing hashes in registers

// EmitLiteral (SnappyCompressor.cs:418-464): tag (+ length bytes) by lane 0, body lane-parallel.  Returns new op.
__device__ __forceinline__ u32 emit_literal(u8* dst, u32 op, const u8* src, u32 s, u32 len, u32 lane)
{
    const u32 k = len - 1;
    u32 hdr;
    if (k < 60) {
        if (lane == 0) dst[op] = static_cast<u8>(k << 2);
        hdr = 1;
    } else {
        const u32 count = (log2_floor(k) >> 3) + 1;                     // :447
        if (lane == 0) dst[op] = static_cast<u8>((59 + count) << 2);    // :451
        if (lane >= 1 && lane <= count) dst[op + lane] = static_cast<u8>(k >> (8 * (lane - 1)));
        hdr = 1 + count;
    }
    if (len <= 64) {
        if (lane < len) dst[op + hdr + lane] = src[s + lane];
    } else {
        wave_copy(dst + op + hdr, src + s, len, lane);
    }
    return op + hdr + len;
}

// EmitCopyLenLessThan12 / EmitCopyLenGreaterThanOrEqualTo12 (SnappyCompressor.cs:467-543) in closed form:
// q tags of length 64, optionally one of 60, then the final 4..64-byte tag.  Returns new op.
__device__ __forceinline__ u32 emit_copy(u8* dst, u32 op, u32 off, u32 len, u32 lane)
{
    u32 q = len >= 68 ? (len - 4) >> 6 : 0;
    u32 r = len - (q << 6);
    for (u32 t = lane; t < q; t += 64) {
        u8* o = dst + op + 3 * t;
        o[0] = static_cast<u8>(2u | (63u << 2));
        o[1] = static_cast<u8>(off);
        o[2] = static_cast<u8>(off >> 8);
    }
    op += 3 * q;
    u64 tok = 0;
    u32 tl = 0;
    if (r > 64) {                                                       // :531-534
        tok = (2u | (59u << 2)) | (static_cast<u64>(off) << 8);
        tl = 3;
        r -= 60;
    }
    if (r < 12 && off < 2048) {                                         // copy-1  :476-489
        const u64 t2 = (1u | ((r - 4) << 2) | ((off >> 8) << 5)) | (static_cast<u64>(off & 0xffu) << 8);
        tok |= t2 << (8 * tl);
        tl += 2;
    } else {                                                            // copy-2  :478,502
        const u64 t3 = (2u | ((r - 1) << 2)) | (static_cast<u64>(off) << 8);
        tok |= t3 << (8 * tl);
        tl += 3;
    }
    if (lane < tl) dst[op + lane] = static_cast<u8>(tok >> (8 * lane));
    return op + tl;
}

// ---- optional phase timing (build with -DSNP_C_PROF=1; scripts/ablate_compress.sh prof) -------------------------
#ifndef SNP_C_PROF
#define SNP_C_PROF 0
#endif
#if SNP_C_PROF
__device__ unsigned long long g_prof[16];
#define PROF_DECL u64 prof_t = __builtin_readcyclecounter(); u64 prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_MARK(k)                                                              \
    do {                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               \
        const u64 now_ = __builtin_readcyclecounter();                            \
        prof_acc[k] += now_ - prof_t;                                             \
        prof_t = now_;                                                            \
    } while (0)
#define PROF_FLUSH                                                                \
    if (lane == 0)                                                                \
        for (int k_ = 0; k_ < 10; ++k_) atomicAdd(&g_prof[k_], prof_acc[k_]);
#else
#define PROF_DECL
#define PROF_MARK(k)
#define PROF_FLUSH
#endif

// STAGED = true first copies the fragment into LDS (64 KiB input + 32 KiB table = one fragment per CU): every probe,
// candidate and extension load is then an LDS access instead of a global round trip.  For batches that cannot fill
// the chip anyway (<= one fragment per CU) this is the fast layout; larger batches want five fragments per CU.
template <int VARIANT, bool STAGED>
__global__ __launch_bounds__(SNP_WAVE) void k_compress(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                      const u32* __restrict__ in_len, u32 nblocks,
                                                      u8* __restrict__ out, const u64* __restrict__ out_off,
                                                      u32* __restrict__ out_len, i32* __restrict__ status,
                                                      int emit_varint)
{
    // Lanes communicate through the table (publish / read back).  The compiler must not forward a lane's own store to
    // its later load of the same entry -- another lane may have overwritten it -- so those hand-offs are separated
    // by lanes_sync() (a compiler-only memory barrier; DS operations of one wave execute in order anyway).
    // (Not `volatile`: volatile LDS accesses are left in the generic address space and compile to flat_load/store.)
    __shared__ u16 table[16384];                                        // HashTable.cs:17-18
    const u32 b = blockIdx.x;
    if (b >= nblocks) return;
    const u32 lane = lane_id();
    const u8* src = in + in_off[b];
    const u32 n = bcast_first(in_len[b]);
    u8* dst = out + out_off[b];

    if (n > SNP_BLOCK_SIZE) {
        if (lane == 0) { out_len[b] = 0; status[b] = SNP_ERR_BAD_ARG; }
        return;
    }
    __shared__ u8 s_in[STAGED ? SNP_BLOCK_SIZE : 16];
    if (STAGED) {
        wave_copy(s_in, src, n, lane);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // staged bytes are read by other lanes
        src = s_in;
    }

    u32 op = 0;
    if (emit_varint) {                                                  // VarIntEncoding.TryWrite  VarIntEncoding.Write.cs:5-79
        const u32 hb = n < (1u << 7) ? 1 : n < (1u << 14) ? 2 : 3;      // n <= 65536 < 2^21
        if (lane < hb) dst[lane] = static_cast<u8>((n >> (7 * lane)) | (lane + 1 < hb ? 0x80u : 0u));
        op = hb;
    }

    u32 next_emit = 0;
    if (n >= 15) {                                                      // Constants.InputMarginBytes  :190
        // HashTable.CalculateTableSize + Clear  HashTable.cs:52,57-71
        const u32 tsize = n > 16384 ? 16384u : n < 256 ? 256u : (2u << log2_floor(n - 1));
        const u32 mask = 2 * (tsize - 1);                               // :181
        for (u32 i = lane * 8; i < tsize; i += 64 * 8) *reinterpret_cast<uint4*>(&table[i]) = make_uint4(0, 0, 0, 0);
        lanes_sync();

        const u32 limit = n - 15;                                       // :192
        // Probe offsets for the first round of a scan, in registers.  Computed, not loaded: a value loaded before the
        // loop makes the compiler re-wait on vmcnt at its first use in every iteration, and because vmcnt retires in
        // order that wait also drains the previous round's stores.
        u32 dA0 = 0, dA1 = 0, dB0 = 0, dB1 = 1;
        {
            u32 v = 0;
            for (u32 k = 0; k < 66; ++k) {
                if (lane == k) dA0 = v;
                if (lane + 1 == k) dA1 = v;
                if (lane == k + 2) dB0 = v;
                if (lane == k + 1 && lane >= 2) dB1 = v;
                v += 1 + (v >> 5);
            }
        }

        bool kind_b = false;
        u32 ip = 0, start = 1, kbase = 0;
        PROF_DECL
        u32 width = kNarrow;      // lanes speculated this round: kNarrow after a match, 64 once a round found nothing
        // Input forwarding: the match-extension step of the previous round loaded a dword per lane from consecutive
        // positions around the end of the match -- exactly where this round probes.  fwd[l] = ld32(src + fwd_base + l).
        u32 fwd = 0, fwd_base = 0;
        bool fwd_ok = false;
        for (;;) {
            // The round state is wave-uniform by construction; say so, or the compiler keeps it in VGPRs and turns
            // every state update into exec-masked vector code.
            ip = bcast_first(ip); start = bcast_first(start); kbase = bcast_first(kbase); width = bcast_first(width);
            op = bcast_first(op); next_emit = bcast_first(next_emit); fwd_base = bcast_first(fwd_base);
            kind_b = bcast_first(kind_b ? 1u : 0u) != 0;
            fwd_ok = bcast_first(fwd_ok ? 1u : 0u) != 0;
            // ---- 1. positions, legality ------------------------------------------------------------------
            const bool first_b = kind_b && kbase == 0;
            u32 p, pnext;
            bool legal, probing;
            if (first_b) {
                p = lane == 0 ? ip - 1 : lane == 1 ? ip : start + dB0;
                pnext = start + dB1;
                legal = lane < 2 || pnext <= limit;
                probing = lane >= 1;
            } else {
                const u32 d0 = kbase == 0 ? dA0 : g_probe.d[kbase + lane];
                const u32 d1 = kbase == 0 ? dA1 : g_probe.d[kbase + lane + 1];
                p = start + d0;
                pnext = start + d1;
                legal = pnext <= limit;
                probing = true;
            }
            lanes_sync();                                               // table entries may have been rewritten by other lanes
            const bool spec = lane < width;
            const bool valid = legal && spec;
            const u64 wmask = lanes_below(width);
            const u64 lmask = ballot64(legal);
            // ---- 2. speculative probe of the lanes against the pre-round table --------------------------------------
            u32 d;
            if (first_b && fwd_ok) {
                // lanes 0..width-1 sit at consecutive positions ip-1, ip, ip+1, ..: a lane shift of the forwarded dwords
                d = bperm(lane + (ip - 1 - fwd_base), fwd);
            } else {
                d = valid ? ld32u(src + p) : 0u;
            }
            PROF_MARK(0);                                               // positions + input load
            const u32 h = table_index<VARIANT>(d, mask);
            PROF_MARK(1);                                               // hash
            const u32 c = table[h];
            PROF_MARK(2);                                               // table gather (LDS)
            const u32 e = (valid && probing) ? ld32u(src + c) : ~d;
            PROF_MARK(3);                                               // candidate gather (global)
            const bool stale = valid && probing && e == d;
            const u64 smask = ballot64(stale);
            const u64 stop = (smask | ~lmask) & wmask;
            const u32 first0 = stop ? static_cast<u32>(__builtin_ctzll(stop)) : width;
            const bool terminated = first0 < width && !((lmask >> first0) & 1ull);
            const bool in_r = valid && (lane < first0 || (lane == first0 && !terminated));
            const u64 rmask = ballot64(in_r);
            // ---- 3. are R's buckets pairwise distinct?  few lanes: compare hashes in registers; many: publish to
            //         the table and read back -----------------------------------------------------------------------
            const bool published = __builtin_popcountll(rmask) > static_cast<int>(kSmallR);
            bool conflict = false;
            if (published) {
                if (in_r) table[h] = static_cast<u16>(p);
                lanes_sync();
                const u32 rb = table[h];
                conflict = ballot64(in_r && rb != p) != 0ull;
            } else {
                u64 it = rmask & (rmask - 1);                           // every real lane but the first
                while (it) {
                    const u32 j = static_cast<u32>(__builtin_ctzll(it));
                    it &= it - 1;
                    if (ballot64(in_r && h == read_lane(h, j)) & lanes_below(j)) { conflict = true; break; }
                }
                if (!conflict && in_r) table[h] = static_cast<u16>(p);  // distinct buckets: plain stores
            }
            conflict = bcast_first(conflict ? 1u : 0u) != 0;
            int m = (first0 < width && !terminated) ? static_cast<int>(first0) : -1;
            u32 cand = m >= 0 ? read_lane(c, static_cast<u32>(m)) : 0u;
            if (conflict) {
                // ---- 4. exact resolution in registers ------------------------------------------------------
                m = -1;
                u64 it = rmask & ballot64(probing);
                while (it) {
                    const u32 j = static_cast<u32>(__builtin_ctzll(it));
                    it &= it - 1;
                    const u32 hj = read_lane(h, j);
                    const u64 same = ballot64(in_r && h == hj) & lanes_below(j);
                    if (same) {
                        const u32 a = 63u - static_cast<u32>(__builtin_clzll(same));
                        if (read_lane(d, a) == read_lane(d, j)) { m = static_cast<int>(j); cand = read_lane(p, a); break; }
                    } else if ((smask >> j) & 1ull) {
                        m = static_cast<int>(j);
                        cand = read_lane(c, j);
                        break;
                    }
                }
                // ---- 5. table fix-up: restore what was published, survivors republish, larger position wins ----
                if (published && in_r) table[h] = static_cast<u16>(c);
                const bool keep = in_r && (m < 0 || lane <= static_cast<u32>(m));
                bool active = keep;
                lanes_sync();
                while (ballot64(active)) {
                    if (active) table[h] = static_cast<u16>(p);
                    lanes_sync();
                    active = keep && static_cast<u32>(table[h]) < p;
                }
            }
            PROF_MARK(4);                                               // decision, conflict check, table update
            m = static_cast<int>(bcast_first(static_cast<u32>(m)));
            cand = bcast_first(cand);
            if (m < 0) {
                if (terminated) break;                                  // :323-327 -> emit_remainder from next_emit
                const u32 done = first0 == width ? width : first0 + 1;  // lanes really processed this round
                if (first_b) {
                    next_emit = ip;                                     // post-copy probe missed: new outer iteration
                    kbase = done > 2 ? done - 2 : 0;
                    kind_b = false;
                } else {
                    kbase += done;
                }
                width = 64;
                continue;
            }
            // ---- 6. literal ------------------------------------------------------------------------------------
            const u32 pm = read_lane(p, static_cast<u32>(m));
            if (pm > next_emit) {                                       // :347
                const u32 llen = pm - next_emit;
                if (first_b && llen < 60) {
                    // the literal bytes src[ip .. pm) are the low bytes of lanes 1 .. m-1 of this round's probes
                    if (lane == 0) dst[op] = static_cast<u8>((llen - 1) << 2);
                    if (lane >= 1 && lane <= llen) dst[op + lane] = static_cast<u8>(d);
                    op += 1 + llen;
                } else {
                    op = emit_literal(dst, op, src, next_emit, llen, lane);
                }
            }
            PROF_MARK(5);                                               // literal
            // ---- 7. match extension (FindMatchLength  :562-688), a dword per lane so the loads can be forwarded --------
            // Lane l looks at position eb + l; eb starts one byte before the first unknown byte, so lane 0 always
            // compares a byte already known to match and the first difference is at t >= 1.
            u32 eb = pm + 3;
            u32 x;
            u32 t;
            for (;;) {
                const u32 pos = eb + lane;
                u32 y;
                if (pos + 4 <= n) {
                    x = ld32u(src + pos);
                    y = ld32u(src + (pos - (pm - cand)));
                } else {                                                // the last three bytes of the fragment
                    x = 0;
                    y = 0;
                    for (u32 k = 0; k < 3; ++k)
                        if (pos + k < n) {
                            x |= static_cast<u32>(src[pos + k]) << (8 * k);
                            y |= static_cast<u32>(src[pos - (pm - cand) + k]) << (8 * k);
                        }
                }
                const bool same = pos < n && ((x ^ y) & 0xffu) == 0;
                const u64 diff = ballot64(!same);
                if (diff) { t = static_cast<u32>(__builtin_ctzll(diff)); break; }
                eb += 63;                                               // lane 63 matched: it becomes the next lane 0
            }
            const u32 matched = eb + t - pm;
            PROF_MARK(6);                                               // match extension
            op = emit_copy(dst, op, pm - cand, matched, lane);          // :371-379
            PROF_MARK(7);                                               // copy tags
            ip = pm + matched;
            next_emit = ip;
            if (ip >= limit) break;                                     // :381-384
            kind_b = true;
            kbase = 0;
            start = ip + 1;
            width = kNarrow;
            // forward the extension dwords to the next round: it needs lanes t-1 .. t-1+width-1 of x
            fwd = x;
            fwd_base = eb;
            fwd_ok = (t - 1 + kNarrow <= 64) && kNarrow <= 34;
        }
        PROF_FLUSH
    }
    if (next_emit < n) op = emit_literal(dst, op, src, next_emit, n - next_emit, lane);   // emit_remainder  :406-411

    if (lane == 0) {
        out_len[b] = op;
        status[b] = SNP_OK;
    }
}

}  // namespace

#if SNP_C_PROF
extern "C" int snp_debug_read_prof(unsigned long long* out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_prof), sizeof(g_prof));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
    }
    return static_cast<int>(e);
}
#endif

extern "C" hipError_t snp_launch_compress(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                          const u64* out_off, u32* out_len, i32* status, int variant,
                                          int emit_varint, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    // input staged in LDS while every fragment can have a CU to itself (SNAPPIER_HIP_STAGED=0/1 pins it)
    const char* env = getenv("SNAPPIER_HIP_STAGED");
    const bool staged = env ? env[0] == '1' : nblocks <= 256;
#define SNP_LAUNCH_C(V, S)                                                                                          \
    hipLaunchKernelGGL((k_compress<V, S>), dim3(nblocks), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, out, \
                       out_off, out_len, status, emit_varint)
    if (variant == SNP_HASH_CRC32C) { if (staged) SNP_LAUNCH_C(SNP_HASH_CRC32C, true); else SNP_LAUNCH_C(SNP_HASH_CRC32C, false); }
    else { if (staged) SNP_LAUNCH_C(SNP_HASH_MUL, true); else SNP_LAUNCH_C(SNP_HASH_MUL, false); }
#undef SNP_LAUNCH_C
    return hipGetLastError();
}
