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

template <int VARIANT>
__global__ __launch_bounds__(SNP_WAVE) void k_compress(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                      const u32* __restrict__ in_len, u32 nblocks,
                                                      u8* __restrict__ out, const u64* __restrict__ out_off,
                                                      u32* __restrict__ out_len, i32* __restrict__ status,
                                                      int emit_varint)
{
    __shared__ u16 table_mem[16384];                                    // HashTable.cs:17-18
    // Lanes communicate through the table (publish / read back), so every access is volatile: the compiler must not
    // forward a lane's own store to its later load of the same entry -- another lane may have overwritten it.
    volatile u16* table = table_mem;
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
        for (u32 i = lane * 8; i < tsize; i += 64 * 8) *reinterpret_cast<uint4*>(&table_mem[i]) = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // order the clear before the volatile traffic below

        const u32 limit = n - 15;                                       // :192
        // probe offsets for the first round of a scan, in registers
        const u32 dA0 = g_probe.d[lane], dA1 = g_probe.d[lane + 1];
        const u32 dB0 = g_probe.d[lane >= 2 ? lane - 2 : 0], dB1 = g_probe.d[lane >= 2 ? lane - 1 : 1];

        bool kind_b = false;
        u32 ip = 0, start = 1, kbase = 0;
        for (;;) {
            // ---- 1. positions, legality ------------------------------------------------------------------
            const bool first_b = kind_b && kbase == 0;
            u32 p, pnext;
            bool valid, probing;
            if (first_b) {
                p = lane == 0 ? ip - 1 : lane == 1 ? ip : start + dB0;
                pnext = start + dB1;
                valid = lane < 2 || pnext <= limit;
                probing = lane >= 1;
            } else {
                const u32 d0 = kbase == 0 ? dA0 : g_probe.d[kbase + lane];
                const u32 d1 = kbase == 0 ? dA1 : g_probe.d[kbase + lane + 1];
                p = start + d0;
                pnext = start + d1;
                valid = pnext <= limit;
                probing = true;
            }
            const u64 vmask = ballot64(valid);
            // ---- 2. speculative probe of all lanes against the pre-round table --------------------------------
            const u32 d = valid ? ld32u(src + p) : 0u;
            const u32 h = table_index<VARIANT>(d, mask);
            const u32 c = table[h];
            const u32 e = ld32u(src + c);
            const bool stale = valid && probing && e == d;
            const u64 smask = ballot64(stale);
            const u64 stop = smask | ~vmask;
            const u32 first0 = stop ? static_cast<u32>(__builtin_ctzll(stop)) : 64u;
            const bool terminated = first0 < 64 && !((vmask >> first0) & 1ull);
            const bool in_r = valid && (lane < first0 || (lane == first0 && !terminated));
            // ---- 3. publish + read back: are R's buckets pairwise distinct? -----------------------------------
            if (in_r) table[h] = static_cast<u16>(p);
            const u32 rb = table[h];
            const bool conflict = ballot64(in_r && rb != p) != 0ull;
            int m = (first0 < 64 && !terminated) ? static_cast<int>(first0) : -1;
            u32 cand = m >= 0 ? read_lane(c, static_cast<u32>(m)) : 0u;
            if (conflict) {
                // ---- 4. exact resolution in registers ------------------------------------------------------
                m = -1;
                const u64 rmask = ballot64(in_r);
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
                // ---- 5. table fix-up: restore, then survivors republish, larger position wins ---------------
                if (in_r) table[h] = static_cast<u16>(c);
                const bool keep = in_r && (m < 0 || lane <= static_cast<u32>(m));
                bool active = keep;
                while (ballot64(active)) {
                    if (active) table[h] = static_cast<u16>(p);
                    active = keep && static_cast<u32>(table[h]) < p;
                }
            }
            if (m < 0) {
                if (terminated) break;                                  // :323-327 -> emit_remainder from next_emit
                const u32 done = first0 == 64 ? 64u : first0 + 1;       // lanes really processed this round
                if (first_b) {
                    next_emit = ip;                                     // post-copy probe missed: new outer iteration
                    kbase = done > 2 ? done - 2 : 0;
                    kind_b = false;
                } else {
                    kbase += done;
                }
                continue;
            }
            // ---- 6. literal, match extension, copy ------------------------------------------------------------
            const u32 pm = read_lane(p, static_cast<u32>(m));
            if (pm > next_emit) op = emit_literal(dst, op, src, next_emit, pm - next_emit, lane);   // :347
            u32 matched = 4;                                            // FindMatchLength  :562-688
            for (;;) {
                const u32 pos = pm + matched + lane;
                const bool same = pos < n && src[cand + matched + lane] == src[pos];
                const u64 diff = ballot64(!same);
                if (diff) { matched += static_cast<u32>(__builtin_ctzll(diff)); break; }
                matched += 64;
            }
            op = emit_copy(dst, op, pm - cand, matched, lane);          // :371-379
            ip = pm + matched;
            next_emit = ip;
            if (ip >= limit) break;                                     // :381-384
            kind_b = true;
            kbase = 0;
            start = ip + 1;
        }
    }
    if (next_emit < n) op = emit_literal(dst, op, src, next_emit, n - next_emit, lane);   // emit_remainder  :406-411

    if (lane == 0) {
        out_len[b] = op;
        status[b] = SNP_OK;
    }
}

}  // namespace

extern "C" hipError_t snp_launch_compress(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                          const u64* out_off, u32* out_len, i32* status, int variant,
                                          int emit_varint, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if (variant == SNP_HASH_CRC32C)
        hipLaunchKernelGGL(k_compress<SNP_HASH_CRC32C>, dim3(nblocks), dim3(SNP_WAVE), 0, stream, in, in_off, in_len,
                           nblocks, out, out_off, out_len, status, emit_varint);
    else
        hipLaunchKernelGGL(k_compress<SNP_HASH_MUL>, dim3(nblocks), dim3(SNP_WAVE), 0, stream, in, in_off, in_len,
                           nblocks, out, out_off, out_len, status, emit_varint);
    return hipGetLastError();
}
