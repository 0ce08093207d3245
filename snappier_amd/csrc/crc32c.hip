// crc32c.hip -- wave-parallel CRC-32C (Castagnoli, reflected 0x82F63B78) for the framing format.
// Replaces Crc32CAlgorithm.Compute / ApplyMask (Snappier/Internal/Crc32CAlgorithm.cs:41-158), which uses the CPU's
// crc32 instruction or 16 x 256-entry tables; gfx950 has neither a CRC instruction nor carry-less multiply.
//
// One wavefront per byte range.  CRC is GF(2)-linear: absorbing dword w into state s is s' = X32(s ^ w), where Xk =
// "multiply by x^k mod P" (k reflected shift/xor steps).  So with dwords w_0..w_{N-1}
//     state = XOR_t  X_{32(N-t)}(w_t)                       (the 0xFFFFFFFF init is XORed into the first 4 bytes)
// The message is right-aligned on a 1024-byte grid by (virtual) leading zero bytes, which change nothing.  Lane l
// owns dwords 4l .. 4l+3 of every 1 KiB row (one coalesced 16-byte load per lane per row) and runs four Horner
// recurrences   acc_j = X8192(acc_j) ^ w_j.   X8192 is a constant 32x32 GF(2) map, applied sliced: round 3 by 8 bits (four
// 256-entry tables in LDS, 4 lookups per dword), round 4 by 11 + 11 + 10 bits (three tables, 20 KiB, 3 lookups per dword; a wavefront then
// takes four byte ranges in a row so that the table copy stays ~2 % of the traffic) with the rows read as non-temporal loads:
// 2.04 -> 1.85 ms per 10 GiB = 5.8 TB/s = 73 % of the HBM peak, 92 % of the 6.3 TB/s a plain copy reaches on this part
// (MI355X_MICROARCH guide).  The bit-by-bit map (96 VALU per dword) is the table-free variant, 1.7 TB/s.  The four accumulators of a
// lane fold with three X32, the 64 lanes with a 6-level butterfly (left half times X_{128*2^k}, xor with the partner lane), one X32 finishes.
#include "snp_device.h"

namespace {

constexpr u32 kPoly = 0x82F63B78u;   // Crc32CAlgorithm.cs:15

constexpr u32 xstep(u32 v, int k)
{
    for (int i = 0; i < k; ++i) v = (v >> 1) ^ ((v & 1u) ? kPoly : 0u);
    return v;
}

// Column table of the linear map X_K: col[b] = X_K(1 << b)
template <int K>
struct XMap {
    u32 col[32];
    constexpr XMap() : col{}
    {
        for (int b = 0; b < 32; ++b) col[b] = xstep(1u << b, K);
    }
};

template <int K>
inline constexpr XMap<K> kXMap{};

template <int K, int B>
__device__ __forceinline__ u32 xmul_bits(u32 v)
{
    if constexpr (B >= 32) return 0u;
    else {
        constexpr u32 col = kXMap<K>.col[B];
        // sign-extended 1-bit field: all-ones when bit B of v is set
        const u32 sel = static_cast<u32>(__builtin_amdgcn_sbfe(static_cast<int>(v), B, 1));
        return (sel & col) ^ xmul_bits<K, B + 1>(v);
    }
}
// X_K(v) for a compile-time K
template <int K>
__device__ __forceinline__ u32 xmul(u32 v) { return xmul_bits<K, 0>(v); }

// X8192 sliced by 8: kLut.t[k][b] = X8192(b << 8k)
struct CrcLut {
    u32 t[4][256];
    constexpr CrcLut() : t{}
    {
        for (int k = 0; k < 4; ++k)
            for (int b = 0; b < 256; ++b) {
                u32 v = 0;
                for (int bit = 0; bit < 8; ++bit)
                    if (b & (1 << bit)) v ^= kXMap<8192>.col[8 * k + bit];
                t[k][b] = v;
            }
    }
};
__device__ const CrcLut g_crc_lut{};

// X8192 sliced 11 + 11 + 10: three lookups per accumulator instead of four (the kernel is bound by LDS instruction throughput: a 64-lane
// ds_read_b32 is ~4.7 cycles of the CU's LDS pipeline with or without bank conflicts, profiles/r02n_microbench_lds_unaligned.jsonl), 20 KiB of LDS.
struct CrcLut11 {
    alignas(16) u32 t[5120];                                            // [0, 2048): bits 0-10; [2048, 4096): bits 11-21; [4096, 5120): bits 22-31 (copied 16 bytes at a time)
    constexpr CrcLut11() : t{}
    {
        for (int k = 0; k < 3; ++k) {
            const int width = k == 2 ? 10 : 11, base = 2048 * k, lo = 11 * k;
            for (int b = 0; b < (1 << width); ++b) {
                u32 v = 0;
                for (int bit = 0; bit < width; ++bit)
                    if (b & (1 << bit)) v ^= kXMap<8192>.col[lo + bit];
                t[base + b] = v;
            }
        }
    }
};
__device__ const CrcLut11 g_crc_lut11{};

__device__ __forceinline__ u32 xstep8(u32 v)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) v = (v >> 1) ^ ((v & 1u) ? kPoly : 0u);
    return v;
}

constexpr u32 kCrcWaves = 4;   // byte ranges per workgroup (they share the LDS table)

// MODE 1 (TABLE_FREE): the form BASELINE.json's north star names -- no table anywhere, X8192 applied bit by bit (32 x sbfe / and / xor
// per dword: VALU-bound, 1.7 TB/s; SNP_OPT_CRC_KERNEL selects it).  MODE 0: four 256-entry tables in LDS (4.5-5.3 TB/s).
// MODE 2: three tables of 2048 / 2048 / 1024 entries (11 + 11 + 10 bits), three lookups per dword; a wavefront then takes kPer11 consecutive
// byte ranges, so that the 20 KiB table is copied once per 16 ranges.
#ifndef SNP_CRC_NT
#define SNP_CRC_NT 1      // the rows as non-temporal loads: 1.85 vs 1.94 ms per 10 GiB (profiles/r04af_crc_variants.txt)
#endif
#ifndef SNP_CRC_DEPTH
#define SNP_CRC_DEPTH 2    // rows in flight per wavefront (1, 2, 4, 8: no difference)
#endif
#ifndef SNP_CRC_PER11
#define SNP_CRC_PER11 4
#endif
#ifndef SNP_CRC_WAVES11
#define SNP_CRC_WAVES11 4
#endif
constexpr u32 kPer11 = SNP_CRC_PER11, kWaves11 = SNP_CRC_WAVES11;
template <int MODE>
__global__ __launch_bounds__(SNP_WAVE * (MODE == 2 ? kWaves11 : kCrcWaves)) void k_crc32c(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                                const u32* __restrict__ in_len, u32 nblocks, int masked,
                                                                u32* __restrict__ out_crc, const u32* __restrict__ expect,
                                                                i32* __restrict__ status)
{
    constexpr bool TABLE_FREE = MODE == 1;
    constexpr u32 kPer = MODE == 2 ? kPer11 : 1u, kWaves = MODE == 2 ? kWaves11 : kCrcWaves;
    __shared__ __attribute__((aligned(16))) u32 T[MODE == 2 ? 5120 : TABLE_FREE ? 1 : 1024];
    if constexpr (MODE == 0) {
        for (u32 e = threadIdx.x; e < 1024; e += SNP_WAVE * kWaves) T[e] = g_crc_lut.t[e >> 8][e & 255u];
        __syncthreads();
    } else if constexpr (MODE == 2) {
        for (u32 e = threadIdx.x * 4; e < 5120; e += SNP_WAVE * kWaves * 4)
            *reinterpret_cast<uint4*>(&T[e]) = *reinterpret_cast<const uint4*>(&g_crc_lut11.t[e]);
        __syncthreads();
    }
    const u32 lane = lane_id();
    for (u32 j = 0; j < kPer; ++j) {
    const u32 b = (blockIdx.x * kWaves + (threadIdx.x >> 6)) * kPer + j;
    if (b >= nblocks) return;
    const u8* src = in + in_off[b];
    const u32 n = bcast_first(in_len[b]);

    u32 crc;
    if (n < 4) {                                                        // tiny inputs: plain byte steps (uniform)
        u32 s = 0xffffffffu;
        for (u32 i = 0; i < n; ++i) s = xstep8(s ^ src[i]);
        crc = s ^ 0xffffffffu;
    } else {
        const u32 padb = (1024u - (n & 1023u)) & 1023u;                 // virtual leading zero bytes
        const u32 rows = (n + padb) >> 10;
        u32 acc[4] = {0, 0, 0, 0};
        // this lane's 16 bytes of row i start at src + i*1024 + lane*16 - padb
        auto load_row = [&](u32 i, u32 (&w)[4]) {
            const i32 rbyte = static_cast<i32>(i * 1024u + lane * 16u) - static_cast<i32>(padb);
            if (rbyte >= 4) {
#if SNP_CRC_NT
                typedef u32 u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
                const u32x4_u q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u*>(src + rbyte));   // read once, never again: do not keep it in L2
                w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
#else
                const snp_u128_unaligned q = *reinterpret_cast<const snp_u128_unaligned*>(src + rbyte);
                w[0] = q.v[0]; w[1] = q.v[1]; w[2] = q.v[2]; w[3] = q.v[3];
#endif
            } else {                                                    // first row only: leading pad and the init xor
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    w[d] = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const i32 bi = rbyte + 4 * d + j;
                        if (bi >= 0) {
                            u32 byte = src[bi];
                            if (bi < 4) byte ^= 0xffu;                  // init 0xFFFFFFFF == first four bytes inverted
                            w[d] |= byte << (8 * j);
                        }
                    }
                }
            }
        };
        // kDepth rows in flight per wavefront (one was not enough to keep HBM busy: 32 wavefronts x 256 CUs x 1 KiB = 8 MB in flight against the
        // ~16 MB that 8 TB/s x 2 us ask for)
        constexpr u32 kDepth = SNP_CRC_DEPTH;
        u32 buf[kDepth][4];
#pragma unroll
        for (u32 k = 0; k < kDepth; ++k) {
            buf[k][0] = buf[k][1] = buf[k][2] = buf[k][3] = 0;
            if (k < rows) load_row(k, buf[k]);
        }
        for (u32 i = 0; i < rows; i += kDepth) {
#pragma unroll
            for (u32 k = 0; k < kDepth; ++k) {
                if (i + k < rows) {                                     // (wave-uniform)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const u32 a = acc[d];
                        if constexpr (TABLE_FREE) acc[d] = xmul<8192>(a) ^ buf[k][d];
                        else if constexpr (MODE == 2) acc[d] = T[a & 2047u] ^ T[2048u + ((a >> 11) & 2047u)] ^ T[4096u + (a >> 22)] ^ buf[k][d];
                        else acc[d] = T[a & 255u] ^ T[256u + ((a >> 8) & 255u)] ^ T[512u + ((a >> 16) & 255u)] ^ T[768u + (a >> 24)] ^ buf[k][d];
                    }
                    if (i + k + kDepth < rows) load_row(i + k + kDepth, buf[k]);   // this slot's next row goes out as soon as the slot is free
                }
            }
        }
        // fold: total = XOR_{l,d} X_{32(3-d) + 128(63-l)}(acc_{l,d}), then one X32
        u32 v = xmul<32>(xmul<32>(xmul<32>(acc[0]) ^ acc[1]) ^ acc[2]) ^ acc[3];
        if (!(lane & 1)) v = xmul<128>(v);
        v ^= __shfl_xor(v, 1, 64);
        if (!(lane & 2)) v = xmul<256>(v);
        v ^= __shfl_xor(v, 2, 64);
        if (!(lane & 4)) v = xmul<512>(v);
        v ^= __shfl_xor(v, 4, 64);
        if (!(lane & 8)) v = xmul<1024>(v);
        v ^= __shfl_xor(v, 8, 64);
        if (!(lane & 16)) v = xmul<2048>(v);
        v ^= __shfl_xor(v, 16, 64);
        if (!(lane & 32)) v = xmul<4096>(v);
        v ^= __shfl_xor(v, 32, 64);
        crc = xmul<32>(v) ^ 0xffffffffu;                                // Crc32CAlgorithm.cs:48,153 final xor
    }
    if (masked & 1) crc = crc32c_mask(crc);                             // ApplyMask  :156-158
    if (lane == 0) {
        if (out_crc) out_crc[b] = crc;
        // framing verify: "Chunk CRC mismatch."  SnappyStreamDecompressor.cs:127-131,170-174
        if (expect && status && status[b] == SNP_OK && expect[b] != crc) status[b] = SNP_ERR_CRC_MISMATCH;
    }
    }
}

}  // namespace

extern "C" hipError_t snp_launch_crc32c(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, int masked,
                                        u32* out_crc, const u32* expect, i32* status, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    // masked: bit 0 = apply the framing mask, bit 1 = the table-free kernel, bit 2 = the 8-bit-sliced tables (round 3's default)
    // (the three-table kernel amortises its 20 KiB table copy over kPer11 byte ranges per wavefront: a batch too small to fill the chip that way --
    //  a few hundred frame chunks -- takes the 8-bit tables, one range per wavefront: ADVICE r4)
    if (!(masked & 6) && nblocks < 16u * 256u * kWaves11) masked |= 4;
    const u32 waves = (masked & 6) ? kCrcWaves : kWaves11, per_wg = waves * ((masked & 6) ? 1u : kPer11);
    const dim3 grid((nblocks + per_wg - 1) / per_wg), block(SNP_WAVE * waves);
    if (masked & 2)
        hipLaunchKernelGGL(k_crc32c<1>, grid, block, 0, stream, in, in_off, in_len, nblocks, masked, out_crc, expect, status);
    else if (masked & 4)
        hipLaunchKernelGGL(k_crc32c<0>, grid, block, 0, stream, in, in_off, in_len, nblocks, masked, out_crc, expect, status);
    else
        hipLaunchKernelGGL(k_crc32c<2>, grid, block, 0, stream, in, in_off, in_len, nblocks, masked, out_crc, expect, status);
    return hipGetLastError();
}
