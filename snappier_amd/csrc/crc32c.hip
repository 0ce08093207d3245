// crc32c.hip -- table-free, wave-parallel CRC-32C (Castagnoli, reflected 0x82F63B78) for the framing format.
// Replaces Crc32CAlgorithm.Compute / ApplyMask (Snappier/Internal/Crc32CAlgorithm.cs:41-158), which uses the CPU's
// crc32 instruction or 16 x 256-entry tables; gfx950 has neither a CRC instruction nor carry-less multiply.
//
// One wavefront per byte range.  CRC is GF(2)-linear: absorbing dword w into state s is s' = X32(s ^ w), where Xk =
// "multiply by x^k mod P" (k reflected shift/xor steps).  So with dwords w_0..w_{N-1}
//     state = XOR_t  X_{32(N-t)}(w_t)                       (the 0xFFFFFFFF init is XORed into the first 4 bytes)
// The message is right-aligned on a 256-byte grid by (virtual) leading zero bytes, which change nothing.  Lane l
// then owns dwords l, l+64, l+128, .. of the grid (coalesced 256 B wave loads) and runs a Horner recurrence
//     acc_l = X2048(acc_l) ^ w
// where X2048 is a constant GF(2) map applied as 32 {bit-extract, and-constant, xor} triples -- no table, no LDS.
// A 6-level butterfly (left half times X_{32*2^k}, xor with the partner lane) folds the 64 accumulators, one X32
// finishes.  Cost: ~100 VALU ops per 256 bytes per wave.
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

__device__ __forceinline__ u32 xstep8(u32 v)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) v = (v >> 1) ^ ((v & 1u) ? kPoly : 0u);
    return v;
}

__global__ __launch_bounds__(SNP_WAVE) void k_crc32c(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                    const u32* __restrict__ in_len, u32 nblocks, int masked,
                                                    u32* __restrict__ out_crc, const u32* __restrict__ expect,
                                                    i32* __restrict__ status)
{
    const u32 b = blockIdx.x;
    if (b >= nblocks) return;
    const u32 lane = lane_id();
    const u8* src = in + in_off[b];
    const u32 n = bcast_first(in_len[b]);

    u32 crc;
    if (n < 4) {                                                        // tiny inputs: plain byte steps (uniform)
        u32 s = 0xffffffffu;
        for (u32 i = 0; i < n; ++i) s = xstep8(s ^ src[i]);
        crc = s ^ 0xffffffffu;
    } else {
        const u32 padb = (256u - (n & 255u)) & 255u;                    // virtual leading zero bytes
        const u32 rows = (n + padb) >> 8;
        u32 acc = 0;
        for (u32 i = 0; i < rows; ++i) {
            const i32 rbyte = static_cast<i32>(i * 256u + lane * 4u) - static_cast<i32>(padb);   // offset of this lane's dword in src
            u32 w;
            if (rbyte >= 4) {
                w = ld32u(src + rbyte);
            } else {                                                    // first rows only: leading pad and the init xor
                w = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32 bi = rbyte + j;
                    if (bi >= 0) {
                        u32 byte = src[bi];
                        if (bi < 4) byte ^= 0xffu;                      // init 0xFFFFFFFF == first four bytes inverted
                        w |= byte << (8 * j);
                    }
                }
            }
            acc = xmul<2048>(acc) ^ w;
        }
        // fold the 64 lane accumulators: total = XOR_l X_{32(63-l)}(acc_l)
        u32 v = acc;
        if (!(lane & 1)) v = xmul<32>(v);
        v ^= __shfl_xor(v, 1, 64);
        if (!(lane & 2)) v = xmul<64>(v);
        v ^= __shfl_xor(v, 2, 64);
        if (!(lane & 4)) v = xmul<128>(v);
        v ^= __shfl_xor(v, 4, 64);
        if (!(lane & 8)) v = xmul<256>(v);
        v ^= __shfl_xor(v, 8, 64);
        if (!(lane & 16)) v = xmul<512>(v);
        v ^= __shfl_xor(v, 16, 64);
        if (!(lane & 32)) v = xmul<1024>(v);
        v ^= __shfl_xor(v, 32, 64);
        crc = xmul<32>(v) ^ 0xffffffffu;                                // Crc32CAlgorithm.cs:48,153 final xor
    }
    if (masked) crc = crc32c_mask(crc);                                 // ApplyMask  :156-158
    if (lane == 0) {
        if (out_crc) out_crc[b] = crc;
        // framing verify: "Chunk CRC mismatch."  SnappyStreamDecompressor.cs:127-131,170-174
        if (expect && status && status[b] == SNP_OK && expect[b] != crc) status[b] = SNP_ERR_CRC_MISMATCH;
    }
}

}  // namespace

extern "C" hipError_t snp_launch_crc32c(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, int masked,
                                        u32* out_crc, const u32* expect, i32* status, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_crc32c, dim3(nblocks), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, masked,
                       out_crc, expect, status);
    return hipGetLastError();
}
