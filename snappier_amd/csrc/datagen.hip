// datagen.hip -- synthetic workload generators for bench.py and the -m gpu tests (NOT part of the codec library;
// built as libsnappier_datagen.so).  Same arithmetic as tests/datagen.py, which is the CPU statement of it.
//   corpus blocks: block b takes file f = b mod nfiles, tiles it cyclically from
//     offset (b*4099) mod len_f, then byte q is decided by draw q of the block's splitmix64 stream (seed ^ b):
//     r = mix64(seed^b + (q+1)*GAMMA); r % 100 == 0 -> byte = (r >> 32) & 0xff.
//   low-entropy blocks (config 3): runs; per run two draws r, r2 of splitmix64(seed ^ b): r % 10 != 0 -> pattern run,
//     period P = PERIODS[(r>>8)&7], length 16 + (r>>16) % 497; else noise run, P = L = 1 + (r>>16) % 16;
//     byte j of the run = (mix64(r2 + (j % P)) >> 24) & 0xff.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

namespace {
constexpr u64 GAMMA = 0x9E3779B97F4A7C15ull;
__device__ __forceinline__ u64 mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_gen_corpus(const u8* __restrict__ corpus, const u64* __restrict__ file_off,
                                                   const u32* __restrict__ file_len, u32 nfiles, u64 first_block,
                                                   u64 seed, u32 block_bytes, u8* __restrict__ out)
{
    const u64 b = first_block + blockIdx.x;
    const u32 f = static_cast<u32>(b % nfiles);
    const u32 L = file_len[f];
    const u8* src = corpus + file_off[f];
    const u32 start = static_cast<u32>((b * 4099ull) % L);
    const u64 s0 = seed ^ b;
    u8* dst = out + static_cast<u64>(blockIdx.x) * block_bytes;
    for (u32 q0 = threadIdx.x * 4; q0 < block_bytes; q0 += 256 * 4) {
        u32 w = 0;
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
            const u32 q = q0 + j;
            u32 byte = src[(start + q) % L];
            const u64 r = mix64(s0 + (static_cast<u64>(q) + 1) * GAMMA);
            if (r % 100 == 0) byte = static_cast<u32>(r >> 32) & 0xffu;
            w |= byte << (8 * j);
        }
        *reinterpret_cast<u32*>(dst + q0) = w;
    }
}

__constant__ u32 k_periods[8] = {1, 2, 3, 4, 7, 8, 16, 64};

__global__ __launch_bounds__(64) void k_gen_low_entropy(u64 first_block, u64 seed, u32 block_bytes, u8* __restrict__ out)
{
    const u64 b = first_block + blockIdx.x;
    const u32 lane = threadIdx.x;
    u8* dst = out + static_cast<u64>(blockIdx.x) * block_bytes;
    u64 st = seed ^ b;
    u32 pos = 0;
    while (pos < block_bytes) {
        st += GAMMA;
        const u64 r = mix64(st);
        st += GAMMA;
        const u64 r2 = mix64(st);
        u32 P, L;
        if (r % 10 != 0) { P = k_periods[(r >> 8) & 7]; L = 16 + static_cast<u32>((r >> 16) % 497); }
        else { L = 1 + static_cast<u32>((r >> 16) % 16); P = L; }
        if (L > block_bytes - pos) L = block_bytes - pos;
        for (u32 j = lane; j < L; j += 64) dst[pos + j] = static_cast<u8>(mix64(r2 + (j % P)) >> 24);
        pos += L;
    }
}
}  // namespace

extern "C" int snp_gen_corpus_blocks(const u8* corpus, const u64* file_off, const u32* file_len, u32 nfiles,
                                     u64 first_block, u32 nblocks, u64 seed, u32 block_bytes, u8* out, void* stream)
{
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(k_gen_corpus, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), corpus, file_off,
                       file_len, nfiles, first_block, seed, block_bytes, out);
    return static_cast<int>(hipGetLastError());
}

extern "C" int snp_gen_low_entropy_blocks(u64 first_block, u32 nblocks, u64 seed, u32 block_bytes, u8* out, void* stream)
{
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(k_gen_low_entropy, dim3(nblocks), dim3(64), 0, static_cast<hipStream_t>(stream), first_block,
                       seed, block_bytes, out);
    return static_cast<int>(hipGetLastError());
}
