// snp_device.h -- shared device-side helpers for the gfx950 Snappy block codec kernels.
// One 64-lane wavefront owns one Snappy block; everything here is wave-level (no __syncthreads anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/snappier_hip.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

#define SNP_WAVE 64

// The SNAPPIER_HIP_* environment variables are test and A/B knobs (kernel variants, layouts, thresholds): the product library reads NONE of them.
// Only variant libraries built with -DSNAPPIER_HIP_DEBUG_ENV do (scripts/build_variant.sh, LAB=1: snappier_amd/variants/, loaded by A/B scripts on request; no test loads one).
#ifdef SNAPPIER_HIP_DEBUG_ENV
#include <stdlib.h>
#define SNP_GETENV(name) getenv(name)
#else
#define SNP_GETENV(name) static_cast<const char*>(nullptr)
#endif

// The lane compressor's hash-table workspace (compress_lanes.hip): up to 16 separately allocated PIECES of `piece_frags` fragments' tables each
// (a multiple of 64, so a workgroup's tables never straddle two pieces); fragment f's table is table f % piece_frags of piece f / piece_frags.
// A workspace that is one allocation is one piece with piece_frags = 0xffffffc0.  Why pieces: capi_pool.hip, build_tables.
#define SNP_TABLE_PIECES_MAX 16
struct snp_table_pieces {
    uint32_t* p[SNP_TABLE_PIECES_MAX];
    uint32_t piece_frags;
    uint32_t n;
};

// Launch shape of the lane compressor (compress_lanes.hip), per context: SNP_OPT_COMPRESS_LANE_* (include/snappier_hip.h).  0 / -1 = chosen
// from the batch size and the context's hint; results never depend on it (tests/test_gpu_parity.py runs every combination against the oracle).
struct snp_lane_tuning {
    int lanes_per_wave;   // 0 auto | 8 | 16 | 32 | 64 fragments per wavefront
    int opts;             // -1 auto | a mask of the kernel's kOpt* bits (0..255)
    int probes;           // 0 auto | 1..4 probes of one lane's scan issued together
    int small_bytes;      // -1 auto (the hint) | 0 never | LDS slot bytes of the launch that keeps small fragments' input in LDS (<= 2048)
    int small_lanes;      // 0 auto (32) | 16 | 32 | 64 lanes per wavefront of that launch
    int hint;             // what the context learnt from its previous batch: 32 = fragments <= 512 B, 256 = <= 4 KiB, bits 9-14 = (longest + 15) / 16 when in (80, 768]
};

// Unaligned little-endian accesses.  gfx950 under HSA runs in unaligned-access mode, so a dword access at any byte
// address is one global_load_dword / global_store_dword (the packed struct tells the compiler align = 1).
struct __attribute__((packed)) snp_u32_unaligned { u32 v; };
struct __attribute__((packed)) snp_u64_unaligned { u64 v; };
struct __attribute__((packed)) snp_u128_unaligned { u32 v[4]; };

__device__ __forceinline__ u32 ld32u(const u8* p) { return reinterpret_cast<const snp_u32_unaligned*>(p)->v; }
__device__ __forceinline__ u64 ld64u(const u8* p) { return reinterpret_cast<const snp_u64_unaligned*>(p)->v; }
__device__ __forceinline__ void st32u(u8* p, u32 v) { reinterpret_cast<snp_u32_unaligned*>(p)->v = v; }

__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ u32 bcast_first(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u32 read_lane(u32 v, u32 lane) { return __builtin_amdgcn_readlane(v, lane); }
// 64-bit masks: lanes strictly below `lane`
__device__ __forceinline__ u64 lanes_below(u32 lane) { return lane >= 64 ? ~0ull : ((1ull << lane) - 1ull); }

// Wave-wide memcpy for a wave-uniform (dst, src, len): 16 B per lane per step while a full 1 KiB remains, then
// 4 B per lane, then a byte tail.  src and dst must not overlap.  All lanes must call it.
__device__ __forceinline__ void wave_copy(u8* dst, const u8* src, u32 len, u32 lane)
{
    u32 done = 0;
    while (len - done >= 1024) {
        snp_u128_unaligned w = *reinterpret_cast<const snp_u128_unaligned*>(src + done + lane * 16);
        *reinterpret_cast<snp_u128_unaligned*>(dst + done + lane * 16) = w;
        done += 1024;
    }
    while (len - done >= 256) {
        st32u(dst + done + lane * 4, ld32u(src + done + lane * 4));
        done += 256;
    }
#pragma clang loop vectorize(disable) unroll(disable)
    for (u32 k = done + lane; k < len; k += 64) dst[k] = src[k];   // (< 256 bytes: at most four trips -- not worth the unrolled code)
}

// Framing mask  Crc32CAlgorithm.ApplyMask  (Snappier/Internal/Crc32CAlgorithm.cs:156-158)
__host__ __device__ __forceinline__ u32 crc32c_mask(u32 x) { return ((x >> 15) | (x << 17)) + 0xa282ead8u; }

// A few control words set to zero ON THE STREAM.  Not hipMemsetAsync: as a node of a captured hipGraph a 4-byte memset runs on the first launch of the
// graph only (ROCm 7.0, scripts/probe_graph_memset.py), and the batch entry points are meant to be capturable (tests/test_gpu_graph_capture.py).
namespace {
__global__ void k_zero_words(u32* p, u32 n)
{
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
}  // namespace
static inline hipError_t snp_zero_words_async(u32* p, u32 n, hipStream_t stream)
{
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, p, n);
    return hipGetLastError();
}
