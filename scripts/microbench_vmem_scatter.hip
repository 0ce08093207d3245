// microbench_vmem_scatter.hip -- what one vector-memory wave-instruction costs the texture path (TA/TCP) on gfx950 when its lanes
// address memory the way the decompressor's lanes do: per-lane UNALIGNED 8/16-byte accesses scattered over a window of the
// wavefront's own recent output, with only some lanes active.  The sub-chain decoder's PMC says TA_TA_BUSY ~ 70 % with no other
// unit saturated, and removing its first-pass copies (6 % of its instructions) removes 25 % of its time: this prices the
// candidates (fewer active lanes?  clustered addresses?  aligned write-out?) before the kernel is rebuilt around them.
// Every wavefront (64-thread workgroups, 32 per CU) owns a private region of REGION bytes, all L2-resident together when
// REGION is small; it issues ITER x 8 operations of one kind, the 8 in flight together, and the kernel time / (ITER * 8 * 32)
// is reported as cycles per wave-instruction per CU at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_vmem_scatter.hip -o scripts/_bin/microbench_vmem_scatter
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct __attribute__((packed)) U64 { uint64_t v; };
struct __attribute__((packed)) U128 { uint32_t v[4]; };

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// OP 0: 16-byte loads at random byte offsets inside `window` bytes        (first-pass copy sources)
// OP 1: 8-byte loads, lane l at base + ~3.2 * l bytes                      (the tag-byte gather of a batch: clustered, unaligned)
// OP 2: 16-byte loads, lane l at base + 16 l + mis                         (coalesced: input staging; mis = 0 aligned)
// OP 3: 16-byte stores, lane l at base + 16 l + mis                        (write-out of a staged run)
// OP 4: 1-byte stores, lanes 0..active-1 at base + l                       (the byte tail of a write-out)
// OP 5: 16-byte stores at random byte offsets inside `window`              (unstaged copies, for reference)
template <int OP>
__global__ __launch_bounds__(64) void k_vmem(uint8_t* buf, uint32_t region, uint32_t window, uint32_t active, uint32_t mis, uint32_t iters,
                                             uint32_t* sink)
{
    const uint32_t lane = threadIdx.x;
    uint8_t* const base = buf + static_cast<uint64_t>(blockIdx.x) * region;
    uint32_t acc = 0, rng = mix(blockIdx.x * 64u + lane + 1u);
    const bool on = lane < active;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint32_t at = (i * 1024u) % (region - window - 2048u);       // the window slides through the region like a block's output does
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            rng = rng * 1664525u + 1013904223u;
            const uint32_t r = (rng >> 8) % (window - 16u);
            if (OP == 0) { if (on) { const U128 t = *reinterpret_cast<const U128*>(base + at + r); acc += t.v[0] ^ t.v[3]; } }
            if (OP == 1) { if (on) { acc += static_cast<uint32_t>(reinterpret_cast<const U64*>(base + at + k * 256 + (lane * 13u >> 2) + mis)->v); } }
            if (OP == 2) { if (on) { const U128 t = *reinterpret_cast<const U128*>(base + at + k * 1024u % window + lane * 16u + mis); acc += t.v[0] ^ t.v[3]; } }
            if (OP == 3) { if (on) { U128 t; t.v[0] = acc; t.v[1] = i; t.v[2] = k; t.v[3] = lane; *reinterpret_cast<U128*>(base + at + (k & 1) * 1024u + lane * 16u + mis) = t; } }
            if (OP == 4) { if (on) base[at + k * 64u + lane + mis] = static_cast<uint8_t>(i); }
            if (OP == 5) { if (on) { U128 t; t.v[0] = acc; t.v[1] = i; t.v[2] = k; t.v[3] = lane; *reinterpret_cast<U128*>(base + at + r) = t; } }
        }
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

template <int OP>
static double run(uint8_t* buf, uint32_t region, uint32_t window, uint32_t active, uint32_t mis, uint32_t* sink)
{
    const uint32_t iters = 400, wpc = 32, grid = 256 * wpc;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_vmem<OP>, dim3(grid), dim3(64), 0, 0, buf, region, window, active, mis, iters, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_vmem<OP>, dim3(grid), dim3(64), 0, 0, buf, region, window, active, mis, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 * 2.4e9 / (static_cast<double>(iters) * 8 * wpc);
}

int main()
{
    const uint32_t region = 65536;                       // one block's output per wavefront: 8192 x 64 KiB = 512 MiB
    uint8_t* buf;
    uint32_t* sink;
    CK(hipMalloc(&buf, static_cast<size_t>(8192) * region + 4096));
    CK(hipMemset(buf, 1, static_cast<size_t>(8192) * region + 4096));
    CK(hipMalloc(&sink, 4));
    for (uint32_t window : {1024u, 4096u, 32768u})
        for (uint32_t active : {64u, 32u, 16u, 8u}) {
            printf("{\"op\": \"load16_random\", \"window\": %u, \"active\": %u, \"cyc\": %.1f}\n", window, active, run<0>(buf, region, window, active, 0, sink));
            fflush(stdout);
        }
    for (uint32_t active : {64u, 32u, 16u})
        for (uint32_t mis : {0u, 1u}) {
            printf("{\"op\": \"load8_clustered\", \"active\": %u, \"mis\": %u, \"cyc\": %.1f}\n", active, mis, run<1>(buf, region, 4096, active, mis, sink));
            fflush(stdout);
        }
    for (uint32_t active : {64u, 44u})
        for (uint32_t mis : {0u, 1u, 4u, 8u}) {
            printf("{\"op\": \"load16_coalesced\", \"active\": %u, \"mis\": %u, \"cyc\": %.1f}\n", active, mis, run<2>(buf, region, 4096, active, mis, sink));
            printf("{\"op\": \"store16_coalesced\", \"active\": %u, \"mis\": %u, \"cyc\": %.1f}\n", active, mis, run<3>(buf, region, 4096, active, mis, sink));
            fflush(stdout);
        }
    for (uint32_t active : {15u, 8u, 1u})
        for (uint32_t mis : {0u, 5u}) {
            printf("{\"op\": \"store1_tail\", \"active\": %u, \"mis\": %u, \"cyc\": %.1f}\n", active, mis, run<4>(buf, region, 4096, active, mis, sink));
            fflush(stdout);
        }
    for (uint32_t active : {64u, 32u, 16u}) {
        printf("{\"op\": \"store16_random\", \"window\": 1024, \"active\": %u, \"cyc\": %.1f}\n", active, run<5>(buf, region, 1024, active, 0, sink));
        fflush(stdout);
    }
    return 0;
}
