// microbench_lds_gather.hip -- price list for an OUTPUT-granular decoder (round 4): what one LDS wave-instruction costs a CU on gfx950
// when the 64 lanes gather from per-lane addresses the way a lane-per-output-byte / lane-per-output-dword executor would.
//   pattern 0  linear     lane l -> base + l * width
//   pattern 1  random     a 4-byte-aligned random address inside a 4 KiB ring per lane (dword gathers), any byte for u8
//   pattern 2  runs       runs of 11 consecutive lanes read consecutive bytes (u8) / consecutive dwords, each run at a random base
//   pattern 3  broadcast  every lane the same address
// Every wavefront issues ITER x 8 independent operations of one kind (inline asm, one s_waitcnt per 8) and the kernel time /
// (ITER * 8 * waves per CU) is reported as cycles per wave-instruction per CU at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_lds_gather.hip -o scripts/_bin/microbench_lds_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { OP_RD_B32, OP_RD2_B32, OP_RD_U8, OP_BPERM, OP_WR_B8, OP_WR_B32, OP_RD_B64, OP_RD_B128, OP_RD2ST64, OP_COUNT };

template <int OP>
__global__ __launch_bounds__(64) void k_lds(uint32_t iters, const uint32_t* __restrict__ addr_tab, uint32_t* sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[4096 + 64];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4160 / 4; i += 64) reinterpret_cast<uint32_t*>(buf)[i] = i * 2654435761u;
    __syncthreads();
    // 8 address sets so consecutive instructions do not hit the same lines
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = addr_tab[k * 64 + lane] + static_cast<uint32_t>(reinterpret_cast<uintptr_t>(buf));
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            if (OP == OP_RD_B32) asm volatile("ds_read_b32 %0, %1" : "=v"(r0) : "v"(a[k]));
            if (OP == OP_RD2_B32) { uint64_t t; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(t) : "v"(a[k])); r0 = static_cast<uint32_t>(t); r1 = static_cast<uint32_t>(t >> 32); }
            if (OP == OP_RD2ST64) { uint64_t t; asm volatile("ds_read2_b32 %0, %1 offset1:16" : "=v"(t) : "v"(a[k])); r0 = static_cast<uint32_t>(t); r1 = static_cast<uint32_t>(t >> 32); }
            if (OP == OP_RD_U8) asm volatile("ds_read_u8 %0, %1" : "=v"(r0) : "v"(a[k]));
            if (OP == OP_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r0) : "v"(a[k]), "v"(acc));
            if (OP == OP_WR_B8) asm volatile("ds_write_b8 %0, %1" :: "v"(a[k]), "v"(acc));
            if (OP == OP_WR_B32) asm volatile("ds_write_b32 %0, %1" :: "v"(a[k]), "v"(acc));
            if (OP == OP_RD_B64) { uint64_t t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a[k])); r0 = static_cast<uint32_t>(t); r1 = static_cast<uint32_t>(t >> 32); }
            if (OP == OP_RD_B128) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                u4 t;
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(a[k]));
                r0 = t.x; r1 = t.y; r2 = t.z; r3 = t.w;
            }
            if (k == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
            if (k == 7) acc ^= r0 ^ r1 ^ r2 ^ r3;
        }
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static void fill(uint32_t* tab, int op, int pattern)
{
    const uint32_t width = (op == OP_RD_U8 || op == OP_WR_B8) ? 1 : (op == OP_RD_B64) ? 8 : (op == OP_RD_B128) ? 16 : 4;
    const uint32_t align = (op == OP_RD_U8 || op == OP_WR_B8) ? 1 : width;
    for (int k = 0; k < 8; ++k) {
        uint32_t run_base = 0;
        for (int l = 0; l < 64; ++l) {
            uint32_t v = 0;
            if (op == OP_BPERM) {
                v = pattern == 0 ? 4u * l : pattern == 3 ? 4u * 5 : pattern == 2 ? 4u * ((l + 64 - 11) & 63) : 4u * (rnd() & 63);
            } else if (pattern == 0) v = (k & 1 ? 2048 : 0) + l * width;
            else if (pattern == 1) v = (rnd() % (4096 / align)) * align;
            else if (pattern == 2) {
                if (l % 11 == 0) run_base = (rnd() % ((4096 - 11 * width) / align)) * align;
                v = run_base + (l % 11) * width;
            } else v = (k * 64) & 4095;
            tab[k * 64 + l] = v;
        }
    }
}

template <int OP>
static double run(uint32_t waves_per_cu, const uint32_t* d_tab, uint32_t* sink)
{
    const uint32_t iters = 2000;
    const uint32_t grid = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_lds<OP>, dim3(grid), dim3(64), 0, 0, iters, d_tab, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_lds<OP>, dim3(grid), dim3(64), 0, 0, iters, d_tab, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 * 2.4e9 / (static_cast<double>(iters) * 8 * waves_per_cu);
}

int main()
{
    uint32_t *sink, *d_tab;
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&d_tab, 512 * 4));
    const char* names[OP_COUNT] = {"ds_read_b32", "ds_read2_b32(+0,+4)", "ds_read_u8", "ds_bpermute_b32", "ds_write_b8", "ds_write_b32", "ds_read_b64", "ds_read_b128", "ds_read2_b32(+0,+64)"};
    const char* pats[4] = {"linear", "random", "runs11", "broadcast"};
    for (uint32_t wpc : {8u, 16u, 32u}) {
        for (int op = 0; op < OP_COUNT; ++op) {
            printf("{\"op\": \"%s\", \"waves_per_cu\": %u, \"cycles_per_wave_instruction_per_cu\": {", names[op], wpc);
            for (int pat = 0; pat < 4; ++pat) {
                uint32_t tab[512];
                fill(tab, op, pat);
                CK(hipMemcpy(d_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
                double c = 0;
                switch (op) {
                    case OP_RD_B32: c = run<OP_RD_B32>(wpc, d_tab, sink); break;
                    case OP_RD2_B32: c = run<OP_RD2_B32>(wpc, d_tab, sink); break;
                    case OP_RD_U8: c = run<OP_RD_U8>(wpc, d_tab, sink); break;
                    case OP_BPERM: c = run<OP_BPERM>(wpc, d_tab, sink); break;
                    case OP_WR_B8: c = run<OP_WR_B8>(wpc, d_tab, sink); break;
                    case OP_WR_B32: c = run<OP_WR_B32>(wpc, d_tab, sink); break;
                    case OP_RD_B64: c = run<OP_RD_B64>(wpc, d_tab, sink); break;
                    case OP_RD_B128: c = run<OP_RD_B128>(wpc, d_tab, sink); break;
                    default: c = run<OP_RD2ST64>(wpc, d_tab, sink); break;
                }
                printf("%s\"%s\": %.2f", pat ? ", " : "", pats[pat], c);
            }
            printf("}}\n");
            fflush(stdout);
        }
    }
    return 0;
}
