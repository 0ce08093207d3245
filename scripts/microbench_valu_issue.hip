// microbench_valu_issue.hip -- how many wave64 VALU instructions one CU of gfx950 issues per cycle (integer / bit ops of the kind the
// codec kernels are made of), at 4 .. 32 wavefronts per CU.  The round-3 bound of the decoder assumed two per cycle per CU (SIMD-32);
// the PMC ratio SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.05 quad-cycles says four cycles per instruction per SIMD, i.e. ONE per cycle per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_valu_issue.hip -o scripts/_bin/microbench_valu_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k_valu(uint32_t iters, uint32_t* sink)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    const uint32_t k = iters | 3;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            if (OP == 1) asm volatile("v_and_or_b32 %0, %0, %8, %1\n v_and_or_b32 %1, %1, %8, %2\n v_and_or_b32 %2, %2, %8, %3\n v_and_or_b32 %3, %3, %8, %4\n v_and_or_b32 %4, %4, %8, %5\n v_and_or_b32 %5, %5, %8, %6\n v_and_or_b32 %6, %6, %8, %7\n v_and_or_b32 %7, %7, %8, %0"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            if (OP == 2) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
            if (OP == 3) asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3\n v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3"
                                      : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
            if (OP == 4) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (OP == 5) asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            if (OP == 6) asm volatile("v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %1, %1, 1, 31\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %3, %3, 1, 31\n v_bfe_u32 %4, %4, 1, 31\n v_bfe_u32 %5, %5, 1, 31\n v_bfe_u32 %6, %6, 1, 31\n v_bfe_u32 %7, %7, 1, 31"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (OP == 7) asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23", "scc");
            if (OP == 8) asm volatile("v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %8\n s_add_u32 s21, s21, 1\n v_add_u32 %2, %2, %8\n s_add_u32 s22, s22, 1\n v_add_u32 %3, %3, %8\n s_add_u32 s23, s23, 1\n v_add_u32 %4, %4, %8\n s_add_u32 s20, s20, 1\n v_add_u32 %5, %5, %8\n s_add_u32 s21, s21, 1\n v_add_u32 %6, %6, %8\n s_add_u32 s22, s22, 1\n v_add_u32 %7, %7, %8\n s_add_u32 s23, s23, 1"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "s20", "s21", "s22", "s23", "scc");
        }
    }
    const uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ static_cast<uint32_t>(b0 ^ b1 ^ b2 ^ b3);
    if (acc == 0x12345679u) sink[0] = acc;
}

template <int OP>
static double run(uint32_t wpc, uint32_t* sink, int per_trip)
{
    const uint32_t iters = 4000, grid = 256 * wpc;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_valu<OP>, dim3(grid), dim3(64), 0, 0, iters, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_valu<OP>, dim3(grid), dim3(64), 0, 0, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return static_cast<double>(iters) * 4 * per_trip * wpc / (ms * 1e-3 * 2.4e9);   // wave-instructions per cycle per CU at 2.4 GHz
}

int main()
{
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    const char* names[9] = {"v_add_u32", "v_and_or_b32 (VOP3)", "v_cndmask_b32", "v_lshrrev_b64", "v_add_u32_dpp (+s_nop 1)", "v_pk_add_u16", "v_bfe_u32 (VOP3)", "s_add_u32 (SALU only)", "v_add_u32 + s_add_u32 interleaved (count = both)"};
    for (int op = 0; op < 9; ++op) {
        printf("{\"op\": \"%s\", \"wave_instructions_per_cycle_per_cu\": {", names[op]);
        bool first = true;
        for (uint32_t wpc : {4u, 8u, 16u, 32u}) {
            double v = 0;
            switch (op) {
                case 0: v = run<0>(wpc, sink, 8); break;
                case 1: v = run<1>(wpc, sink, 8); break;
                case 2: v = run<2>(wpc, sink, 8); break;
                case 3: v = run<3>(wpc, sink, 8); break;
                case 4: v = run<4>(wpc, sink, 8); break;
                case 5: v = run<5>(wpc, sink, 8); break;
                case 6: v = run<6>(wpc, sink, 8); break;
                case 7: v = run<7>(wpc, sink, 8); break;
                default: v = run<8>(wpc, sink, 16); break;
            }
            printf("%s\"%u\": %.3f", first ? "" : ", ", wpc, v);
            first = false;
        }
        printf("}}\n");
        fflush(stdout);
    }
    return 0;
}
