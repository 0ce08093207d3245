// microbench_lds_unaligned.hip -- what an LDS access at an arbitrary byte address costs on gfx950, per wave-instruction.
// The decompressor assembles a batch's output in LDS from tags that start at any byte (decompress.hip, staged batches), and
// PMC showed SQ_LDS_UNALIGNED_STALL at a third of its LDS time; this measures the price list the kernel is designed to.
// Every wavefront (64-thread workgroups, WAVES of them per CU) issues ITER x 8 operations of one kind, lane l at
//   base + l * STRIDE + MIS   (STRIDE = the access width, so the lanes tile the stage as the tags of a batch do),
// and the kernel time / (ITER * 8 * waves per CU) is reported as cycles per wave-instruction per CU at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_lds_unaligned.hip -o scripts/_bin/microbench_lds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct __attribute__((packed)) U16 { uint16_t v; };
struct __attribute__((packed)) U32 { uint32_t v; };
struct __attribute__((packed)) U64 { uint64_t v; };
struct __attribute__((packed)) U128 { uint32_t v[4]; };

// OP: 0 write 1 B, 1 write 2 B, 2 write 4 B, 3 write 8 B, 4 write 16 B, 5 read 1 B, 6 read 2, 7 read 4, 8 read 8, 9 read 16
template <int OP>
__global__ __launch_bounds__(64) void k_lds(uint32_t iters, uint32_t mis, uint32_t stride, uint32_t* sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[4096 + 64];
    const uint32_t lane = threadIdx.x;
    uint8_t* p = buf + lane * stride + mis;
    uint32_t acc = lane;
    for (uint32_t i = 0; i < 4096 / 4; i += 64) reinterpret_cast<uint32_t*>(buf)[i + lane < 1040 ? i + lane : 0] = lane;
    __syncthreads();
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint8_t* q = p + ((k & 1) ? 2048 : 0);       // two alternating spots so that consecutive stores are not merged
            if (OP == 0) *reinterpret_cast<volatile uint8_t*>(q) = static_cast<uint8_t>(acc);
            if (OP == 1) reinterpret_cast<volatile U16*>(q)->v = static_cast<uint16_t>(acc);
            if (OP == 2) reinterpret_cast<volatile U32*>(q)->v = acc;
            if (OP == 3) reinterpret_cast<volatile U64*>(q)->v = acc;
            if (OP == 4) { U128 t; t.v[0] = acc; t.v[1] = acc; t.v[2] = acc; t.v[3] = acc; *reinterpret_cast<U128*>(q) = t; asm volatile("" ::: "memory"); }
            if (OP == 5) acc += *reinterpret_cast<volatile uint8_t*>(q);
            if (OP == 6) acc += reinterpret_cast<volatile U16*>(q)->v;
            if (OP == 7) acc += reinterpret_cast<volatile U32*>(q)->v;
            if (OP == 8) acc += static_cast<uint32_t>(reinterpret_cast<volatile U64*>(q)->v);
            if (OP == 9) { asm volatile("" ::: "memory"); const U128 t = *reinterpret_cast<const U128*>(q); acc += t.v[0] ^ t.v[3]; }
        }
    }
    if (acc == 0x12345679u) sink[0] = acc;
}

template <int OP>
static double run(uint32_t waves_per_cu, uint32_t mis, uint32_t stride, uint32_t* sink)
{
    const uint32_t iters = 2000;
    const uint32_t grid = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_lds<OP>, dim3(grid), dim3(64), 0, 0, iters, mis, stride, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_lds<OP>, dim3(grid), dim3(64), 0, 0, iters, mis, stride, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 * 2.4e9 / (static_cast<double>(iters) * 8 * waves_per_cu);
}

int main()
{
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    const char* names[10] = {"ds_write_b8", "ds_write_b16", "ds_write_b32", "ds_write_b64", "ds_write_b128",
                             "ds_read_u8", "ds_read_u16", "ds_read_b32", "ds_read_b64", "ds_read_b128"};
    const uint32_t width[10] = {1, 2, 4, 8, 16, 1, 2, 4, 8, 16};
    for (uint32_t wpc : {8u, 32u}) {
        for (int op = 0; op < 10; ++op) {
            printf("{\"op\": \"%s\", \"waves_per_cu\": %u, \"cycles_per_wave_instruction_per_cu\": {", names[op], wpc);
            bool first = true;
            for (uint32_t mis : {0u, 1u, 2u, 3u, 4u, 5u, 8u, 9u}) {
                if (mis && mis % width[op] == 0 && mis != width[op]) continue;
                double c = 0;
                const uint32_t st = width[op] < 4 ? 4 : width[op];       // sub-dword accesses: one per dword (distinct banks)
                switch (op) {
                    case 0: c = run<0>(wpc, mis, st, sink); break;
                    case 1: c = run<1>(wpc, mis, st, sink); break;
                    case 2: c = run<2>(wpc, mis, st, sink); break;
                    case 3: c = run<3>(wpc, mis, st, sink); break;
                    case 4: c = run<4>(wpc, mis, st, sink); break;
                    case 5: c = run<5>(wpc, mis, st, sink); break;
                    case 6: c = run<6>(wpc, mis, st, sink); break;
                    case 7: c = run<7>(wpc, mis, st, sink); break;
                    case 8: c = run<8>(wpc, mis, st, sink); break;
                    default: c = run<9>(wpc, mis, st, sink); break;
                }
                printf("%s\"+%u\": %.2f", first ? "" : ", ", mis, c);
                first = false;
            }
            printf("}}\n");
            fflush(stdout);
        }
    }
    return 0;
}
