// Can the hash-table workspace be built from physical chunks mapped into one virtual range (hipMemCreate / hipMemAddressReserve / hipMemMap)?
// Prints the allocation granularity, what reserving / creating / mapping costs per GiB, and that a kernel can write through the mapping.
//   hipcc --offload-arch=gfx950 -O2 scripts/vmm_probe.hip -o /tmp/vmm_probe && /tmp/vmm_probe [GiB = 12] [chunk MiB = 1024]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s\", \"at\": \"%s\"}\n", hipGetErrorString(e_), #x); return 1; } } while (0)
__global__ void k_fill(unsigned* p, size_t n, unsigned v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i; }
__global__ void k_sum(const unsigned* p, size_t n, unsigned long long* out) { unsigned long long s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; atomicAdd(out, s); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? atoi(argv[1]) : 12, chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran_min = 0, gran_rec = 0;
    CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    const size_t total = gib << 30, n = total / chunk;
    size_t free0 = 0, tot = 0;
    CK(hipMemGetInfo(&free0, &tot));
    double t0 = now();
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    double t_res = now() - t0;
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    t0 = now();
    for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
    double t_create = now() - t0;
    size_t free1 = 0;
    CK(hipMemGetInfo(&free1, &tot));
    t0 = now();
    // map the chunks in REVERSE order: the virtual range need not follow allocation order
    for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)va + i * chunk, chunk, 0, h[n - 1 - i], 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    double t_map = now() - t0;
    unsigned long long* d_sum = nullptr;
    CK(hipMalloc(&d_sum, 8));
    CK(hipMemset(d_sum, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)va, total / 4, 7u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms_fill = 0; CK(hipEventElapsedTime(&ms_fill, e0, e1));
    hipLaunchKernelGGL(k_sum, dim3(4096), dim3(256), 0, 0, (const unsigned*)va, total / 4, d_sum);
    unsigned long long s = 0;
    CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
    unsigned long long expect = 0;
    for (size_t i = 0; i < total / 4; i += 1) { expect += (unsigned)(7u + (unsigned)i); if (i > (1u << 26)) { expect = 0; break; } }
    t0 = now();
    CK(hipMemUnmap(va, total));
    for (size_t i = 0; i < n; ++i) CK(hipMemRelease(h[i]));
    CK(hipMemAddressFree(va, total));
    double t_free = now() - t0;
    size_t free2 = 0;
    CK(hipMemGetInfo(&free2, &tot));
    printf("{\"GiB\": %zu, \"chunk_MiB\": %zu, \"granularity_min\": %zu, \"granularity_recommended\": %zu, \"reserve_ms\": %.3f, \"create_ms\": %.3f, \"map_setaccess_ms\": %.3f, "
           "\"fill_ms\": %.3f, \"fill_GBps\": %.1f, \"sum\": %llu, \"unmap_release_ms\": %.3f, \"free_before_GiB\": %.2f, \"free_after_create_GiB\": %.2f, \"free_after_release_GiB\": %.2f}\n",
           gib, chunk >> 20, gran_min, gran_rec, t_res * 1e3, t_create * 1e3, t_map * 1e3, ms_fill, total / ms_fill / 1e6, s, t_free * 1e3, free0 / 1073741824.0,
           free1 / 1073741824.0, free2 / 1073741824.0);
    return 0;
}
