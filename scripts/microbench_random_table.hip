// microbench_random_table.hip -- what the memory system gives the lane compressor's hash-table access pattern.
// Every lane owns a private 64 KiB table (16384 x u32) in HBM, exactly like k_compress_lanes' workspace, and runs a
// chain of DEPENDENT probes: h = hash(state); v = table[h]; table[h] = i; state = f(state, v).  No input, no output,
// no compare: just the table traffic of `probes` probes per fragment.  Prints probes/s and the time 10 600 probes per
// fragment (the html-like workload's measured average) would take over 163 840 fragments.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_random_table.hip -o scripts/_bin/microbench_random_table
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: read only; 1: read + write same entry; 2: write only (entry never read: a partial-sector write);
//      3: read + non-temporal write; 4: non-temporal read + write;
//      5/6/7: read the entry, then write back the whole aligned 16 / 64 / 32 bytes around it (is a partially dirty
//      sector a DRAM read-modify-write?)
//      9: ONE returning atomic exchange per probe (round 3: what k_compress_lanes issues); 10: a non-returning atomic exchange
//      (an insert nobody reads back: is it cheaper than the plain store of mode 2?); 11: exchange probe + every third trip a plain
//      write-only insert into another bucket (the kernel's mix: 9 928 probes + 4 228 inserts per html-like fragment)
template <int ILP, int MODE>
__global__ __launch_bounds__(64) void k_walk(uint32_t* __restrict__ tables, uint32_t nfrag, uint32_t probes, uint32_t* __restrict__ sink, uint32_t span_log2, uint32_t interleave)
{
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nfrag) return;
    uint32_t* t = tables + (static_cast<uint64_t>(g) << span_log2);      // span_log2 = 14: 64 KiB per lane (the kernel's layout)
    // interleave = 1: sector k of every lane's table is adjacent to sector k of its neighbours (entry h of lane g at
    // ((h >> 4) * nfrag + g) * 16 + (h & 15)), so the traffic is uniform over the whole workspace at 64-byte granularity
    auto at = [&](uint32_t h) -> uint32_t* {
        return interleave ? tables + ((static_cast<uint64_t>(h >> 4) * nfrag + g) << 4) + (h & 15u) : t + h;
    };
    uint32_t st[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) st[k] = g * 2654435761u + k * 40503u + 1u;
    for (uint32_t i = 0; i < probes; i += ILP) {
        uint32_t v[ILP], h[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            h[k] = (st[k] * 0x1e35a7bdu) >> (32 - span_log2);
            v[k] = (MODE == 2 || MODE == 10) ? 0u : MODE == 4 ? __builtin_nontemporal_load(at(h[k]))
                   : (MODE == 9 || MODE == 11) ? __hip_atomic_exchange(at(h[k]), i + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *at(h[k]);
        }
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            if (MODE == 1 || MODE == 2 || MODE == 4) *at(h[k]) = i + k;
            if (MODE == 3) __builtin_nontemporal_store(i + k, t + h[k]);
            if (MODE == 10) (void)__hip_atomic_exchange(at(h[k]), i + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 11 && (i % 7u) < 3u) *at((h[k] * 40503u + 977u) & ((1u << span_log2) - 1u)) = i;   // 3 inserts per 7 probes
            if (MODE >= 5) {
                constexpr uint32_t W = MODE == 5 ? 4 : MODE == 6 ? 16 : MODE == 8 ? 32 : 8;     // dwords
                uint4* q = reinterpret_cast<uint4*>(t + (h[k] & ~(W - 1)));
                uint4 x[W / 4];
#pragma unroll
                for (uint32_t j = 0; j < W / 4; ++j) x[j] = q[j];
                x[0].x += i + k + v[k];
#pragma unroll
                for (uint32_t j = 0; j < W / 4; ++j) q[j] = x[j];
            }
            st[k] = st[k] * 1664525u + 1013904223u + v[k];
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) acc ^= st[k];
    if (acc == 0x12345678u) sink[0] = acc;
}

static uint32_t g_span_log2 = 14, g_interleave = 0;
template <int ILP, int MODE>
static void run(uint32_t* tables, uint32_t* sink, uint32_t nfrag, uint32_t probes, const char* name)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemsetAsync(tables, 0, static_cast<size_t>(nfrag) * 65536, 0));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_walk<ILP, MODE>), dim3((nfrag + 63) / 64), dim3(64), 0, 0, tables, nfrag, probes, sink, g_span_log2, g_interleave);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 1) {
            const double total = static_cast<double>(nfrag) * probes;
            printf("{\"case\": \"%s\", \"fragments\": %u, \"probes_per_fragment\": %u, \"ms\": %.3f, \"Gprobes_per_s\": %.2f, "
                   "\"ms_for_10600_probes_x_163840\": %.1f, \"table_bytes_per_lane\": %u}\n", name, nfrag, probes, ms, total / ms / 1e6,
                   10600.0 * 163840.0 / (total / ms), 4u << g_span_log2);
            fflush(stdout);
        }
    }
}

int main(int argc, char** argv)
{
    const uint32_t nfrag = argc > 1 ? atoi(argv[1]) : 163840;
    const uint32_t probes = argc > 2 ? atoi(argv[2]) : 4096;
    uint32_t *tables, *sink;
    CK(hipMalloc(&sink, 64));
    if (argc > 3 && argv[3][0] == 'p') {     // placement sweep: the same test with the tables allocated after k x 10 GiB of other buffers
        g_interleave = argv[3][1] == 'i';
        for (int k = 0; k < 26; ++k) {
            void* pad;
            if (k && hipMalloc(&pad, 10ull << 30) != hipSuccess) break;
            CK(hipMalloc(&tables, static_cast<size_t>(nfrag) * 65536));
            char name[96];
            snprintf(name, sizeof name, "read+write after %d x 10 GiB at %p", k, (void*)tables);
            run<1, 1>(tables, sink, nfrag, probes, name);
            CK(hipFree(tables));
        }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'a') {     // allocation flavours, several instances each
        const size_t bytes = static_cast<size_t>(nfrag) * 65536;
        for (int rep = 0; rep < 3; ++rep) {
            for (int flavour = 0; flavour < 4; ++flavour) {
                void* p = nullptr;
                hipError_t e = hipSuccess;
                const char* what = "";
                if (flavour == 0) { e = hipMalloc(&p, bytes); what = "hipMalloc"; }
                if (flavour == 1) { e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached); what = "hipExtMallocWithFlags(Uncached)"; }
                if (flavour == 2) { e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained); what = "hipExtMallocWithFlags(Finegrained)"; }
                if (flavour == 3) { e = hipMallocAsync(&p, bytes, 0); what = "hipMallocAsync"; }
                if (e != hipSuccess) { printf("{\"case\": \"%s failed: %s\"}\n", what, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
                char name[96];
                snprintf(name, sizeof name, "read+write on %s #%d", what, rep);
                run<1, 1>(static_cast<uint32_t*>(p), sink, nfrag, probes, name);
                // keep it allocated so the next one lands elsewhere
            }
        }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'o') {     // offset sweep inside ONE allocation: does the level depend on where the tables start?
        const size_t step = (argc > 4 ? atol(argv[4]) : 1024) << 20, steps = argc > 5 ? atoi(argv[5]) : 17;
        uint8_t* arena;
        CK(hipMalloc(&arena, static_cast<size_t>(nfrag) * 65536 + step * steps));
        for (size_t k = 0; k < steps; ++k) {
            char name[96];
            snprintf(name, sizeof name, "read+write at arena + %zu MiB", (k * step) >> 20);
            run<1, 1>(reinterpret_cast<uint32_t*>(arena + k * step), sink, nfrag, probes, name);
        }
        return 0;
    }
    CK(hipMalloc(&tables, static_cast<size_t>(nfrag) * 65536));
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipMemsetAsync(tables, 0, static_cast<size_t>(nfrag) * 65536, 0));
        CK(hipEventRecord(a, 0));
        CK(hipMemsetAsync(tables, 0, static_cast<size_t>(nfrag) * 65536, 0));
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("{\"case\": \"memset tables\", \"bytes\": %zu, \"ms\": %.3f}\n", static_cast<size_t>(nfrag) * 65536, ms);
    }
    if (argc > 3 && argv[3][0] == 's') {     // footprint sweep: same probes, smaller table span per lane (TLB reach / locality)
        for (uint32_t sl = 14; sl >= 10; sl -= 2) {
            g_span_log2 = sl;
            run<1, 1>(tables, sink, nfrag, probes, "read+write, 1 chain per lane");
            run<1, 0>(tables, sink, nfrag, probes, "read only, 1 chain per lane");
        }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'x') {     // round 3: the exchange forms
        run<1, 1>(tables, sink, nfrag, probes, "load + store, 1 chain per lane");
        run<1, 9>(tables, sink, nfrag, probes, "returning atomic exchange, 1 chain per lane");
        run<1, 2>(tables, sink, nfrag, probes, "write only (plain store), 1 chain per lane");
        run<1, 10>(tables, sink, nfrag, probes, "write only (non-returning atomic exchange), 1 chain per lane");
        run<1, 11>(tables, sink, nfrag, probes, "exchange probes + 3 plain inserts per 7 probes, 1 chain per lane");
        run<2, 9>(tables, sink, nfrag, probes, "returning atomic exchange, 2 chains per lane");
        run<1, 9>(tables, sink, nfrag, probes, "returning atomic exchange, 1 chain per lane (again)");
        return 0;
    }
    if (argc > 3) {      // cache-resident sweep: few fragments, many chains
        run<1, 1>(tables, sink, nfrag, probes, "read+write, 1 chain per lane");
        run<8, 1>(tables, sink, nfrag, probes, "read+write, 8 chains per lane");
        run<8, 0>(tables, sink, nfrag, probes, "read only, 8 chains per lane");
        return 0;
    }
    run<1, 0>(tables, sink, nfrag, probes, "read only, 1 chain per lane");
    run<1, 1>(tables, sink, nfrag, probes, "read+write, 1 chain per lane");
    run<4, 1>(tables, sink, nfrag, probes, "read+write, 4 chains per lane");
    run<1, 2>(tables, sink, nfrag, probes, "write only (no read of the sector), 1 chain per lane");
    run<1, 5>(tables, sink, nfrag, probes, "read + write back aligned 16 B, 1 chain per lane");
    run<1, 7>(tables, sink, nfrag, probes, "read + write back aligned 32 B, 1 chain per lane");
    run<1, 6>(tables, sink, nfrag, probes, "read + write back aligned 64 B, 1 chain per lane");
    run<1, 8>(tables, sink, nfrag, probes, "read + write back aligned 128 B, 1 chain per lane");
    run<1, 1>(tables, sink, nfrag, probes, "read+write, 1 chain per lane (again)");
    return 0;
}
