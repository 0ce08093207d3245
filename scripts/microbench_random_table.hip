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
#include <algorithm>
#include <chrono>
#include <utility>
#include <vector>

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

// Round 3: the same exchange walk over a workspace made of PIECES (separate allocations): lane g's table is table (g % per_piece) of
// piece (g / per_piece) % npieces -- per_piece < nfrag / npieces folds several lanes onto one table, which probes ONE small candidate with
// the whole grid's concurrency.
// KIND 0: atomic exchange (the compressor's probe), 1: dependent load only, 2: store only (nothing read back), 3: dependent load + store
template <int KIND>
__global__ __launch_bounds__(64) void k_walk_pieces_t(uint32_t* const* __restrict__ pieces, uint32_t npieces, uint32_t per_piece, uint32_t nfrag,
                                                      uint32_t probes, uint32_t* __restrict__ sink)
{
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nfrag) return;
    uint32_t* t = pieces[(g / per_piece) % npieces] + (static_cast<uint64_t>(g % per_piece) << 14);
    uint32_t st = g * 2654435761u + 1u;
    for (uint32_t i = 0; i < probes; ++i) {
        const uint32_t h = (st * 0x1e35a7bdu) >> 18;
        uint32_t v = 0;
        if (KIND == 0) v = __hip_atomic_exchange(t + h, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (KIND == 1 || KIND == 3) v = t[h];
        if (KIND == 2 || KIND == 3) t[h] = i;
        st = st * 1664525u + 1013904223u + v;
    }
    if (st == 0x12345678u) sink[0] = st;
}
__global__ __launch_bounds__(64) void k_walk_pieces(uint32_t* const* __restrict__ pieces, uint32_t npieces, uint32_t per_piece, uint32_t nfrag,
                                                    uint32_t probes, uint32_t* __restrict__ sink)
{
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nfrag) return;
    uint32_t* t = pieces[(g / per_piece) % npieces] + (static_cast<uint64_t>(g % per_piece) << 14);
    uint32_t st = g * 2654435761u + 1u;
    for (uint32_t i = 0; i < probes; ++i) {
        const uint32_t h = (st * 0x1e35a7bdu) >> 18;
        const uint32_t v = __hip_atomic_exchange(t + h, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st = st * 1664525u + 1013904223u + v;
    }
    if (st == 0x12345678u) sink[0] = st;
}
template <int KIND>
static float time_pieces_t(uint32_t* const* d_ptrs, uint32_t npieces, uint32_t per_piece, uint32_t nfrag, uint32_t probes, uint32_t* sink)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_walk_pieces_t<KIND>, dim3((nfrag + 63) / 64), dim3(64), 0, 0, d_ptrs, npieces, per_piece, nfrag, probes, sink);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
    }
    return ms;
}

static float time_pieces(uint32_t* const* d_ptrs, uint32_t npieces, uint32_t per_piece, uint32_t nfrag, uint32_t probes, uint32_t* sink)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k_walk_pieces, dim3((nfrag + 63) / 64), dim3(64), 0, 0, d_ptrs, npieces, per_piece, nfrag, probes, sink);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
    }
    return ms;
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
    if (argc > 3 && argv[3][0] == 'g') {     // round 3: prototype of the piece SEARCH the library runs when it allocates a large workspace
        const uint32_t parts = argc > 4 ? atoi(argv[4]) : 16, mult = argc > 5 ? atoi(argv[5]) : 3, climbs = argc > 6 ? atoi(argv[6]) : 64;
        const uint32_t per_piece = ((nfrag + parts - 1) / parts + 63) / 64 * 64;
        const size_t piece_bytes = static_cast<size_t>(per_piece) * 65536;
        std::vector<uint32_t*> cand;
        for (uint32_t k = 0; k < parts * mult; ++k) {
            uint32_t* p = nullptr;
            if (hipMalloc(&p, piece_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            CK(hipMemsetAsync(p, 0, piece_bytes, 0));
            cand.push_back(p);
        }
        CK(hipDeviceSynchronize());
        uint32_t** d_ptrs;
        CK(hipMalloc(&d_ptrs, 64 * sizeof(uint32_t*)));
        uint32_t trials = 0;
        auto probe = [&](const std::vector<uint32_t>& idx, uint32_t pr) {
            std::vector<uint32_t*> ptrs;
            for (uint32_t i : idx) ptrs.push_back(cand[i]);
            CK(hipMemcpy(d_ptrs, ptrs.data(), ptrs.size() * sizeof(uint32_t*), hipMemcpyHostToDevice));
            ++trials;
            return time_pieces(d_ptrs, static_cast<uint32_t>(ptrs.size()), per_piece, nfrag, pr, sink);
        };
        auto report = [&](const char* name, const std::vector<uint32_t>& set, double search_ms) {
            const float ms = probe(set, probes);
            printf("{\"case\": \"%s\", \"pieces\": %zu, \"probes\": %u, \"ms\": %.3f, \"Gprobes_per_s\": %.2f, \"search_ms\": %.1f, \"trials\": %u}\n", name, set.size(), probes, ms,
                   static_cast<double>(nfrag) * probes / ms / 1e6, search_ms, trials);
            fflush(stdout);
        };
        std::vector<uint32_t> firsts;
        for (uint32_t i = 0; i < parts; ++i) firsts.push_back(i);
        report("the first pieces allocated", firsts, 0);
        // A: alone scores; balanced pieces first, then slow pieces in complementary pairs
        auto t0 = std::chrono::steady_clock::now();
        auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        trials = 0;
        std::vector<float> alone(cand.size());
        float lo = 1e9f, hi = 0;
        for (uint32_t k = 0; k < cand.size(); ++k) { alone[k] = probe({k}, 512); lo = std::min(lo, alone[k]); hi = std::max(hi, alone[k]); }
        std::vector<uint32_t> balanced, slow, setA;
        for (uint32_t k = 0; k < cand.size(); ++k) (hi > 1.04f * lo && alone[k] > 0.5f * (lo + hi) ? slow : balanced).push_back(k);
        for (uint32_t k : balanced) if (setA.size() < parts) setA.push_back(k);
        std::vector<uint32_t> same, opposite;
        if (setA.size() < parts && slow.size() >= 2) {
            std::vector<float> pm(slow.size(), 0);
            float plo = 1e9f, phi = 0;
            for (size_t i = 1; i < slow.size(); ++i) { pm[i] = probe({slow[0], slow[i]}, 512); plo = std::min(plo, pm[i]); phi = std::max(phi, pm[i]); }
            same.push_back(slow[0]);
            for (size_t i = 1; i < slow.size(); ++i) (phi > 1.04f * plo && pm[i] < 0.5f * (plo + phi) ? opposite : same).push_back(slow[i]);
            for (size_t i = 0; setA.size() + 2 <= parts && i < same.size() && i < opposite.size(); ++i) { setA.push_back(same[i]); setA.push_back(opposite[i]); }
        }
        for (uint32_t k = 0; setA.size() < parts && k < cand.size(); ++k)
            if (std::find(setA.begin(), setA.end(), k) == setA.end()) setA.push_back(k);
        const double a_ms = since();
        printf("{\"case\": \"classes\", \"candidates\": %zu, \"balanced\": %zu, \"slow_same\": %zu, \"slow_opposite\": %zu, \"alone_lo\": %.3f, \"alone_hi\": %.3f}\n", cand.size(),
               balanced.size(), same.size(), opposite.size(), lo, hi);
        report("A: balanced pieces, then complementary pairs", setA, a_ms);
        // B: hill climb on the composed score from A (swap a spare piece in, keep it when the probe gets faster)
        auto climb = [&](std::vector<uint32_t> set, uint32_t n, const char* name) {
            t0 = std::chrono::steady_clock::now();
            trials = 0;
            std::vector<uint32_t> spare;
            for (uint32_t k = 0; k < cand.size(); ++k) if (std::find(set.begin(), set.end(), k) == set.end()) spare.push_back(k);
            float cur = probe(set, 768);
            uint32_t accepted = 0, rng = 12345;
            for (uint32_t t = 0; t < n && !spare.empty(); ++t) {
                rng = rng * 1664525u + 1013904223u;
                const uint32_t pos = (rng >> 8) % set.size();
                rng = rng * 1664525u + 1013904223u;
                const uint32_t sp = (rng >> 8) % spare.size();
                std::swap(set[pos], spare[sp]);
                const float ms = probe(set, 768);
                if (ms < cur * 0.996f) { cur = ms; ++accepted; } else std::swap(set[pos], spare[sp]);
            }
            const double ms = since();
            printf("{\"case\": \"climb\", \"accepted\": %u, \"of\": %u}\n", accepted, n);
            report(name, set, ms);
            return set;
        };
        climb(setA, climbs, "B: A, then hill climb");
        climb(firsts, climbs * 2, "C: the first pieces, then twice the hill climb");
        report("A again", setA, 0);
        for (uint32_t* p : cand) CK(hipFree(p));
        uint32_t* whole;
        for (int k = 0; k < 3; ++k) {
            CK(hipMalloc(&whole, static_cast<size_t>(nfrag) * 65536));
            CK(hipMemsetAsync(whole, 0, static_cast<size_t>(nfrag) * 65536, 0));
            CK(hipMemcpy(d_ptrs, &whole, sizeof(uint32_t*), hipMemcpyHostToDevice));
            const float ms = time_pieces(d_ptrs, 1, nfrag, nfrag, probes, sink);
            printf("{\"case\": \"one allocation #%d\", \"probes\": %u, \"ms\": %.3f, \"Gprobes_per_s\": %.2f}\n", k, probes, ms, static_cast<double>(nfrag) * probes / ms / 1e6);
        }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'k') {     // round 3: how many KINDS of memory are there?  pieces over all of device memory, classified against successive references
        const uint32_t parts = argc > 4 ? atoi(argv[4]) : 16, ncand = argc > 5 ? atoi(argv[5]) : 440;
        const uint32_t per_piece = ((nfrag + parts - 1) / parts + 63) / 64 * 64;
        const size_t piece_bytes = static_cast<size_t>(per_piece) * 65536;
        std::vector<uint32_t*> cand;
        for (uint32_t k = 0; k < ncand; ++k) {
            uint32_t* p = nullptr;
            if (hipMalloc(&p, piece_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            cand.push_back(p);
        }
        uint32_t** d_ptrs;
        CK(hipMalloc(&d_ptrs, 64 * sizeof(uint32_t*)));
        auto probe = [&](const std::vector<uint32_t>& idx, uint32_t pr) {
            std::vector<uint32_t*> ptrs;
            for (uint32_t i : idx) ptrs.push_back(cand[i]);
            CK(hipMemcpy(d_ptrs, ptrs.data(), ptrs.size() * sizeof(uint32_t*), hipMemcpyHostToDevice));
            return time_pieces(d_ptrs, static_cast<uint32_t>(ptrs.size()), per_piece, nfrag, pr, sink);
        };
        std::vector<float> alone(cand.size());
        float lo = 1e9f, hi = 0;
        for (uint32_t k = 0; k < cand.size(); ++k) { alone[k] = probe({k}, 512); lo = std::min(lo, alone[k]); hi = std::max(hi, alone[k]); }
        std::vector<int> kind(cand.size(), -1);                           // -1 unclassified, 0 = fast alone (straddles kinds), 1.. = kinds
        for (uint32_t k = 0; k < cand.size(); ++k) if (alone[k] < 0.5f * (lo + hi)) kind[k] = 0;
        std::vector<uint32_t> refs;
        for (int round = 1; round <= 8; ++round) {
            int ref = -1;
            for (uint32_t k = 0; k < cand.size(); ++k) if (kind[k] < 0) { ref = static_cast<int>(k); break; }
            if (ref < 0) break;
            refs.push_back(ref);
            kind[ref] = round;
            std::vector<std::pair<uint32_t, float>> pm;
            float plo = 1e9f, phi = 0;
            for (uint32_t k = 0; k < cand.size(); ++k) {
                if (kind[k] >= 0) continue;
                const float ms = probe({static_cast<uint32_t>(ref), k}, 512);
                pm.push_back({k, ms}); plo = std::min(plo, ms); phi = std::max(phi, ms);
            }
            uint32_t same = 1;
            for (auto& q : pm) if (phi <= 1.05f * plo || q.second > 0.5f * (plo + phi)) { kind[q.first] = round; ++same; }
            printf("{\"case\": \"kind\", \"round\": %d, \"ref\": %d, \"pair_lo_ms\": %.3f, \"pair_hi_ms\": %.3f, \"members\": %u}\n", round, ref, plo, phi, same);
        }
        printf("{\"case\": \"map\", \"piece_GiB\": %.3f, \"pieces\": %zu, \"alone_lo\": %.3f, \"alone_hi\": %.3f, \"kinds\": \"", piece_bytes / 1073741824.0, cand.size(), lo, hi);
        for (uint32_t k = 0; k < cand.size(); ++k) printf("%c", kind[k] == 0 ? 'F' : kind[k] < 0 ? '?' : static_cast<char>('0' + kind[k]));
        printf("\"}\n");
        // cross pairs of the references: which kinds complement each other?
        for (size_t i = 0; i < refs.size(); ++i)
            for (size_t j = i; j < refs.size(); ++j) {
                // (a reference paired with a second piece of its own kind when i == j)
                uint32_t b = refs[j];
                if (i == j) for (uint32_t k = 0; k < cand.size(); ++k) if (kind[k] == static_cast<int>(i + 1) && k != refs[i]) { b = k; break; }
                printf("{\"case\": \"reference pair\", \"kinds\": [%zu, %zu], \"ms\": %.3f}\n", i + 1, j + 1, probe({refs[i], b}, 512));
            }
        // which KIND OF ACCESS sees the kinds of memory?  16 pieces of kind 1 against 8 + 8 of kinds 1 and 2: exchanges, loads only, stores only, load + store
        {
            std::vector<std::vector<uint32_t>> mem(refs.size() + 1);
            for (uint32_t k = 0; k < cand.size(); ++k) if (kind[k] >= 1) mem[kind[k]].push_back(k);
            if (refs.size() >= 2 && mem[1].size() >= parts && mem[2].size() >= parts / 2) {
                std::vector<uint32_t> one(mem[1].begin(), mem[1].begin() + parts), two;
                for (uint32_t i = 0; i < parts / 2; ++i) { two.push_back(mem[1][i]); two.push_back(mem[2][i]); }
                auto run4 = [&](const char* name, const std::vector<uint32_t>& idx) {
                    std::vector<uint32_t*> ptrs;
                    for (uint32_t i : idx) ptrs.push_back(cand[i]);
                    CK(hipMemcpy(d_ptrs, ptrs.data(), ptrs.size() * sizeof(uint32_t*), hipMemcpyHostToDevice));
                    const uint32_t np = static_cast<uint32_t>(ptrs.size());
                    printf("{\"case\": \"access kinds on %s\", \"probes\": %u, \"exchange_ms\": %.3f, \"load_only_ms\": %.3f, \"store_only_ms\": %.3f, \"load_store_ms\": %.3f}\n", name, probes,
                           time_pieces_t<0>(d_ptrs, np, per_piece, nfrag, probes, sink), time_pieces_t<1>(d_ptrs, np, per_piece, nfrag, probes, sink),
                           time_pieces_t<2>(d_ptrs, np, per_piece, nfrag, probes, sink), time_pieces_t<3>(d_ptrs, np, per_piece, nfrag, probes, sink));
                };
                run4("16 pieces of one kind", one);
                run4("8 + 8 pieces of two kinds", two);
            }
        }
        // composed workspaces: 16 pieces of one kind, and 16 pieces drawn evenly from all kinds
        std::vector<std::vector<uint32_t>> members(refs.size() + 1);
        for (uint32_t k = 0; k < cand.size(); ++k) if (kind[k] >= 0) members[kind[k]].push_back(k);
        for (size_t kd = 0; kd < members.size(); ++kd) {
            if (members[kd].size() < parts) continue;
            std::vector<uint32_t> set(members[kd].begin(), members[kd].begin() + parts);
            printf("{\"case\": \"workspace of one kind\", \"kind\": %zu, \"ms_4096\": %.3f}\n", kd, probe(set, probes));
        }
        {
            std::vector<uint32_t> set;
            for (uint32_t i = 0; set.size() < parts; ++i)
                for (size_t kd = 1; kd < members.size() && set.size() < parts; ++kd)
                    if (i < members[kd].size()) set.push_back(members[kd][i]);
            printf("{\"case\": \"workspace drawn evenly from all kinds\", \"ms_4096\": %.3f}\n", probe(set, probes));
            for (size_t a = 1; a < members.size(); ++a)
                for (size_t b = a + 1; b < members.size(); ++b) {
                    std::vector<uint32_t> two;
                    for (uint32_t i = 0; i < parts / 2 && i < members[a].size() && i < members[b].size(); ++i) { two.push_back(members[a][i]); two.push_back(members[b][i]); }
                    if (two.size() == parts) printf("{\"case\": \"workspace half and half\", \"kinds\": [%zu, %zu], \"ms_4096\": %.3f}\n", a, b, probe(two, probes));
                }
        }
        return 0;
    }
    if (argc > 3 && argv[3][0] == 'm') {     // round 3: is the placement level a property of SMALL allocations too, and does a workspace of good pieces beat one allocation?
        const uint32_t parts = argc > 4 ? atoi(argv[4]) : 4, ncand = argc > 5 ? atoi(argv[5]) : 40;
        const uint32_t per_piece = nfrag / parts;
        const size_t piece_bytes = static_cast<size_t>(per_piece) * 65536;
        std::vector<uint32_t*> cand;
        for (uint32_t k = 0; k < ncand; ++k) {
            uint32_t* p = nullptr;
            if (hipMalloc(&p, piece_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            CK(hipMemsetAsync(p, 0, piece_bytes, 0));
            cand.push_back(p);
        }
        uint32_t** d_ptrs;
        CK(hipMalloc(&d_ptrs, 64 * sizeof(uint32_t*)));
        std::vector<std::pair<float, uint32_t>> score;
        for (uint32_t k = 0; k < cand.size(); ++k) {                      // each candidate alone, the whole grid folded onto it
            CK(hipMemcpy(d_ptrs, &cand[k], sizeof(uint32_t*), hipMemcpyHostToDevice));
            const float ms = time_pieces(d_ptrs, 1, per_piece, nfrag, 768, sink);
            score.push_back({ms, k});
            printf("{\"case\": \"candidate alone\", \"piece_GiB\": %.2f, \"index\": %u, \"ms\": %.3f, \"va\": \"%p\"}\n", piece_bytes / 1073741824.0, k, ms, (void*)cand[k]);
        }
        for (int pass = 1; pass < 4; ++pass)                              // spatial or temporal?  the same candidates again, in the same order
            for (uint32_t k = 0; k < cand.size(); ++k) {
                CK(hipMemcpy(d_ptrs, &cand[k], sizeof(uint32_t*), hipMemcpyHostToDevice));
                const float ms = time_pieces(d_ptrs, 1, per_piece, nfrag, 768, sink);
                printf("{\"case\": \"candidate alone, pass %d\", \"index\": %u, \"ms\": %.3f}\n", pass, k, ms);
            }
        // Model under test: a slow piece is UNBALANCED between two groups of some memory-side resource, in one of two directions; a pair of slow
        // pieces of opposite direction is fast.  Classify the slow ones against the first slow one, then compose sets by class.
        std::vector<uint32_t> balanced, same, opposite;
        {
            float lo = 1e9f, hi = 0;
            for (auto& sc : score) { lo = std::min(lo, sc.first); hi = std::max(hi, sc.first); }
            const float mid = 0.5f * (lo + hi);
            int ref = -1;
            for (uint32_t k = 0; k < cand.size(); ++k) {
                if (score[k].first < mid) { balanced.push_back(k); continue; }
                if (ref < 0) { ref = static_cast<int>(k); same.push_back(k); continue; }
                uint32_t* two[2] = {cand[ref], cand[k]};
                CK(hipMemcpy(d_ptrs, two, sizeof(two), hipMemcpyHostToDevice));
                const float ms = time_pieces(d_ptrs, 2, per_piece, nfrag, 768, sink);
                printf("{\"case\": \"slow pair\", \"ref\": %d, \"index\": %u, \"ms\": %.3f}\n", ref, k, ms);
                (ms < mid ? opposite : same).push_back(k);
            }
            printf("{\"case\": \"classes\", \"balanced\": %zu, \"same_as_ref\": %zu, \"opposite\": %zu, \"threshold_ms\": %.3f}\n", balanced.size(), same.size(),
                   opposite.size(), mid);
        }
        std::sort(score.begin(), score.end());
        auto run_set = [&](const char* name, const std::vector<uint32_t>& idx) {
            std::vector<uint32_t*> ptrs;
            for (uint32_t i : idx) ptrs.push_back(cand[i]);
            CK(hipMemcpy(d_ptrs, ptrs.data(), ptrs.size() * sizeof(uint32_t*), hipMemcpyHostToDevice));
            const float ms = time_pieces(d_ptrs, static_cast<uint32_t>(ptrs.size()), per_piece, nfrag, probes, sink);
            printf("{\"case\": \"%s\", \"pieces\": %zu, \"probes\": %u, \"ms\": %.3f, \"Gprobes_per_s\": %.2f}\n", name, ptrs.size(), probes, ms,
                   static_cast<double>(nfrag) * probes / ms / 1e6);
        };
        if (score.size() >= 2 * parts) {
            std::vector<uint32_t> best, worst, firsts;
            for (uint32_t i = 0; i < parts; ++i) { best.push_back(score[i].second); worst.push_back(score[score.size() - 1 - i].second); firsts.push_back(i); }
            run_set("workspace = the best pieces", best);
            run_set("workspace = the first pieces allocated", firsts);
            run_set("workspace = the worst pieces", worst);
            run_set("workspace = the best pieces (again)", best);
            run_set("workspace = the first pieces allocated (again)", firsts);
            run_set("workspace = the worst pieces (again)", worst);
            auto take = [&](const std::vector<uint32_t>& a, uint32_t na, const std::vector<uint32_t>& b, uint32_t nb) {
                std::vector<uint32_t> r;
                for (uint32_t i = 0; i < na && i < a.size(); ++i) r.push_back(a[i]);
                for (uint32_t i = 0; i < nb && i < b.size(); ++i) r.push_back(b[i]);
                return r;
            };
            if (balanced.size() >= parts) run_set("class: all balanced", take(balanced, parts, balanced, 0));
            if (same.size() >= parts) run_set("class: all same-as-ref", take(same, parts, same, 0));
            if (opposite.size() >= parts) run_set("class: all opposite", take(opposite, parts, opposite, 0));
            if (same.size() >= parts / 2 && opposite.size() >= parts / 2) run_set("class: half same, half opposite", take(same, parts / 2, opposite, parts / 2));
            if (same.size() >= parts / 4 && opposite.size() >= parts - parts / 4) run_set("class: quarter same, rest opposite", take(same, parts / 4, opposite, parts - parts / 4));
            if (balanced.size() >= parts / 2 && same.size() >= parts / 2) run_set("class: half balanced, half same", take(balanced, parts / 2, same, parts / 2));
        }
        for (uint32_t* p : cand) CK(hipFree(p));
        uint32_t* whole;                                                  // and ONE allocation of the whole size, three times
        for (int k = 0; k < 3; ++k) {
            CK(hipMalloc(&whole, static_cast<size_t>(nfrag) * 65536));
            CK(hipMemsetAsync(whole, 0, static_cast<size_t>(nfrag) * 65536, 0));
            CK(hipMemcpy(d_ptrs, &whole, sizeof(uint32_t*), hipMemcpyHostToDevice));
            const float ms = time_pieces(d_ptrs, 1, nfrag, nfrag, probes, sink);
            printf("{\"case\": \"one allocation #%d\", \"probes\": %u, \"ms\": %.3f, \"Gprobes_per_s\": %.2f, \"va\": \"%p\"}\n", k, probes, ms, static_cast<double>(nfrag) * probes / ms / 1e6, (void*)whole);
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
