// piece_search_model.cpp -- the library's workspace search (snappier_amd/csrc/piece_search.h, the code capi.hip runs) against a MODEL of
// device memory on the CPU.  The model is the one the microbenchmark measured: every candidate piece has a share of each of
// three kinds of memory; a probe's time is a + b x (largest kind's share of the probed set), with (a, b) fitted to the measured levels
// (alone 3.58 / 3.93 ms, pairs 3.67 / 4.33, sixteen pieces 3.79 / 3.99 / 4.75), plus 0.3 % noise.  Prints one JSON line per scenario;
// tests/test_piece_search_model.py asserts on them.
#include <array>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../snappier_amd/csrc/piece_search.h"

using snp_piece_search::PieceSearch;
typedef std::array<float, 3> Shares;

static uint32_t g_rng = 1;
static float g_contrast = 1.0f;   // scales the kind-dependent part of every probe: a batch size at which one kind and two kinds are closer together
static float noise() { g_rng = g_rng * 1664525u + 1013904223u; return 1.0f + (static_cast<float>(g_rng >> 8) / 16777216.0f - 0.5f) * 0.006f; }

struct Memory {
    std::vector<Shares> layout;     // what the driver would hand out, in allocation order
    size_t fail_after;              // hipMalloc fails from this candidate on
    std::vector<Shares> got;
    bool alloc() { if (got.size() >= fail_after || got.size() >= layout.size()) return false; got.push_back(layout[got.size()]); return true; }
    float largest(const std::vector<uint32_t>& set) const
    {
        Shares s{0, 0, 0};
        for (uint32_t k : set) for (int d = 0; d < 3; ++d) s[d] += got[k][d];
        return std::max(s[0], std::max(s[1], s[2])) / static_cast<float>(set.size());
    }
    float probe(const std::vector<uint32_t>& set) const
    {
        const float m = largest(set);
        const float a = set.size() == 2 ? 3.01f : 3.23f, b = set.size() == 1 ? 0.70f : set.size() == 2 ? 1.32f : 1.52f;
        return (a + b * g_contrast * (m - 0.5f) + b * 0.5f) * noise();   // (contrast 1: a + b x m; the balanced level stays where it is)
    }
};

static std::vector<Shares> runs(const std::vector<std::pair<int, int>>& spec)   // (kind or -1 = a piece straddling the neighbours, count)
{
    std::vector<Shares> v;
    for (size_t i = 0; i < spec.size(); ++i)
        for (int c = 0; c < spec[i].second; ++c) {
            Shares s{0, 0, 0};
            if (spec[i].first >= 0) s[spec[i].first] = 1.f;
            else { s[spec[i - 1].first] = 0.5f; s[spec[i + 1].first] = 0.5f; }
            v.push_back(s);
        }
    return v;
}

static void scenario(const char* name, const std::vector<Shares>& layout, size_t max_cand, size_t fail_after = ~size_t(0), size_t patience = 0)
{
    Memory mem{layout, fail_after, {}};
    PieceSearch ps{};
    ps.n = 16;
    ps.max_cand = max_cand;
    ps.piece_gib = 0.625;
    ps.dbg = getenv("SNAPPIER_HIP_DEBUG") != nullptr;
    if (patience) ps.patience = patience;
    ps.alloc_one = [&]() { return mem.alloc(); };
    ps.probe_set = [&](const std::vector<uint32_t>& set) { return mem.probe(set); };
    std::vector<uint32_t> set;
    const float ms = ps.run(set);
    bool distinct = true;
    for (size_t i = 0; i < set.size(); ++i) for (size_t j = i + 1; j < set.size(); ++j) distinct = distinct && set[i] != set[j];
    bool in_range = true;
    for (uint32_t k : set) in_range = in_range && k < mem.got.size();
    printf("{\"scenario\": \"%s\", \"ms\": %.3f, \"candidates\": %zu, \"chosen\": %zu, \"distinct\": %s, \"in_range\": %s, \"largest_share\": %.3f, \"first_16_largest_share\": %.3f, "
           "\"references\": %zu, \"probes\": %u}\n", name, ms, mem.got.size(), set.size(), distinct ? "true" : "false", in_range ? "true" : "false",
           ms >= 0 && set.size() == 16 && in_range ? mem.largest(set) : -1.f,
           mem.got.size() >= 16 ? [&] { std::vector<uint32_t> f; for (uint32_t i = 0; i < 16; ++i) f.push_back(i); return mem.largest(f); }() : -1.f,
           ps.refs.size(), ps.trials);
}

int main()
{
    // a fresh process (profiles/r03y_microbench_memory_kinds.jsonl): short runs of three kinds with straddling pieces in between
    scenario("three kinds early", runs({{0, 2}, {-1, 1}, {1, 9}, {-1, 2}, {2, 14}, {-1, 1}, {0, 30}, {1, 200}}), 256);
    // two kinds within the first rounds, the third far away
    scenario("two kinds early", runs({{0, 20}, {-1, 1}, {1, 120}, {2, 100}}), 256);
    // ... the same memory searched by snp_ctx_reserve_compress (the caller has time: patience as far as max_cand allows): the third kind is found
    scenario("two kinds early, thorough", runs({{0, 20}, {-1, 1}, {1, 120}, {2, 100}}), 256, ~size_t(0), 64);
    // a process whose first hundred candidates are of one kind (seen once: profiles/r03y_piece_search_compress.txt)
    scenario("second kind after 110 candidates", runs({{0, 110}, {-1, 1}, {1, 100}}), 212);
    // one kind only, as far as the search may go
    scenario("one kind only", runs({{0, 300}}), 64);
    // another batch size: one kind and two kinds only 10 % apart instead of 18 %
    g_contrast = 0.55f;
    scenario("two kinds early, weak contrast", runs({{0, 20}, {-1, 1}, {1, 120}, {2, 100}}), 256);
    g_contrast = 1.0f;
    // everything straddles (balanced pieces alone)
    scenario("all pieces balanced", runs({{0, 1}, {-1, 60}, {1, 1}}), 256);
    // a process whose first rounds are mostly MIXED pieces (each lies across two kinds, 60 : 40 or so -- the "F" pieces of the memory map, a third of device
    // memory): no reference explains them, but they are not kinds of their own.  Seen on the GPU in round 4 (profiles/r04ae_search_mixed_first_round.txt):
    // a search that made references of them stopped after 16-32 candidates with "4 references, largest share 0.30" and a workspace 6 % slower than
    // the one a longer search finds.
    {
        std::vector<Shares> v;
        const Shares mixes[6] = {{0.6f, 0.4f, 0.f}, {0.4f, 0.6f, 0.f}, {0.65f, 0.35f, 0.f}, {0.35f, 0.65f, 0.f}, {0.55f, 0.45f, 0.f}, {0.45f, 0.55f, 0.f}};
        v.push_back(Shares{1.f, 0.f, 0.f});
        v.push_back(Shares{1.f, 0.f, 0.f});
        for (int i = 0; i < 30; ++i) v.push_back(mixes[i % 6]);
        for (const Shares& x : runs({{1, 60}, {-1, 2}, {0, 30}, {-1, 2}, {2, 100}})) v.push_back(x);
        scenario("mixed pieces first", v, 216, ~size_t(0), 64);
    }
    // no room for spare candidates: the workspace is what could be allocated
    scenario("no spare candidates", runs({{0, 300}}), 16);
    // out of memory: before the workspace is complete / after one round
    scenario("out of memory at 10", runs({{0, 8}, {1, 100}}), 256, 10);
    scenario("out of memory at 20", runs({{0, 8}, {1, 100}}), 256, 20);
    return 0;
}
