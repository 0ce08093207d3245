// frame_scan.hip -- chunk-header walk of a framed stream that is already in HBM and has no chunk table
// (SnappyStreamDecompressor.ReadChunkHeader / Decompress, Snappier/Internal/SnappyStreamDecompressor.cs:53-199,215-289).
//
// A framed stream carries no index: header i gives the position of header i+1, a serial chain of ~64 KiB hops, each a
// dependent HBM round trip (round 1: one lane, 0.9 us per chunk -- 146 ms of the 176 ms a 10 GiB stream took to decode).
// Here the chain is broken into 1 MiB SPANS that are walked concurrently:
//   A  k_span_candidates  one wavefront per span: test every byte of the span's first 80 KiB for "a data chunk or a stream
//                         identifier could start here" (type, 24-bit size, preamble varint, expansion bound, fits the stream:
//                         passes ~1e-5 of random positions, and always the true one if the chunk before it is a spec-sized
//                         data chunk), then lane c follows candidate c's chain to the end of the span: exit position, data
//                         chunks met, bytes they declare, the error that ended it if any;
//   B  k_span_resolve     one wavefront: the true chain enters span 0 at byte 0; its exit selects the candidate of the next
//                         span that starts exactly there, and so on -- 64 spans per batch out of registers.  An entry that
//                         is no candidate (behind a skippable chunk, a chunk larger than the window) is walked on the spot;
//   C  k_span_emit        one wavefront per span: walk the span again from its true entry and write the chunk table rows at
//                         their final indices; entries past the last chunk become empty chunks (the decode and CRC launches
//                         run over max_chunks rows without a host round trip).
// The rules of one hop (frame_hop) are those of the host walk in capi_frame.hip (scan_chunks); the table is what it produces.
#include "snp_device.h"

namespace {

constexpr u64 kSpan = 1ull << 20;
constexpr u32 kWindow = 80 * 1024;          // > 8 + MaxCompressedLength(65536): the next header after any spec-sized data chunk
constexpr u32 kMaxCand = 4;                 // candidates kept per span (the lowest plausible positions of its window)
constexpr u32 kEmptyMaskedCrcS = 0xa282ead8u;   // crc32c_mask(crc32c of no bytes)
constexpr u64 kNoEntry = ~0ull;

enum HopKind : u32 { HOP_DATA = 0, HOP_SKIP = 1, HOP_END = 2, HOP_ERR = 3 };

struct Hop {
    u32 kind;
    i32 err;        // HOP_ERR: the status that ends the walk
    u32 type;       // HOP_DATA: 0 compressed, 1 uncompressed
    u32 body_len, crc, dec;
    u64 next;       // position of the next header
};

// One header at ip (< n or == n).  Same rules, in the same order, as scan_chunks (capi_frame.hip) / the reference reader.
__device__ __forceinline__ Hop frame_hop(const u8* __restrict__ in, u64 n, u64 ip)
{
    Hop h{};
    h.next = ip;
    if (ip >= n) { h.kind = HOP_END; return h; }
    if (n - ip < 4) { h.kind = HOP_ERR; h.err = SNP_ERR_TRUNCATED_STREAM; return h; }
    u32 b[4] = {0, 0, 0, 0};                                            // 16 bytes at ip (fewer at the very end)
    if (n - ip >= 16) {
        const snp_u128_unaligned q = *reinterpret_cast<const snp_u128_unaligned*>(in + ip);
        b[0] = q.v[0]; b[1] = q.v[1]; b[2] = q.v[2]; b[3] = q.v[3];
    } else {
        for (u32 i = 0; i < static_cast<u32>(n - ip); ++i) b[i >> 2] |= static_cast<u32>(in[ip + i]) << (8 * (i & 3));
    }
    const u32 t = b[0] & 0xffu;
    const u32 size = b[0] >> 8;                                         // :64-65
    if (n - (ip + 4) < size) { h.kind = HOP_ERR; h.err = SNP_ERR_TRUNCATED_STREAM; return h; }
    h.next = ip + 4 + size;
    if (t <= 1) {
        if (size < 4) { h.kind = HOP_ERR; h.err = SNP_ERR_TRUNCATED_STREAM; return h; }
        u32 dec = size - 4;
        if (t == 0) {                                                   // block preamble  VarIntEncoding.Read.cs:38-79
            const u64 pre = b[2] | (static_cast<u64>(b[3]) << 32);
            const u32 avail = size - 4 < 5 ? size - 4 : 5;
            u32 result = 0, shift = 0;
            bool done = false, bad = false;
            for (u32 i = 0; i < avail && !done && !bad; ++i) {
                const u32 c = static_cast<u32>(pre >> (8 * i)) & 0xffu;
                const u32 val = c & 0x7fu;
                if (val & ~(0xffffffffu >> shift)) { bad = true; break; }
                result |= val << shift;
                shift += 7;
                if (c < 128) done = true;
            }
            if (bad || !done || result > 0x7fffffffu) { h.kind = HOP_ERR; h.err = SNP_ERR_BAD_LENGTH; return h; }
            dec = result;
            // no tag expands more than 3 bytes -> 64: such a chunk can only end "Incomplete Snappy block." (capi_frame.hip scan_chunks)
            if (static_cast<u64>(dec) > (static_cast<u64>(size - 4 - (shift / 7)) / 3 + 1) * 64) { h.kind = HOP_ERR; h.err = SNP_ERR_INCOMPLETE; return h; }
        }
        h.kind = HOP_DATA;
        h.type = t;
        h.body_len = size - 4;
        h.crc = b[1];                                                   // ReadChunkCrc  :260-289
        h.dec = dec;
        return h;
    }
    if (t < 0x80) { h.kind = HOP_ERR; h.err = SNP_ERR_CHUNK_TYPE; return h; }   // :182-185
    h.kind = HOP_SKIP;                                                  // 0x80..0xff skipped unvalidated  :187-196
    return h;
}

// "Could the true chain enter here?"  Only shapes a spec-conforming writer emits are candidates (a data chunk of at most
// 65536 raw bytes, or the stream identifier); everything else still DECODES -- it just is not guessed, the resolver walks it.
// Compressed payload is full of bytes that look like a raw-chunk header (0x01 is the commonest copy tag, followed by small
// numbers: ~3e-4 of positions), so a candidate must also be FOLLOWED by such a shape, or end the stream: ~1e-8.
__device__ __forceinline__ bool chunk_shape(const u8* __restrict__ in, u64 n, u64 p, u64* next)
{
    if (n - p < 8) return false;
    const u32 w0 = ld32u(in + p);
    const u32 t = w0 & 0xffu, size = w0 >> 8;
    *next = p + 4 + size;
    if (t == 0xffu) return size == 6 && n - p >= 10 && ld32u(in + p + 4) == 0x50614e73u && in[p + 8] == 0x70 && in[p + 9] == 0x59;
    if (t > 1) return false;
    if (n - (p + 4) < size) return false;
    if (t == 1) return size >= 4 && size <= 65536 + 4;
    if (size < 5 || size > 76496 + 4) return false;
    const Hop h = frame_hop(in, n, p);
    return h.kind == HOP_DATA && h.dec <= 65536;
}
__device__ __forceinline__ bool plausible_start(const u8* __restrict__ in, u64 n, u64 p)
{
    u64 next = 0, next2 = 0;
    if (!chunk_shape(in, n, p, &next)) return false;
    return next == n || chunk_shape(in, n, next, &next2);
}

// What following a chain from `start` to the end of its span yields.
struct Chain {
    u64 exit;       // position of the first header at or beyond the span's end (or where the chain stopped)
    u64 dec;        // bytes declared by the data chunks met
    u32 ndata;      // data chunks met
    i32 stop;       // 0: left the span; -1: clean end of stream; > 0: the status that ended it (chunks before it still count)
};

__device__ __forceinline__ Chain follow_chain(const u8* __restrict__ in, u64 n, u64 start, u64 span_end)
{
    Chain c{start, 0, 0, 0};
    u64 ip = start;
    while (ip < span_end) {
        const Hop h = frame_hop(in, n, ip);
        if (h.kind == HOP_END) { c.stop = -1; break; }
        if (h.kind == HOP_ERR) { c.stop = h.err; break; }
        if (h.kind == HOP_DATA) { ++c.ndata; c.dec += h.dec; }
        ip = h.next;
    }
    if (c.stop == 0 && ip >= n) c.stop = ip == n ? -1 : 0;   // ip > n cannot happen (a body never runs past n)
    c.exit = ip;
    return c;
}

// per-span record, structure of arrays over [nspans][kMaxCand]; count[span] = candidates kept (the lowest positions)
struct SpanTables {
    u32* count;
    u32* start_rel;     // start - span * kSpan
    u64* exit;
    u64* dec;
    u32* ndata;
    i32* stop;
    // resolver -> emitter
    u64* entry;         // true entry position of each span, kNoEntry if the chain never starts a header inside it
    u32* chunk_base;    // index of the span's first data chunk
    u64* out_base;      // decoded bytes before the span
};

__device__ __host__ inline SpanTables span_tables(void* base, u64 nspans)
{
    u8* p = static_cast<u8*>(base);
    SpanTables t;
    auto take = [&](u64 bytes) { u8* r = p; p += (bytes + 15) / 16 * 16; return r; };
    t.count = reinterpret_cast<u32*>(take(nspans * 4));
    t.start_rel = reinterpret_cast<u32*>(take(nspans * kMaxCand * 4));
    t.exit = reinterpret_cast<u64*>(take(nspans * kMaxCand * 8));
    t.dec = reinterpret_cast<u64*>(take(nspans * kMaxCand * 8));
    t.ndata = reinterpret_cast<u32*>(take(nspans * kMaxCand * 4));
    t.stop = reinterpret_cast<i32*>(take(nspans * kMaxCand * 4));
    t.entry = reinterpret_cast<u64*>(take(nspans * 8));
    t.chunk_base = reinterpret_cast<u32*>(take(nspans * 4));
    t.out_base = reinterpret_cast<u64*>(take(nspans * 8));
    return t;
}

// ---- A: candidates of every span and where their chains lead ----------------------------------------------------------
__global__ __launch_bounds__(SNP_WAVE) void k_span_candidates(const u8* __restrict__ in, u64 n, u64 nspans, void* work)
{
    __shared__ u32 s_cand[kMaxCand + 1];
    __shared__ u32 s_n;
    const u64 k = blockIdx.x;
    if (k >= nspans) return;
    const SpanTables t = span_tables(work, nspans);
    const u32 lane = lane_id();
    const u64 s0 = k * kSpan;
    const u64 s1 = s0 + kSpan < n ? s0 + kSpan : n;
    if (lane == 0) s_n = 0;
    __syncthreads();
    if (k == 0) {
        if (lane == 0) { s_cand[0] = 0; s_n = 1; }                      // a stream starts at byte 0, whatever is there
    } else {
        // the kMaxCand LOWEST plausible positions: the true entry is the first true header of the span, and fewer than
        // kMaxCand false positives precede it except in adversarial payloads (then the resolver walks the span itself)
        const u64 wend = s0 + kWindow < s1 ? s0 + kWindow : s1;
        for (u64 base = s0; base < wend; base += SNP_WAVE) {
            const u64 p = base + lane;
            const bool ok = p < wend && plausible_start(in, n, p);
            const u64 m = ballot64(ok);
            if (m) {
                const u32 have = s_n;
                if (ok) {
                    const u32 idx = have + static_cast<u32>(__builtin_popcountll(m & lanes_below(lane)));
                    if (idx < kMaxCand) s_cand[idx] = static_cast<u32>(p - s0);
                }
                __syncthreads();
                if (lane == 0) { const u32 tot = have + static_cast<u32>(__builtin_popcountll(m)); s_n = tot < kMaxCand ? tot : kMaxCand; }
                __syncthreads();
                if (s_n == kMaxCand) break;
            }
        }
    }
    __syncthreads();
    const u32 nc = s_n;
    if (lane == 0) t.count[k] = nc;
    if (lane < nc) {
        const u64 start = s0 + s_cand[lane];
        const Chain c = follow_chain(in, n, start, s0 + kSpan);
        const u64 i = k * kMaxCand + lane;
        t.start_rel[i] = s_cand[lane];
        t.exit[i] = c.exit;
        t.dec[i] = c.dec;
        t.ndata[i] = c.ndata;
        t.stop[i] = c.stop;
    }
}

// ---- B: the true chain through the spans ------------------------------------------------------------------------------
// hdr: [0] total decoded bytes, [1] tail status, [2] data chunks listed.  Lane l of a batch holds the candidates of span
// base + l; the chain is followed with readlane, so a hop from span to span costs ~20 scalar instructions, no memory.
__global__ __launch_bounds__(SNP_WAVE) void k_span_resolve(const u8* __restrict__ in, u64 n, u64 cap, u32 max_chunks, u64 nspans,
                                                          void* work, u64* __restrict__ hdr)
{
    const SpanTables t = span_tables(work, nspans);
    const u32 lane = lane_id();
    for (u64 k = lane; k < nspans; k += SNP_WAVE) t.entry[k] = kNoEntry;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    u64 e = 0, total = 0;
    u32 nc = 0;
    i32 tail = SNP_OK;
    const bool stop = n == 0;
    u64 batch0 = ~0ull;                                                 // first span of the batch held in registers
    u32 cnt = 0, srel[kMaxCand] = {}, cnd[kMaxCand] = {};
    i32 cst[kMaxCand] = {};
    u64 cex[kMaxCand] = {}, cde[kMaxCand] = {};
    while (!stop) {
        if (e >= n) break;                                              // clean end (n > 0: at least one header was walked)
        const u64 k = e / kSpan;
        if (k < batch0 || k >= batch0 + SNP_WAVE) {                     // load the candidates of 64 spans
            batch0 = k;
            const u64 mine = batch0 + lane;
            cnt = mine < nspans ? t.count[mine] : 0;
#pragma unroll
            for (u32 j = 0; j < kMaxCand; ++j) {
                const bool have = mine < nspans && j < cnt;
                srel[j] = have ? t.start_rel[mine * kMaxCand + j] : 0xffffffffu;
                cex[j] = have ? t.exit[mine * kMaxCand + j] : 0;
                cde[j] = have ? t.dec[mine * kMaxCand + j] : 0;
                cnd[j] = have ? t.ndata[mine * kMaxCand + j] : 0;
                cst[j] = have ? t.stop[mine * kMaxCand + j] : 0;
            }
        }
        const u32 l = static_cast<u32>(k - batch0);
        const u32 rel = static_cast<u32>(e - k * kSpan);
        Chain c{};
        bool found = false;
#pragma unroll
        for (u32 j = 0; j < kMaxCand; ++j) {
            if (!found && read_lane(srel[j], l) == rel) {
                found = true;
                c.exit = (static_cast<u64>(read_lane(static_cast<u32>(cex[j] >> 32), l)) << 32) | read_lane(static_cast<u32>(cex[j]), l);
                c.dec = (static_cast<u64>(read_lane(static_cast<u32>(cde[j] >> 32), l)) << 32) | read_lane(static_cast<u32>(cde[j]), l);
                c.ndata = read_lane(cnd[j], l);
                c.stop = static_cast<i32>(read_lane(static_cast<u32>(cst[j]), l));
            }
        }
        if (!found) c = follow_chain(in, n, e, (k + 1) * kSpan);        // not guessed: walk this span here
        if (nc + c.ndata > max_chunks) {
            // the chunk table fills up inside this span: list what fits, exactly as a serial walk would
            if (lane == 0) { t.entry[k] = e; t.chunk_base[k] = nc; t.out_base[k] = total; }   // the emitter clips its rows at hdr[2]
            u64 ip = e;
            for (;;) {
                const Hop h = frame_hop(in, n, ip);
                if (h.kind == HOP_END) break;
                if (h.kind == HOP_ERR) { tail = h.err; break; }
                if (h.kind == HOP_DATA) {
                    if (nc == max_chunks) { tail = SNP_ERR_OUTPUT_TOO_SMALL; break; }   // chunk table full
                    ++nc;
                    total += h.dec;
                }
                ip = h.next;
            }
            break;
        }
        if (lane == 0) { t.entry[k] = e; t.chunk_base[k] = nc; t.out_base[k] = total; }
        nc += c.ndata;
        total += c.dec;
        if (c.stop > 0) { tail = c.stop; break; }
        if (c.stop < 0) break;
        e = c.exit;
    }
    if (total > cap) { tail = SNP_ERR_OUTPUT_TOO_SMALL; nc = 0; total = 0; }   // nothing is decoded
    if (lane == 0) {
        hdr[0] = total;
        hdr[1] = static_cast<u64>(static_cast<u32>(tail));
        hdr[2] = nc;
    }
}

// ---- C: the chunk table ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(SNP_WAVE) void k_span_emit(const u8* __restrict__ in, u64 n, u32 max_chunks, u64 nspans, void* work,
                                                       const u64* __restrict__ hdr, u8* __restrict__ type,
                                                       u64* __restrict__ body_off, u32* __restrict__ body_len,
                                                       u32* __restrict__ crc, u64* __restrict__ out_off, u32* __restrict__ out_cap)
{
    const u64 k = blockIdx.x;
    const u32 lane = lane_id();
    const u32 nc_total = static_cast<u32>(hdr[2]);
    if (k >= nspans) {                                                  // the extra workgroups pad the table with empty chunks
        const u64 total = hdr[0];
        for (u64 r = nc_total + (k - nspans) * SNP_WAVE + lane; r < max_chunks; r += (gridDim.x - nspans) * SNP_WAVE) {
            type[r] = 1;
            body_off[r] = 0;
            body_len[r] = 0;
            crc[r] = kEmptyMaskedCrcS;
            out_off[r] = total;
            out_cap[r] = 0;
        }
        return;
    }
    const SpanTables t = span_tables(work, nspans);
    const u64 e = t.entry[k];
    if (e == kNoEntry || lane != 0) return;
    u32 idx = t.chunk_base[k];
    u64 off = t.out_base[k];
    u64 ip = e;
    const u64 s1 = (k + 1) * kSpan;
    while (ip < s1 && idx < nc_total) {
        const Hop h = frame_hop(in, n, ip);
        if (h.kind == HOP_END || h.kind == HOP_ERR) break;
        if (h.kind == HOP_DATA) {
            type[idx] = static_cast<u8>(h.type);
            body_off[idx] = ip + 8;
            body_len[idx] = h.body_len;
            crc[idx] = h.crc;
            out_off[idx] = off;
            out_cap[idx] = h.dec;
            off += h.dec;
            ++idx;
        }
        ip = h.next;
    }
}

}  // namespace

extern "C" size_t snp_frame_scan_workspace(u64 n)
{
    const u64 nspans = (n + kSpan - 1) / kSpan;
    const SpanTables t = span_tables(nullptr, nspans ? nspans : 1);
    return reinterpret_cast<size_t>(t.out_base) + (nspans ? nspans : 1) * 8 + 64;
}

extern "C" hipError_t snp_launch_frame_scan_spans(const u8* in, u64 n, u64 cap, u32 max_chunks, u8* type, u64* body_off,
                                                  u32* body_len, u32* crc, u64* out_off, u32* out_cap, u64* hdr, void* work,
                                                  hipStream_t stream)
{
    const u64 nspans = (n + kSpan - 1) / kSpan;
    if (nspans) hipLaunchKernelGGL(k_span_candidates, dim3(static_cast<u32>(nspans)), dim3(SNP_WAVE), 0, stream, in, n, nspans, work);
    hipLaunchKernelGGL(k_span_resolve, dim3(1), dim3(SNP_WAVE), 0, stream, in, n, cap, max_chunks, nspans, work, hdr);
    const u32 pad = max_chunks ? 64u : 1u;
    hipLaunchKernelGGL(k_span_emit, dim3(static_cast<u32>(nspans) + pad), dim3(SNP_WAVE), 0, stream, in, n, max_chunks, nspans, work,
                       hdr, type, body_off, body_len, crc, out_off, out_cap);
    return hipGetLastError();
}
