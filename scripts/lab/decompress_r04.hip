// lab/decompress_r04.hip -- the round-4 block decoder with every front end that was measured and lost (queued, batched, the round-4 sub-chain
// form with its A/B switches, the LDS ring, the producer/consumer pair).  NOT part of the product: it replaces csrc/decompress.hip only in
// libraries built by `LAB=1 scripts/build_variant.sh <name>` (snappier_amd/variants/), which the variant-parametrised tests and the A/B
// scripts load.  The default decoder is csrc/decode_chains.hip in both builds.
//
// Snappy block decompression, one 64 KiB block per wavefront (gfx950).
//
// Replaces the tag loop of SnappyDecompressor.DecompressAllTags + Append / AppendFromSelf
// (Snappier/Internal/SnappyDecompressor.cs:184-347,568-611; copy semantics CopyHelpers.cs:222-230) for whole blocks.
//
// Layout / data movement per wavefront:
//   * compressed input: a 512-byte sliding window lives in two VGPRs per lane (aligned dword loads, 256 B per
//     coalesced wave load, the next 256 B always in flight); tag bytes are pulled out with v_readlane so the
//     whole tag decode runs on the scalar unit -- no memory round trip per tag;
//   * literals: lanes copy straight from the input stream to the output (1 byte/lane up to 64 B, 16 B/lane for
//     long literals);
//   * copies (len <= 64 per tag): lane k reads out[op - off + (k mod off)] and writes out[op + k]; a pattern copy
//     (off < len) is resolved arithmetically, so the source never overlaps the bytes this instruction writes;
//   * back-references read the block's own earlier output through L1/L2 (a 64 KiB block stays cache resident);
//     HBM sees the algorithmic bytes only: C read + U written.
// Vector memory operations of one wave are issued and serviced in order, so a later load observes an earlier store
// of the same wave (FENCED = true additionally drains vmcnt when a source range is younger than the last drain).
//
// FRONT = 1 puts a token-parallel front end before that serial loop (the scalar unit -- one instruction per cycle per
// CU -- is what bounds the serial loop: ~65 SALU instructions per tag):
//   1. all 64 lanes decode "the tag that would start at input byte ip + lane" (one unaligned 8-byte load each);
//   2. the true tag starts are picked out by walking next-pointers (1, 2, 3, 4 hops precomputed with ds_bpermute),
//      four tags per scalar step;
//   3. a DPP prefix sum of the output lengths gives every tag its output offset;
//   4. tags execute one per LANE with wide unaligned copies (16/8/4/2/1 B), in dependency rounds: a copy is ready
//      once its source lies below the watermark of completed output; literals are always ready.  Pattern copies
//      (offset < length) and literals > 64 B are done cooperatively by the whole wave.
//   Anything irregular (an error, a tag or literal running past the input, the last < 72 input bytes) leaves the
//   batch untouched and falls through to the serial loop, which owns the exact error semantics.
// FRONT = 2 parses windows the same way but appends their tags to a queue in LDS and executes 64 tags at a time, so
// every vector-memory instruction is issued with all its lanes busy (a 64-byte window holds only ~21 tags).
// FRONT = 3 (k_decompress_chains, the default for whole blocks) finds the tag starts of 2 KiB of input at once -- every
// lane walks a chain of tags through its own 32 bytes, chains that meet are the same chain from there on -- and executes
// 64 tags at a time straight from the resulting list (see the block comment at `if (FRONT == 3)`).
// FRAG = true decodes one 64 KiB fragment of a larger block from a tag start found by tag_index.hip (FRONT 0 or 2).
#include "snp_device.h"

namespace {

struct InWindow {
    const u8* a0;   // block start rounded down to a dword boundary
    const u8* end;  // one past the last compressed byte
    u32 wv;         // window start, bytes from a0, multiple of 256
    u32 lo, hi;     // this lane's dwords at a0 + wv + 4*lane and a0 + wv + 256 + 4*lane
};

// Aligned dword that contains at least one valid byte: never crosses a page, so it cannot fault.
__device__ __forceinline__ u32 win_load(const InWindow& w, u32 voff)
{
    const u8* p = w.a0 + voff;
    return p < w.end ? *reinterpret_cast<const u32*>(p) : 0u;
}

// 8 bytes at virtual offset v (wave-uniform), served from the register window.
__device__ __forceinline__ u64 win_fetch(InWindow& w, u32 v, u32 lane)
{
    u32 rel = v - w.wv;
    if (rel >= 512) {                       // jumped over a long literal: re-seat the window
        w.wv = v & ~255u;
        w.lo = win_load(w, w.wv + 4 * lane);
        w.hi = win_load(w, w.wv + 256 + 4 * lane);
        rel = v - w.wv;
    } else if (rel >= 256) {                // slide: the prefetched half becomes current, fetch the next
        w.lo = w.hi;
        w.wv += 256;
        w.hi = win_load(w, w.wv + 256 + 4 * lane);
        rel -= 256;
    }
    const u32 idx = rel >> 2;
    const u32 d0 = read_lane(w.lo, idx);
    const u32 d1 = idx < 63 ? read_lane(w.lo, (idx + 1) & 63) : read_lane(w.hi, 0);
    const u64 q = (static_cast<u64>(d1) << 32) | d0;
    return q >> ((v & 3u) * 8u);            // >= 5 valid bytes
}

// Inclusive prefix sum across the 64 lanes with DPP row shifts / row broadcasts (no LDS, no bpermute).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_or_zero(u32 v)
{
    return static_cast<u32>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ u32 wave_inclusive_scan(u32 x)
{
    u32 y = x + dpp_or_zero<0x111, 0xf>(x);            // row_shr:1
    y += dpp_or_zero<0x112, 0xf>(x);                   // row_shr:2
    y += dpp_or_zero<0x113, 0xf>(x);                   // row_shr:3   -> sums of 4 within a row of 16
    y += dpp_or_zero<0x114, 0xf>(y);                   // row_shr:4   -> 8
    y += dpp_or_zero<0x118, 0xf>(y);                   // row_shr:8   -> 16 (whole row)
    y += dpp_or_zero<0x142, 0xa>(y);                   // row_bcast:15 into rows 1 and 3
    y += dpp_or_zero<0x143, 0xc>(y);                   // row_bcast:31 into rows 2 and 3
    return y;
}

__device__ __forceinline__ u32 bperm(u32 src_lane, u32 v)
{
    return static_cast<u32>(__builtin_amdgcn_ds_bpermute(static_cast<int>(src_lane << 2), static_cast<int>(v)));
}

struct __attribute__((packed)) snp_u16_unaligned { u16 v; };

// One lane copies len (1..64) bytes from s to d (non-overlapping).  Stores are exact; LOADS are not: every lane reads
// 16 bytes at s (and, above 16, the 16 bytes ending exactly at len), so one memory round trip serves every size class
// and the 8/4/2/1-byte stores of a short copy are cut out of the registers.  The caller guarantees that reading up to 15
// bytes past the source is safe.  Copies longer than 32 bytes take one or two extra 16-byte pieces in the middle.
__device__ __forceinline__ void lane_copy(u8* d, const u8* s, u32 len)
{
    const snp_u128_unaligned p0 = *reinterpret_cast<const snp_u128_unaligned*>(s);
    if (len >= 16) {
        snp_u128_unaligned p1 = p0;
        if (len > 16) p1 = *reinterpret_cast<const snp_u128_unaligned*>(s + len - 16);
        *reinterpret_cast<snp_u128_unaligned*>(d) = p0;
        if (len > 16) *reinterpret_cast<snp_u128_unaligned*>(d + len - 16) = p1;
        if (len > 32) {
            *reinterpret_cast<snp_u128_unaligned*>(d + 16) = *reinterpret_cast<const snp_u128_unaligned*>(s + 16);
            if (len > 48) *reinterpret_cast<snp_u128_unaligned*>(d + 32) = *reinterpret_cast<const snp_u128_unaligned*>(s + 32);
        }
    } else {
        // 8 / 4 / 2 / 1-byte pieces, each cut from the front of what is left of the 16 bytes (a running "shift")
        const bool c8 = (len & 8u) != 0, c4 = (len & 4u) != 0, c2 = (len & 2u) != 0;
        const u32 a0 = c8 ? p0.v[2] : p0.v[0];                          // after the 8-byte piece
        const u32 a1 = c8 ? p0.v[3] : p0.v[1];
        const u32 b0 = c4 ? a1 : a0;                                    // after the 4-byte piece
        const u32 c0 = c2 ? b0 >> 16 : b0;                              // after the 2-byte piece
        const u32 o4 = len & 8u, o2 = len & 12u, o1 = len & 14u;
        if (c8) reinterpret_cast<snp_u64_unaligned*>(d)->v = p0.v[0] | (static_cast<u64>(p0.v[1]) << 32);
        if (c4) st32u(d + o4, a0);
        if (c2) reinterpret_cast<snp_u16_unaligned*>(d + o2)->v = static_cast<u16>(b0);
        if (len & 1u) d[o1] = static_cast<u8>(c0);
    }
}

// lane_copy in two halves, so that a pass first REQUESTS every piece of every lane's copy (one memory round trip for the whole
// pass: lane_copy's ladder waits for one piece before it asks for the next, up to four round trips for a 64-byte tag) and
// then stores them.  Same over-read contract as lane_copy.  (Plain vector values, not a struct behind a reference: that
// one ended up in scratch memory.)
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld128u(const u8* p)
{
    const snp_u128_unaligned t = *reinterpret_cast<const snp_u128_unaligned*>(p);
    return u32x4{t.v[0], t.v[1], t.v[2], t.v[3]};
}
__device__ __forceinline__ void st128u(u8* p, u32x4 v)
{
    snp_u128_unaligned t;
    t.v[0] = v.x; t.v[1] = v.y; t.v[2] = v.z; t.v[3] = v.w;
    *reinterpret_cast<snp_u128_unaligned*>(p) = t;
}
// One lane copies len (1..64) bytes, requesting its pieces in pairs: head + tail first (every copy of <= 32 bytes is one round
// trip), then the two middle pieces of a longer one.
__device__ __forceinline__ void lane_copy2(u8* d, const u8* s, u32 len)
{
    const u32x4 p0 = ld128u(s);
    u32x4 p1;                                                           // read only where it was loaded
    if (len > 16) p1 = ld128u(s + len - 16);
#if SNP_D_PIECES == 4
    u32x4 p2, p3;                                                       // all four pieces in flight together (8 more VGPRs)
    if (len > 32) {
        p2 = ld128u(s + 16);
        p3 = ld128u(s + min(32u, len - 16u));
        asm volatile("" ::"v"(p2), "v"(p3));
    }
#endif
    if (len >= 16) {
        st128u(d, p0);
        if (len > 16) st128u(d + len - 16, p1);
    } else {
        const bool c8 = (len & 8u) != 0, c4 = (len & 4u) != 0, c2 = (len & 2u) != 0;
        const u32 a0 = c8 ? p0.z : p0.x;
        const u32 a1 = c8 ? p0.w : p0.y;
        const u32 b0 = c4 ? a1 : a0;
        const u32 c0 = c2 ? b0 >> 16 : b0;
        const u32 o4 = len & 8u, o2 = len & 12u, o1 = len & 14u;
        if (c8) reinterpret_cast<snp_u64_unaligned*>(d)->v = p0.x | (static_cast<u64>(p0.y) << 32);
        if (c4) st32u(d + o4, a0);
        if (c2) reinterpret_cast<snp_u16_unaligned*>(d + o2)->v = static_cast<u16>(b0);
        if (len & 1u) d[o1] = static_cast<u8>(c0);
    }
    if (len > 32) {
#if SNP_D_PIECES != 4
        const u32x4 p2 = ld128u(s + 16), p3 = ld128u(s + min(32u, len - 16u));   // (unconditional: both in flight together;
        asm volatile("" ::"v"(p2), "v"(p3));                                     //  the compiler must not sink the second one)
#endif
        st128u(d + 16, p2);
        if (len > 48) st128u(d + 32, p3);
    }
}

// Bytes from the start of the tag whose first 8 bytes are q to the start of the next tag (Constants.cs:42-76: tag byte, 0..4
// trailer bytes, and the body of a literal).  At least 2; a literal's length saturates so that positions stay below 2^31.
__device__ __forceinline__ u32 tag_advance(u64 q)
{
    const u32 lo = static_cast<u32>(q);
    const u32 t = lo & 3u;
    const u32 h = __builtin_amdgcn_ubfe(lo, 2u, 6u);
    u32 adv = t ? __builtin_amdgcn_ubfe(0x05030200u, 8u * t, 8u) : h + 2u;   // copy-1/2/4: 2, 3, 5 bytes; short literal: tag + h + 1
    if (__builtin_expect((lo & 0xf3u) == 0xf0u, 0)) {                        // literal with 1..4 length bytes (rare: kept out of the common path)
        const u32 ex = h - 59u;
        const u32 b1234 = static_cast<u32>(q >> 8);
        const u32 tr = ex >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * ex);
        adv = 2u + ex + min(tr, 0x3fffffffu);
    }
    return adv;
}
// The same from a staged copy of the input: only the tag BYTE is read (ds_read_u8 -- an 8-byte read at an arbitrary address
// stalls the LDS pipe: SQ_LDS_UNALIGNED_STALL was a third of this kernel's LDS time); the length bytes of a long literal are
// fetched in the rare branch.
__device__ __forceinline__ u32 tag_advance_staged(const u8* at)
{
    const u32 c = at[0];
    const u32 t = c & 3u;
    const u32 h = c >> 2;
    u32 adv = t ? __builtin_amdgcn_ubfe(0x05030200u, 8u * t, 8u) : h + 2u;
    if (__builtin_expect((c & 0xf3u) == 0xf0u, 0)) {
        const u32 ex = h - 59u;
        const u32 b1234 = reinterpret_cast<const snp_u32_unaligned*>(at + 1)->v;
        const u32 tr = ex >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * ex);
        adv = 2u + ex + min(tr, 0x3fffffffu);
    }
    return adv;
}
// The same through a 256-entry table in LDS (SNP_D_ADV_LUT, FRONT = 3): entry c = the advance of tag byte c, 0 for a literal with length bytes
// (0xf0 / 0xf4 / 0xf8 / 0xfc: the rare branch computes it).  One more LDS read per trip instead of ~7 VALU instructions.
__device__ __forceinline__ u32 tag_advance_lut(const u8* at, const u8* lut)
{
    const u32 c = at[0];
    u32 adv = lut[c];
    if (__builtin_expect(adv == 0u, 0)) {
        const u32 ex = (c >> 2) - 59u;
        const u32 b1234 = reinterpret_cast<const snp_u32_unaligned*>(at + 1)->v;
        const u32 tr = ex >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * ex);
        adv = 2u + ex + min(tr, 0x3fffffffu);
    }
    return adv;
}
__device__ __forceinline__ u64 lds_ld64u(const u8* p) { return reinterpret_cast<const snp_u64_unaligned*>(p)->v; }

#ifndef SNP_D_STAGE
#define SNP_D_STAGE 2048    // queued mode: a batch whose output is contiguous and at most this long is assembled in LDS (0 = off)
#endif
#ifndef SNP_D_STAGE_MIN
#define SNP_D_STAGE_MIN 2   // staged batches: shortest prefix worth staging (a lone tag before a > 64-byte literal takes the unstaged path)
#endif
#ifndef SNP_D_WALK
#define SNP_D_WALK 1        // queued mode, finding the tag starts of a window: 0 scalar walk (4 tags per step), 1 pointer doubling through LDS
#endif
// DS operations of one wavefront execute in order; this only stops the compiler from reordering or forwarding them.
__device__ __forceinline__ void lanes_sync_lds() { asm volatile("" ::: "memory"); }

#ifndef SNP_D_TOPWAIT
#define SNP_D_TOPWAIT 0     // 1 = the round-2 form of the batch top (vmcnt drained at a join in every batch); A/B only
#endif
#ifndef SNP_D_ABLATE
#define SNP_D_ABLATE 0      // TIMING-ONLY ablations of the sub-chain front end (the output is wrong): 1 no first-pass copies, 2 no serial finish,
#endif                      // 4 no write-out, 16 no second pass, 32 tag lists only (no batches), 64 first-pass copy sources pulled to within 1 KiB (no far reads),
                            // 128 one 16-byte piece per tag in the first pass whatever its length (10.95 -> 9.70 ms: what the 18 % of tags longer than 16 bytes cost)
#ifndef SNP_D_PASS2
#define SNP_D_PASS2 1       // sub-chain front end: second lane-parallel pass over the tags pass 1 could not take (0: they all finish one by one)
#endif
#ifndef SNP_D_PIECES
#define SNP_D_PIECES 2      // lane_copy2: pieces requested together (2: head + tail, then the middle pair; 4: all at once)
#endif
#ifndef SNP_D_PF
#define SNP_D_PF 1          // sub-chain front end: request the next batch's tag bytes while this batch executes
#endif
#ifndef SNP_D_PASSES
#ifndef SNP_D_P2MIN
#define SNP_D_P2MIN 2     // FRONT 3: the second lane-parallel pass runs only for batches with at least this many pending tags
#endif
#define SNP_D_PASSES 1      // queued mode: extra lane-parallel passes over pending tags before the serial finish
#endif
#ifndef SNP_D_RING
#define SNP_D_RING 2048   // FRONT = 4: bytes of recent output per wavefront kept in an LDS ring (power of two).  Measured, 10 GiB html-like: 1 KiB 16.3 ms,
                          // 2 KiB 15.6, 4 KiB 16.7, 8 KiB 23.6 (22 / 20 / 16 / 11 wavefronts per CU: occupancy against far tags; profiles/r04o_ring_sizes*.txt)
#endif
#ifndef SNP_D_RING_SPAN
#define SNP_D_RING_SPAN 1024
#endif
#ifndef SNP_D_ADV_LUT
#define SNP_D_ADV_LUT 0   // sub-chain front end (FRONT = 3): tag advance of the chain walks from a 256-byte table in LDS (A/B)
#endif
#ifndef SNP_D_A2
#define SNP_D_A2 0        // sub-chain front end (FRONT = 3), phase A: two tags per trip when the first is a copy (A/B)
#endif
#ifndef SNP_D_PC
#define SNP_D_PC 0        // 1: k_decompress_chains becomes the two-wavefront producer / consumer form (FRONT = 5), A/B only
#endif
#ifndef SNP_D_CAP
#define SNP_D_CAP 128     // sub-chain front end: bytes a chain may overrun its region before the wave takes over (multiple of 32)
#endif
#ifndef SNP_D_ROUNDS
#define SNP_D_ROUNDS 1      // lane-parallel dependency rounds per batch before the rest is finished tag by tag (measured: 1 > 2 > 3)
#endif

// ---- optional event counters (build with -DSNP_D_PROF=1; scripts/prof_decompress.py) --------------------------------
#ifndef SNP_D_PROF
#define SNP_D_PROF 0
#endif
#if SNP_D_PROF
__device__ unsigned long long g_dprof[16];
#define DPROF_ADD(k, v) do { if (lane == 0 && ((k) >= 10 || SNP_D_PROF == 2)) atomicAdd(&g_dprof[k], static_cast<unsigned long long>(v)); } while (0)
#define DPROF_T0 u64 dprof_t = __builtin_readcyclecounter(); u64 dprof_acc[6] = {0, 0, 0, 0, 0, 0};
#define DPROF_TIME(k)                                                             \
    do {                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               \
        const u64 now_ = __builtin_readcyclecounter();                            \
        dprof_acc[(k) - 10] += now_ - dprof_t;                                    \
        dprof_t = now_;                                                           \
    } while (0)
#define DPROF_FLUSH do { for (int k_ = 0; k_ < 6; ++k_) DPROF_ADD(10 + k_, dprof_acc[k_]); } while (0)
#define DPROF_TRIP(v) ++(v)
#define DPROF_ADD_MAX(k, v)                                                                          \
    do {                                                                                             \
        u32 m_ = (v);                                                                                \
        for (int s_ = 32; s_ > 0; s_ >>= 1) m_ = max(m_, static_cast<u32>(__shfl_xor(static_cast<int>(m_), s_, 64)));   \
        DPROF_ADD(k, m_);                                                                            \
    } while (0)
#else
#define DPROF_TRIP(v)
#define DPROF_ADD_MAX(k, v)
#define DPROF_ADD(k, v)
#define DPROF_T0
#define DPROF_TIME(k)
#define DPROF_FLUSH
#endif

// FRAG = true decodes one 64 KiB output FRAGMENT of a larger block (see tag_index.hip): the wave starts at a tag
// boundary at or before the fragment (`frag_skip[b]` output bytes early), parses the tags in between without producing
// them, and stops when the fragment is full.  in_off/in_len = that tag start and the rest of the stream, out_off/out_cap
// = the fragment.  A tag that straddles the fragment start or a copy that reaches back before it (legal Snappy, but
// never produced by a compressor that works in independent 64 KiB fragments) ends with kIrregular; the caller then
// decodes the whole block with one wavefront instead, which also owns the exact error semantics.
constexpr i32 kIrregular = 99;

#ifndef SNP_D_WAVES
#define SNP_D_WAVES 0       // > 0: ask the compiler to fit this many wavefronts per SIMD (caps VGPRs at 512 / n)
#endif
#if SNP_D_WAVES
#define SNP_D_OCC __attribute__((amdgpu_waves_per_eu(SNP_D_WAVES, SNP_D_WAVES)))
#else
#define SNP_D_OCC
#endif

template <bool FENCED, int FRONT, bool FRAG>   // FRONT: 0 serial loop only, 1 token-parallel batches, 2 batches feeding a 64-tag execution queue, 3 sub-chain parse
__device__ __forceinline__ void decompress_block(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                 const u32* __restrict__ in_len, u32 nblocks, u8* out,
                                                 const u64* __restrict__ out_off,
                                                 const u32* __restrict__ out_cap, u32* __restrict__ out_len,
                                                 i32* __restrict__ status, const u8* __restrict__ chunk_type,
                                                 const u32* __restrict__ frag_skip, int redo_only, const u32 b)
{
    static_assert(!FRAG || (FRONT != 1 && FRONT != 3 && FRONT != 4 && FRONT != 5), "fragment mode: serial loop or queued front end");
    if (b >= nblocks) return;
    if (redo_only && status[b] != -1) return;            // decompress_small.hip finished this block (it marks the others -1)
    const u32 lane = lane_id();
    const u8* src = in + in_off[b];
    const u32 n = bcast_first(in_len[b]);
    const u32 skip = FRAG ? bcast_first(frag_skip[b]) : 0u;             // output bytes parsed but not produced
    u8* dst = out + out_off[b] - skip;                                  // output offsets below count from `skip` bytes early
    const u32 cap = bcast_first(out_cap[b]);

    if (n > 0x7fffffffu - 1024u) {                       // the reference's spans are int-length; keeps ip + k arithmetic below 2^32
        if (lane == 0) { out_len[b] = 0; status[b] = SNP_ERR_BAD_ARG; }
        return;
    }
    if (!FRAG && chunk_type && chunk_type[b] == 1) {     // framing: uncompressed chunk body  SnappyStreamDecompressor.cs:137-163
        const bool fits = n <= cap;
        if (fits) wave_copy(dst, src, n, lane);
        if (lane == 0) {
            out_len[b] = fits ? n : 0u;
            status[b] = fits ? SNP_OK : SNP_ERR_OUTPUT_TOO_SMALL;
        }
        return;
    }

    InWindow w;
    const u32 mis = static_cast<u32>(reinterpret_cast<uintptr_t>(src) & 3u);
    w.a0 = src - mis;
    w.end = src + n;
    w.wv = 0;
    w.lo = win_load(w, 4 * lane);
    w.hi = win_load(w, 256 + 4 * lane);

    i32 st = SNP_OK;
    u32 ip = 0, op = 0, expected = 0;

    // ---- varint preamble  (VarIntEncoding.TryReadSlow  VarIntEncoding.Read.cs:38-79) -------------------------
    if (FRAG) {
        expected = skip + cap;                                          // no preamble: the fragment ends `cap` bytes after its start
    } else {
        const u64 q = win_fetch(w, mis, lane);
        u32 shift = 0, result = 0;
        bool done = false;
        for (u32 i = 0; i < 5 && !done; ++i) {
            if (i >= n) { st = SNP_ERR_INCOMPLETE; break; }          // NeedMoreData -> never AllDataDecompressed
            const u32 c = static_cast<u32>(q >> (8 * i)) & 0xffu;
            const u32 val = c & 0x7fu;
            if (val & ~(0xffffffffu >> shift)) { st = SNP_ERR_BAD_LENGTH; break; }   // LeftShiftOverflows  Helpers.cs:65-70
            result |= val << shift;
            shift += 7;
            ip = i + 1;
            if (c < 128) done = true;
        }
        if (st == SNP_OK && !done) st = SNP_ERR_BAD_LENGTH;            // five continuation bytes: shift >= 32  :65-69
        expected = result;
        if (st == SNP_OK && expected > 0x7fffffffu) st = SNP_ERR_BAD_LENGTH;   // (int)length < 0 in the reference
        if (st == SNP_OK && cap < expected) st = SNP_ERR_OUTPUT_TOO_SMALL;     // Snappy.cs:183-185
    }

    // ---- token-parallel batches (see the header) ------------------------------------------------------------------
    if (FRONT == 1) {
        // the next batch's input window is requested as soon as this batch's length is known, so its latency overlaps
        // this batch's copies
        u64 q_next = (st == SNP_OK && ip + 72 <= n) ? ld64u(src + ip + lane) : 0ull;
        DPROF_T0
        while (st == SNP_OK && ip + 72 <= n && op < expected) {
            // 1. every lane decodes the tag that would start at ip + lane
            const u64 q = q_next;
            DPROF_TIME(10);                                             // wait for the input window
            const u32 c = static_cast<u32>(q) & 0xffu;
            const u32 type = c & 3u;
            const u32 hi6 = c >> 2;
            const u32 b1234 = static_cast<u32>(q >> 8);
            const u32 extra = type == 0 ? (hi6 >= 60 ? hi6 - 59 : 0) : (type == 3 ? 4 : type);
            const u32 trailer = extra >= 4 ? b1234 : (b1234 & ((1u << (8 * extra)) - 1u));
            u32 len, off = 0;
            if (type == 0) len = hi6 >= 60 ? trailer + 1 : hi6 + 1;    // wraps to 0 for a 2^32-byte literal: caught below
            else if (type == 1) { len = (hi6 & 7u) + 4; off = ((c >> 5) << 8) | (b1234 & 0xffu); }
            else { len = hi6 + 1; off = trailer; }
            const u32 body = lane + 1 + extra;                          // literal body starts at ip + body
            const u32 n1 = body + (type == 0 ? min(len, 0x40000000u) : 0);   // next tag, relative to ip; always > lane
            // 2. next-pointers 2, 3 and 4 hops ahead; a value >= 64 leaves the window and then sticks
            // (the bpermutes run with every lane active: a lane that is masked off reads back as 0 to its readers)
            const u32 h2 = bperm(n1, n1);
            const u32 n2 = n1 < 64 ? h2 : n1;
            const u32 h3 = bperm(n1, n2);
            const u32 n3 = n1 < 64 ? h3 : n1;
            const u32 h4 = bperm(n2, n2);
            const u32 n4 = n2 < 64 ? h4 : n2;
            DPROF_TIME(11);                                             // tag decode + next-pointers
            u64 tags = 0;
            u32 pos = 0;
            do {
                const u32 a = read_lane(n1, pos), bq = read_lane(n2, pos), cq = read_lane(n3, pos), dq = read_lane(n4, pos);
                tags |= ballot64(lane == pos || lane == a || lane == bq || lane == cq);
                pos = dq;
            } while (pos < 64);
            const u32 consumed = pos;                                   // input bytes this batch covers
            DPROF_TIME(12);                                             // chain walk
            if (consumed <= n - ip && ip + consumed + 72 <= n) q_next = ld64u(src + ip + consumed + lane);
            const bool real = (tags >> lane) & 1ull;
            // 3. output offsets
            const u32 olen = real ? len : 0u;
            const u32 incl = wave_inclusive_scan(olen);
            const u32 total = read_lane(incl, 63);
            const u32 ostart = op + incl - olen;
            // irregular batches are left to the serial loop (it reports the exact status)
            // (+16: lane_copy may read up to 15 bytes past a literal's body / a copy's source)
            const bool bad = real && (type == 0 ? (len == 0 || len + 16 > n - ip || body > n - ip - len - 16)
                                                : (off == 0 || off > ostart));
            if (ballot64(bad) != 0ull || total + 16 > expected - op || consumed > n - ip) break;

            DPROF_TIME(13);                                             // prefix sum + checks (+ issue of the next window load)
            DPROF_ADD(0, 1);                                            // batches
            DPROF_ADD(1, __builtin_popcountll(tags));                   // tags in batches
            DPROF_ADD(2, total);                                        // output bytes of batches
            const bool is_lit = type == 0;
            u8* const d = dst + ostart;
            const u8* const s = is_lit ? src + ip + body : dst + (ostart - off);
            // 4a. literals longer than 64 bytes: whole-wave memcpy each
            u64 pend = tags;
            u64 big = ballot64(real && is_lit && len > 64);
            pend &= ~big;
            while (big) {
                const u32 t = static_cast<u32>(__builtin_ctzll(big));
                big &= big - 1;
                wave_copy(dst + read_lane(ostart, t), src + ip + read_lane(body, t), read_lane(len, t), lane);
            }
            // 4b. dependency rounds
            const bool simple = is_lit || off >= len;                   // executable by one lane with wide copies
            const u32 src_end = ostart - off + len;                     // copies: one past the last source byte
            u32 mark = op;                                              // all output below `mark` is complete
            for (u32 round = 0;; ++round) {
                DPROF_ADD(3, 1);                                        // rounds (incl. the finishing pass)
                if (round == SNP_D_ROUNDS) {
                    DPROF_ADD(4, __builtin_popcountll(pend));           // tags finished one by one
                    // a long dependency chain inside the batch: finish it tag by tag, whole wave per tag
                    while (pend) {
                        const u32 f = static_cast<u32>(__builtin_ctzll(pend));
                        pend &= pend - 1;
                        const u32 f_o = read_lane(ostart, f), f_off = read_lane(off, f), f_len = read_lane(len, f);
                        if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        u32 sidx = lane;
                        if (f_off < f_len) {
#pragma unroll
                            for (int sh = 5; sh >= 0; --sh) {
                                const u32 t = f_off << sh;
                                sidx = min(sidx, sidx - t);
                            }
                        }
                        if (lane < f_len) dst[f_o + lane] = dst[f_o - f_off + sidx];
                    }
                    break;
                }
                if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const bool ready = ((pend >> lane) & 1ull) && simple && (is_lit || src_end <= mark);
                const u64 rmask = ballot64(ready);
                DPROF_ADD(5 + (round < 3 ? round : 2), __builtin_popcountll(rmask));   // tags executed in round 0 / 1 / 2
                if (ready) lane_copy(d, s, len);
                pend &= ~rmask;
                if (!pend) break;
                const u32 f = static_cast<u32>(__builtin_ctzll(pend));  // first tag not yet executed: everything before it is
                mark = read_lane(ostart, f);
                const u32 f_off = read_lane(off, f), f_len = read_lane(len, f);
                if (f_off < f_len) {                                    // pattern copy: cooperative, as in the serial loop
                    DPROF_ADD(8, 1);                                    // pattern copies done by the whole wave
                    if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    u32 sidx = lane;
#pragma unroll
                    for (int sh = 5; sh >= 0; --sh) {
                        const u32 t = f_off << sh;
                        sidx = min(sidx, sidx - t);
                    }
                    if (lane < f_len) dst[mark + lane] = dst[mark - f_off + sidx];
                    pend &= ~(1ull << f);
                    mark += f_len;
                    if (!pend) break;
                }
            }
            DPROF_TIME(14);                                             // copies (all rounds)
            ip += consumed;
            op += total;
        }
        DPROF_FLUSH;
        w.wv = 0x80000000u;                                             // force the serial loop to re-seat its window
    }

    // ---- token-parallel parse feeding an execution queue -----------------------------------------------------------------
    // Every vector-memory instruction costs the texture-address unit ~16 cycles whether 5 or 64 of its lanes are active,
    // and a 64-byte window holds only ~21 tags.  So windows are parsed as above but their tags are appended to a queue
    // in LDS, and copies are executed 64 tags at a time: the same ~10 memory instructions then serve three windows.
    if (FRONT == 2) {
        __shared__ u32 q_ostart[128], q_arg[128], q_meta[128];          // ring: output offset | copy offset or literal
        __shared__ u8 s_reach[64];                                      // tag-start flags of the window being parsed
#if SNP_D_STAGE
        __shared__ u8 s_stage[SNP_D_STAGE + 64];                        // the current batch's output bytes (staged batches)
#endif
        __shared__ u64 s_busy[65];                                      // one bit per output byte of the batch that a pending tag still has to write
        u32 head = 0, count = 0;                                        // input position | length + literal flag
        bool parsing = st == SNP_OK;
        u64 q_next = (parsing && ip + 72 <= n) ? ld64u(src + ip + lane) : 0ull;
        DPROF_T0
        for (;;) {
            head = bcast_first(head);
            count = bcast_first(count);
            while (parsing && count <= 64 && ip + 72 <= n && op < expected) {
                // ---- parse one 64-byte window (steps 1-3 of the batched path) ----
                const u64 q = q_next;
                DPROF_TIME(10);                                         // wait for the input window
                const u32 c = static_cast<u32>(q) & 0xffu;
                const u32 type = c & 3u;
                const u32 hi6 = c >> 2;
                const u32 b1234 = static_cast<u32>(q >> 8);
                // (selects, not branches: all 64 lanes decode, and the three tag classes are evenly mixed)
                const bool is_lit = type == 0;
                const bool long_lit = is_lit && hi6 >= 60;
                const u32 extra = is_lit ? (long_lit ? hi6 - 59 : 0u) : (type == 3 ? 4u : type);
                const u32 trailer = extra >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * extra);
                const u32 len = (long_lit ? trailer : (hi6 & (type == 1 ? 7u : 63u))) + (type == 1 ? 4u : 1u);   // literal / copy-1 / copy-2,4
                const u32 off = is_lit ? 0u : (type == 1 ? (((c >> 5) << 8) | (b1234 & 0xffu)) : trailer);
                const u32 body = lane + 1 + extra;
                const u32 n1 = body + (is_lit ? min(len, 0x40000000u) : 0u);
#if SNP_D_WALK == 0
                const u32 h2 = bperm(n1, n1);
                const u32 n2 = n1 < 64 ? h2 : n1;
                const u32 h3 = bperm(n1, n2);
                const u32 n3 = n1 < 64 ? h3 : n1;
                const u32 h4 = bperm(n2, n2);
                const u32 n4 = n2 < 64 ? h4 : n2;
                u64 tags = 0;
                u32 pos = 0;
                do {
                    const u32 a = read_lane(n1, pos), bq = read_lane(n2, pos), cq = read_lane(n3, pos), dq = read_lane(n4, pos);
                    tags |= ballot64(lane == pos || lane == a || lane == bq || lane == cq);
                    pos = dq;
                } while (pos < 64);
                const u32 consumed = pos;
#else
                // Which positions are tag starts = which are reachable from position 0 along n1.  Pointer doubling: hop
                // tables for 1, 2, 4, 8, 16 tags (a value >= 64 leaves the window and sticks), then five rounds in which
                // every position already reached marks the one 2^k tags further on -- through a 64-byte flag array in LDS,
                // because a scatter needs the senders masked.  Five rounds reach every tag: a tag is at least two bytes
                // long, so a 64-byte window starts at most 32 of them (hops 0..31).  ~60 vector/LDS instructions per
                // window whatever the number of tags, instead of a scalar walk of ~4 instructions per tag.
                // (round 0 needs no scatter: position 0 reaches exactly n1 of lane 0)
                u32 hop = n1;
                bool reached = lane == 0 || lane == read_lane(n1, 0);
                s_reach[lane] = reached ? 1 : 0;
#pragma unroll
                for (int k = 1; k < 5; ++k) {
                    const u32 h = bperm(hop, hop);                      // hop table for 2^k tags
                    hop = hop < 64 ? h : hop;
                    lanes_sync_lds();
                    if (reached && hop < 64) s_reach[hop] = 1;
                    lanes_sync_lds();
                    reached = s_reach[lane] != 0;
                }
                const u64 tags = ballot64(reached);
                const u32 consumed = read_lane(n1, 63u - static_cast<u32>(__builtin_clzll(tags)));   // where the last tag of the window ends
#endif
                if (consumed <= n - ip && ip + consumed + 72 <= n) q_next = ld64u(src + ip + consumed + lane);
#if SNP_D_WALK == 0
                const bool real = (tags >> lane) & 1ull;
#else
                const bool real = reached;
#endif
                const u32 olen = real ? len : 0u;
                const u32 incl = wave_inclusive_scan(olen);
                const u32 total = read_lane(incl, 63);
                const u32 ostart = op + incl - olen;
                const bool live = real && (!FRAG || ostart >= skip);    // FRAG: tags before the fragment are only parsed
                // literal: 1 <= len, body + len + 16 <= n - ip (lane_copy over-reads 15 bytes); copy: 1 <= off <= bytes produced
                const u32 room = n - ip - 16;                           // n - ip >= 72 here
                // (bitwise, not short-circuit: these are lane masks, and branches over a handful of compares cost more)
                const bool lit_ok = ((len - 1u) < room) & (body <= room - len);
                const bool copy_ok = (off - 1u) < (ostart - skip);
                const bool tag_ok = (is_lit & lit_ok) | (!is_lit & copy_ok);
                bool bad = live & !tag_ok;
                if (FRAG) bad = bad | (real & is_lit & (len == 0)) | (real & (ostart < skip) & (len > skip - ostart));   // straddles the fragment start
                if (ballot64(bad) != 0ull || total + 16 > expected - op || consumed > n - ip) { parsing = false; continue; }
                // literals longer than 64 bytes do not depend on anything: whole-wave memcpy right away
                u64 big = ballot64(live && is_lit && len > 64);
                const u64 enq = (FRAG ? ballot64(live) : tags) & ~big;
                while (big) {
                    const u32 t = static_cast<u32>(__builtin_ctzll(big));
                    big &= big - 1;
                    wave_copy(dst + read_lane(ostart, t), src + ip + read_lane(body, t), read_lane(len, t), lane);
                }
                // append the window's tags to the queue, in order
                if ((enq >> lane) & 1ull) {
                    const u32 slot = (head + count + static_cast<u32>(__builtin_popcountll(enq & lanes_below(lane)))) & 127u;
                    q_ostart[slot] = ostart;
                    q_arg[slot] = is_lit ? ip + body : off;
                    q_meta[slot] = len | (is_lit ? 0x100u : 0u);
                }
                count += static_cast<u32>(__builtin_popcountll(enq));
                ip += consumed;
                op += total;
                count = bcast_first(count);
                DPROF_TIME(11);                                         // parse: decode, chain walk, prefix sum, enqueue
            }
            if (count == 0) break;
            // ---- execute up to 64 queued tags: one lane-parallel pass, then dependent tags one by one ----
            asm volatile("" ::: "memory");                              // the queue was written by other lanes
            const u32 ne = count < 64 ? count : 64u;
            const bool act = lane < ne;
            const u32 slot = (head + lane) & 127u;
            const u32 e_ostart = q_ostart[slot], e_arg = q_arg[slot], e_meta = q_meta[slot];
            const u32 e_len = e_meta & 0xffu;
            const bool e_lit = (e_meta & 0x100u) != 0;
            const u32 e_off = e_lit ? 0u : e_arg;
            const u32 mark = read_lane(e_ostart, 0);                    // everything before the first queued tag is complete
            const bool ready = act && (e_lit || (e_off >= e_len && e_ostart - e_off + e_len <= mark));
            if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if SNP_D_STAGE
            // Staged batch.  When the batch's output is one contiguous run of at most SNP_D_STAGE bytes it is assembled
            // in LDS and leaves as one coalesced copy: tags write their 1..64 bytes into the stage (LDS stores instead of
            // ~8 partial-sector global store instructions per pass), tags that depend on this batch's own output read it
            // from the stage at LDS latency, and the wave then writes the whole run, 16 bytes per lane per instruction.
            {
                const u32 my_end = e_ostart + e_len;
                const u32 prev_end = static_cast<u32>(__shfl_up(static_cast<int>(my_end), 1, 64));
                const bool gap = act && lane > 0 && e_ostart != prev_end;      // a > 64-byte literal was copied at parse time
                // the longest prefix of the batch that is contiguous and fits the stage (all of it, normally)
                const u64 gapm = ballot64(gap);
                u32 nst = static_cast<u32>(__builtin_popcountll(ballot64(act && my_end - mark <= SNP_D_STAGE)));
                if (gapm) nst = min(nst, static_cast<u32>(__builtin_ctzll(gapm)));
                if (nst >= SNP_D_STAGE_MIN) {
                    const u32 ne = nst;                                 // (shadows: this batch is the prefix)
                    const bool act = lane < ne;
                    const bool ready = act && (e_lit || (e_off >= e_len && e_ostart - e_off + e_len <= mark));
                    const u32 span = read_lane(my_end, ne - 1) - mark;
                    u8* const my = s_stage + (e_ostart - mark);
                    const u32 s_lo = e_ostart - e_off;
                    if (ready) lane_copy(my, e_lit ? src + e_arg : dst + s_lo, e_len);
                    u64 pend = ballot64(act && !ready);
                    DPROF_ADD(0, 1);
                    DPROF_ADD(1, ne);
                    DPROF_ADD(5, __builtin_popcountll(pend));
                    DPROF_ADD(6, 1);                                    // staged batches
                    DPROF_TIME(12);
                    if (pend) {
                        // second lane-parallel pass: sources that lie entirely inside the batch and do not touch the
                        // output of a tag that is still pending (bitmap of pending output bytes, as below)
                        bool blocked = e_off < e_len || s_lo < mark;    // pattern copies and sources straddling `mark`: serial finish
                        const bool mine = (pend >> lane) & 1ull;
                        if (pend & (pend - 1)) {
                            s_busy[lane] = 0ull;
                            lanes_sync_lds();
                            if (mine) {
                                const u32 r = e_ostart - mark, b0 = r & 63u;
                                const u64 m = e_len >= 64 ? ~0ull : ((1ull << e_len) - 1ull);
                                atomicOr(reinterpret_cast<unsigned long long*>(&s_busy[r >> 6]), static_cast<unsigned long long>(m << b0));
                                if (b0 && (m >> (64u - b0)))
                                    atomicOr(reinterpret_cast<unsigned long long*>(&s_busy[(r >> 6) + 1]), static_cast<unsigned long long>(m >> (64u - b0)));
                            }
                            lanes_sync_lds();
                            if (mine && !blocked) {
                                const u32 lo = s_lo - mark, b0 = lo & 63u;
                                const u64 m = e_len >= 64 ? ~0ull : ((1ull << e_len) - 1ull);
                                const u64 w0 = s_busy[lo >> 6], w1 = s_busy[(lo >> 6) + 1];
                                blocked = ((w0 & (m << b0)) | (b0 ? (w1 & (m >> (64u - b0))) : 0ull)) != 0ull;
                            }
                        }
                        const bool ready2 = mine && !blocked;
                        lanes_sync_lds();
                        if (ready2) lane_copy(my, s_stage + (s_lo - mark), e_len);
                        pend &= ~ballot64(ready2);
                        DPROF_ADD(4, __builtin_popcountll(pend));
                        DPROF_TIME(13);
                        while (pend) {                                  // the rest in order, whole wave per tag, a byte per lane
                            const u32 f = static_cast<u32>(__builtin_ctzll(pend));
                            pend &= pend - 1;
                            const u32 f_o = read_lane(e_ostart, f), f_off = read_lane(e_off, f), f_len = read_lane(e_len, f);
                            u32 sidx = lane;
                            if (f_off < f_len) {
#pragma unroll
                                for (int sh = 5; sh >= 0; --sh) {
                                    const u32 t = f_off << sh;
                                    sidx = min(sidx, sidx - t);
                                }
                            }
                            const u32 spos = f_o - f_off + sidx;        // output position this lane's byte comes from
                            lanes_sync_lds();
                            u32 byte = 0;
                            if (f_o - f_off >= mark) {                  // the whole source lies in this batch
                                if (lane < f_len) byte = s_stage[spos - mark];
                            } else if (lane < f_len) {                  // it starts before the batch: those bytes are in global memory
                                if (spos < mark) byte = dst[spos];
                                else byte = s_stage[spos - mark];       // (two branches: one select would make this a flat load)
                            }
                            lanes_sync_lds();
                            if (lane < f_len) s_stage[f_o - mark + lane] = static_cast<u8>(byte);
                        }
                    }
                    // the whole run, coalesced
                    lanes_sync_lds();
                    u8* const g = dst + mark;
                    for (u32 i = lane * 16; i + 16 <= span; i += SNP_WAVE * 16)
                        *reinterpret_cast<snp_u128_unaligned*>(g + i) = *reinterpret_cast<const snp_u128_unaligned*>(s_stage + i);
                    const u32 tail = span & ~15u;
                    if (tail + lane < span) g[tail + lane] = s_stage[tail + lane];
                    lanes_sync_lds();
                    head = (head + ne) & 127u;
                    count -= ne;
                    DPROF_TIME(14);
                    continue;
                }
            }
#endif
            if (ready) lane_copy(dst + e_ostart, e_lit ? src + e_arg : dst + (e_ostart - e_off), e_len);
            u64 pend = ballot64(act && !ready);
            DPROF_ADD(0, 1);                                            // execution batches
            DPROF_ADD(1, ne);                                           // tags executed
            DPROF_ADD(5, __builtin_popcountll(pend));                   // tags not ready in the first pass
            DPROF_TIME(12);                                             // first pass
            // More lane-parallel passes: a pending copy may run as soon as its source no longer overlaps the output of
            // another pending tag (those bytes do not exist yet).  Most near copies read what an earlier pass just wrote;
            // each pass peels one level off every dependency chain.  Pattern copies go through the serial finish.
            const u32 s_lo = e_ostart - e_off, s_hi = s_lo + e_len, e_end = e_ostart + e_len;
#if SNP_D_PASSES >= 1
            // One more lane-parallel pass: a pending copy may run as soon as its source no longer overlaps the OUTPUT of a
            // tag that is still pending (those bytes do not exist yet).  The union of the pending outputs is a bitmap over
            // the batch's output span, one bit per byte, 64 words of 64 bits in LDS (a batch of 64 tags of <= 64 bytes
            // spans <= 4096 bytes unless > 64-byte literals sit in between): every pending tag ORs its range in, then
            // every pending tag tests its source range against it -- two LDS atomics and two LDS reads per lane instead
            // of a scalar loop over all pending tags (~15 instructions per pending tag, 12 of them per batch on html).
            if (pend & (pend - 1)) {
                const u32 span = read_lane(e_end, ne - 1) - mark;
                bool blocked = e_off < e_len;                           // pattern copies go through the serial finish
                if (span <= 4096) {
                    s_busy[lane] = 0ull;
                    if (lane == 0) s_busy[64] = 0ull;
                    lanes_sync_lds();
                    const bool mine = (pend >> lane) & 1ull;
                    if (mine) {
                        const u32 r = e_ostart - mark, b0 = r & 63u;
                        const u64 m = e_len >= 64 ? ~0ull : ((1ull << e_len) - 1ull);
                        atomicOr(reinterpret_cast<unsigned long long*>(&s_busy[r >> 6]), static_cast<unsigned long long>(m << b0));
                        if (b0 && (m >> (64u - b0)))
                            atomicOr(reinterpret_cast<unsigned long long*>(&s_busy[(r >> 6) + 1]), static_cast<unsigned long long>(m >> (64u - b0)));
                    }
                    lanes_sync_lds();
                    if (mine && !blocked) {
                        const u32 lo = s_lo > mark ? s_lo - mark : 0u;   // bytes below `mark` are complete
                        if (s_hi > mark) {
                            const u32 n_b = s_hi - mark - lo, b0 = lo & 63u;      // 1 .. 64 source bytes inside the batch
                            const u64 m = n_b >= 64 ? ~0ull : ((1ull << n_b) - 1ull);
                            const u64 w0 = s_busy[lo >> 6], w1 = s_busy[(lo >> 6) + 1];
                            blocked = ((w0 & (m << b0)) | (b0 ? (w1 & (m >> (64u - b0))) : 0ull)) != 0ull;
                        }
                    }
                } else {
                    u64 it = pend;
                    while (it) {
                        const u32 f = static_cast<u32>(__builtin_ctzll(it));
                        it &= it - 1;
                        const u32 f_o = read_lane(e_ostart, f), f_end = read_lane(e_end, f);
                        blocked = blocked || (s_lo < f_end && s_hi > f_o);   // f >= lane cannot overlap: s_hi <= own ostart
                    }
                }
                const bool ready2 = ((pend >> lane) & 1ull) && !blocked;
                if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (ready2) lane_copy(dst + e_ostart, dst + s_lo, e_len);
                pend &= ~ballot64(ready2);
            }
#endif
            DPROF_ADD(4, __builtin_popcountll(pend));                   // tags finished one by one
            DPROF_TIME(13);                                             // extra pass(es)
            while (pend) {
                const u32 f = static_cast<u32>(__builtin_ctzll(pend));
                pend &= pend - 1;
                const u32 f_o = read_lane(e_ostart, f), f_off = read_lane(e_off, f), f_len = read_lane(e_len, f);
                if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                u32 sidx = lane;
                if (f_off < f_len) {
#pragma unroll
                    for (int sh = 5; sh >= 0; --sh) {
                        const u32 t = f_off << sh;
                        sidx = min(sidx, sidx - t);
                    }
                }
                if (lane < f_len) dst[f_o + lane] = dst[f_o - f_off + sidx];
            }
            asm volatile("" ::: "memory");
            head = (head + ne) & 127u;
            count -= ne;
            DPROF_TIME(14);                                             // serial finish
        }
        DPROF_FLUSH;
        w.wv = 0x80000000u;
    }

    // ---- sub-chain parse feeding lane-parallel execution (FRONT = 3) ------------------------------------------------------------
    // The 64-byte windows above cost ~160 wave instructions and ~15 dependent LDS round trips per ~21 tags, all to find out
    // WHERE the tags start.  Here a SUPER-WINDOW of 64 x 32 = 2 KiB of compressed input is staged in LDS and every lane
    // walks a chain of tags through its own 32-byte region, starting blindly at the region's first byte.  A chain that
    // starts in the middle of a tag reads garbage, but a garbage chain and the true chain that land on the same byte
    // are the same chain from there on, and they meet within a few tags.  So:
    //   A   lane k walks region k from its first byte and records the positions it visits (32-bit mask V_k);
    //   A'  it walks on past the region's end until it lands on a position the owner of that region has visited (the
    //       chains have merged: m_k, next lane nx_k), recording these overrun positions too (a per-lane bitmap in LDS,
    //       kCap = 128 bytes far at most);
    //   R   lane 0's chain is the true one (the super-window starts at a tag): the lanes reachable from lane 0 along nx
    //       are the lanes whose chains are true from their entry m_prev on (pointer doubling over the lanes); a true
    //       chain that did not merge within kCap bytes is walked on by the whole wave, one tag at a time (rare);
    //   T   true tag starts = each active lane's V_k from its entry on, plus its overrun positions: a 2048-bit map, its
    //       popcount prefix numbers the tags, and the positions are written out as a u16 list (over the staged input).
    // ~1 500 wave instructions per ~620 tags (html) instead of ~4 700, and ~40 dependent LDS round trips instead of ~440.
    // Tags then execute 64 at a time straight from that list: position -> tag bytes (one 8-byte load per lane) -> decode ->
    // prefix sum of the output lengths -> the staged batch of the queued front end (assembled in LDS, written out coalesced).
    // A batch ends before the first tag it cannot take (malformed, a literal > 64 bytes, the last 16 output bytes): a long
    // literal is copied by the whole wave and parsing goes on; anything else falls to the serial loop below.
    if (FRONT == 3) {
        constexpr u32 kR = 32;                                          // input bytes per lane region
        constexpr u32 kW = SNP_WAVE * kR;                               // the super-window
        constexpr u32 kCap = SNP_D_CAP;                                       // a chain may overrun its region by this much before the wave takes over
        __shared__ __attribute__((aligned(16))) u8 c_in[kW];            // its bytes; afterwards the tag positions (u16 each, < kW / 2 of them)
        __shared__ __attribute__((aligned(16))) u8 c_stage[SNP_D_STAGE + 64];
        __shared__ u64 c_busy[65];                                      // batches: pending output bytes; while a super-window is built: V and T
#if SNP_D_ADV_LUT
        __shared__ u8 c_adv[256];
        for (u32 e = lane; e < 256; e += SNP_WAVE) {
            const u32 t = e & 3u, h = e >> 2;
            c_adv[e] = static_cast<u8>(t ? (t == 1 ? 2u : t == 2 ? 3u : 5u) : (h >= 60 ? 0u : h + 2u));
        }
        lanes_sync_lds();
#define SNP_ADV(at) tag_advance_lut(at, c_adv)
#else
#define SNP_ADV(at) tag_advance_staged(at)
#endif
        u32* const c_V = reinterpret_cast<u32*>(c_busy);
        u32* const c_T = c_V + SNP_WAVE;
        u16* const c_pos = reinterpret_cast<u16*>(c_in);
        const u32 r0 = kR * lane;
        u32 wbase = ip, ntok = 0, emitted = 0, consumed = 0;
        u64 q_pf = 0;                                                   // tag bytes of the batch that starts at list index pf_at,
        u32 pf_at = ~0u;                                                // requested while the batch before it executes
        DPROF_T0
        while (st == SNP_OK) {
            if (emitted == ntok) {
                // ---- the next super-window ----
                ip = wbase + consumed;
                if (ip + 72 > n || op >= expected) break;
                wbase = ip;
                const u32 avail = n - wbase;
                const u32 L = min(kW, avail) - 8u;                      // tags may start below L: their 8 bytes lie inside the staged input
                const u8* const wsrc = src + wbase;
                {
                    // 2 x 16 bytes per lane; a piece that would cross the end of the input is pulled back inside it (avail >= 72;
                    // the bytes it rewrites are the same bytes), pieces beyond it are not needed
                    const u32 oa = lane * 16u, ob = oa + 1024u;
                    const u32 la = min(oa, avail - 16u), lb = min(ob, avail - 16u);
                    const u32x4 va = ld128u(wsrc + la), vb = ld128u(wsrc + lb);
                    st128u(c_in + la, va);
                    st128u(c_in + lb, vb);
                }
                lanes_sync_lds();
                DPROF_TIME(10);                                         // input staged
                // A: the chain from the first byte of the lane's region
                u32 p = r0, V = 0;
                [[maybe_unused]] u32 trips = 0;
                {
                    const u32 lim = min(r0 + kR, L);
#if SNP_D_A2
                    // Two tags per trip when the first is a COPY (74 % of html tags): its successor can only start 2, 3 or 5 bytes on, so those
                    // three bytes are read together with the tag byte and the second advance costs no second LDS round trip
                    // (p < L = staged - 8: the reads stay inside the staged bytes).
                    while (p < lim) {
                        const u32 c0 = c_in[p], b2 = c_in[p + 2], b3 = c_in[p + 3], b5 = c_in[p + 5];
                        V |= 1u << (p - r0);
                        const u32 t0 = c0 & 3u;
                        if (__builtin_expect((c0 & 0xf3u) == 0xf0u, 0)) {       // literal with length bytes: the plain form
                            p += tag_advance_staged(c_in + p);
                        } else if (t0 == 0) {
                            p += (c0 >> 2) + 2u;
                        } else {
                            const u32 a0 = __builtin_amdgcn_ubfe(0x05030200u, 8u * t0, 8u);
                            const u32 p1 = p + a0;
                            const u32 c1 = t0 == 1 ? b2 : t0 == 2 ? b3 : b5;
                            p = p1;
                            if (p1 < lim) {
                                V |= 1u << (p1 - r0);
                                if (__builtin_expect((c1 & 0xf3u) == 0xf0u, 0)) p = p1 + tag_advance_staged(c_in + p1);
                                else p = p1 + ((c1 & 3u) ? __builtin_amdgcn_ubfe(0x05030200u, 8u * (c1 & 3u), 8u) : (c1 >> 2) + 2u);
                            }
                        }
                        DPROF_TRIP(trips);
                    }
#else
                    while (p < lim) {
                        V |= 1u << (p - r0);
                        p += SNP_ADV(c_in + p);
                        DPROF_TRIP(trips);
                    }
#endif
                }
                DPROF_ADD_MAX(3, trips);                                // loop trips of phase A
                c_V[lane] = V;
                c_T[lane] = 0;
                lanes_sync_lds();
                // A': on past the region until the chain lands on a position its owner visited (one loop exit: the exec-mask
                // bookkeeping of a divergent loop is scalar work, and the scalar unit is the busiest one in this kernel)
                u32 nx = 64u;                                           // 64: the chain leaves the super-window, 65: no merge within kCap bytes
                u32* const c_O = reinterpret_cast<u32*>(c_stage + 512) + lane * (kCap / 32);   // overrun positions, a bit each, from obase
#pragma unroll                                                          // (in the idle stage: keeping them in registers costs 20 instructions per trip)
                for (u32 w = 0; w < kCap / 32; ++w) c_O[w] = 0;
                const u32 obase = p & ~(kR - 1u);
                for (bool go = p < L; go;) {
                    const u32 v = c_V[p >> 5];
                    const u32 adv = SNP_ADV(c_in + p);
                    const u32 rel = p - obase;
                    const bool hit = (v >> (p & 31u)) & 1u;
                    const bool stop = hit | (rel >= kCap);
                    nx = stop ? (hit ? p >> 5 : 65u) : nx;
                    atomicOr(&c_O[min(rel >> 5, kCap / 32 - 1)], stop ? 0u : 1u << (rel & 31u));   // (unconditional: no exec-mask bookkeeping)
                    p = stop ? p : p + adv;
                    go = !stop & (p < L);
                    DPROF_TRIP(trips);
                }
                const u32 m = p;                                        // where the chain merged, gave up or left
                DPROF_ADD_MAX(6, trips);                                // ... of A and A' together
                // R: the lanes on the true chain = the lanes reachable from lane 0 along nx, by pointer doubling (flags through
                // LDS: a scatter needs its senders masked); each of them tells its successor where it enters.  ~70 wave
                // instructions instead of a scalar walk of ~14 per lane on the chain (~46 of them on html).
                u64 active;
                u32 entry = 0;
                {
                    u8* const c_reach = c_stage;                        // (the stage is idle while a super-window is built)
                    u32* const c_entry = reinterpret_cast<u32*>(c_stage + SNP_WAVE);
                    u32 hop = nx;
                    bool reached = lane == 0;
                    c_reach[lane] = reached ? 1 : 0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        lanes_sync_lds();
                        if (reached && hop < 64u) c_reach[hop] = 1;
                        lanes_sync_lds();
                        reached = c_reach[lane] != 0;
                        const u32 h2 = bperm(hop, hop);
                        hop = hop < 64u ? h2 : hop;
                    }
                    if (reached && nx < 64u) c_entry[nx] = m;
                    lanes_sync_lds();
                    if (lane) entry = c_entry[lane];
                    lanes_sync_lds();
                    active = ballot64(reached);
                    const u64 ends = ballot64(reached && nx == 64u);    // the lane whose chain leaves the super-window, if the chain gets there
                    consumed = ends ? read_lane(m, static_cast<u32>(__builtin_ctzll(ends))) : 0u;
                }
                if (ballot64(((active >> lane) & 1ull) && nx == 65u)) {
                    // a chain on the true path did not merge within kCap bytes (rare: ~2 per block on html): follow the path lane by
                    // lane on the scalar unit instead, walking such a chain on, whole wave, until it merges or leaves
                    active = 0;
                    entry = 0;
                    for (u32 k = 0, e = 0;;) {
                        active |= 1ull << k;
                        entry = lane == k ? e : entry;
                        u32 mk = read_lane(m, k), nk = read_lane(nx, k);
                        if (nk == 65u) {
                            DPROF_ADD(7, 1);
                            nk = 64u;
                            while (mk < L) {
                                const u32 v = bcast_first(c_V[mk >> 5]);
                                if ((v >> (mk & 31u)) & 1u) { nk = mk >> 5; break; }
                                if (lane == 0) atomicOr(&c_T[mk >> 5], 1u << (mk & 31u));
                                mk += bcast_first(tag_advance_staged(c_in + mk));
                                DPROF_ADD(8, 1);
                            }
                        }
                        if (nk >= 64u) { consumed = mk; break; }
                        e = mk;
                        k = nk;
                    }
                }
                // T: the true tag starts
                if ((active >> lane) & 1ull) {
                    const u32 own = V & ~((1u << (entry & 31u)) - 1u);
                    const u32 w0 = obase >> 5;
                    if (own) atomicOr(&c_T[lane], own);
#pragma unroll
                    for (u32 w = 0; w < kCap / 32; ++w) {
                        const u32 ow = c_O[w];
                        if (ow && w0 + w < SNP_WAVE) atomicOr(&c_T[w0 + w], ow);
                    }
                }
                lanes_sync_lds();
                const u32 Tw = c_T[lane];
                const u32 cnt = static_cast<u32>(__builtin_popcount(Tw));
                const u32 cincl = wave_inclusive_scan(cnt);
                ntok = read_lane(cincl, 63);
                lanes_sync_lds();                                       // (every read of c_in is done: the list overwrites it)
                u32 t = cincl - cnt, bits = Tw;
                while (bits) {
                    c_pos[t++] = static_cast<u16>(r0 + static_cast<u32>(__builtin_ctz(bits)));
                    bits &= bits - 1u;
                }
                lanes_sync_lds();
                emitted = (SNP_D_ABLATE & 32) ? ntok : 0;               // (ablation: build the tag lists only)
                pf_at = ~0u;
                DPROF_ADD(2, 1);                                        // super-windows
                DPROF_ADD(9, __builtin_popcountll(active));             // lanes on the true chain
                DPROF_TIME(11);                                         // chains, merge, tag list
            }
            // ---- one batch: the next <= 64 tags of the list ----
            const u32 t = emitted + lane;
            const bool have = t < ntok;
            const u32 pos = have ? c_pos[t] : 0u;
#if SNP_D_TOPWAIT
            const u64 q = pf_at == emitted ? q_pf : ld64u(src + wbase + pos);   // pos < L (idle lanes re-read position 0)
#else
            // The tag bytes were requested a batch ago (q_pf); only the first batch of a super-window loads them here.  That load's wait
            // is kept on ITS path: merged with the prefetched value at a join, the compiler drains vmcnt in EVERY batch -- and what is
            // still in flight at that point is the previous batch's write-out, so every batch waited ~1 k cycles for its stores to be
            // acknowledged (the "write-out" that cost 13 % in the ablations was this wait, not the stores).
            u64 q = q_pf;
            if (pf_at != emitted) {
                q = ld64u(src + wbase + pos);                           // pos < L (idle lanes re-read position 0)
                asm volatile("" : "+v"(q));
            }
#endif
            const u32 c = static_cast<u32>(q) & 0xffu;
            const u32 type = c & 3u;
            const u32 hi6 = c >> 2;
            const u32 b1234 = static_cast<u32>(q >> 8);
            const bool is_lit = type == 0;
            const bool long_lit = is_lit && hi6 >= 60;
            const u32 extra = is_lit ? (long_lit ? hi6 - 59 : 0u) : (type == 3 ? 4u : type);
            const u32 trailer = extra >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * extra);
            const u32 len = (long_lit ? trailer : (hi6 & (type == 1 ? 7u : 63u))) + (type == 1 ? 4u : 1u);
            const u32 off = is_lit ? 0u : (type == 1 ? (((c >> 5) << 8) | (b1234 & 0xffu)) : trailer);
            const u32 body = pos + 1u + extra;                          // a literal's bytes, from wbase
            const u32 olen = have ? len : 0u;
            const u32 incl = wave_inclusive_scan(olen);
            const u32 ostart = op + incl - olen;
            const u32 room = n - wbase - 16u;                           // lane_copy over-reads 15 bytes
            const bool lit_ok = ((len - 1u) < room) & (body <= room - len);
            const bool copy_ok = (off - 1u) < ostart;
            const bool ok = have & ((is_lit & lit_ok) | (!is_lit & copy_ok)) & (incl + 16u <= expected - op);
            const bool big = is_lit & (len > 64u);
            const u64 okm = ballot64(ok & !big & (incl <= SNP_D_STAGE));
            const u32 ne = okm == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~okm));
            if (ne == 0) {
                const u32 f0 = read_lane((ok ? 1u : 0u) | (big ? 2u : 0u), 0);
                if (f0 != 3u) {                                         // not ours: the serial loop decides, from this tag on
                    ip = wbase + read_lane(pos, 0);
                    emitted = ntok = consumed = 0;                      // (ip is final)
                    wbase = ip;
                    break;
                }
                const u32 l0 = read_lane(len, 0);
                wave_copy(dst + op, src + wbase + read_lane(body, 0), l0, lane);
                op += l0;
                emitted += 1;
                continue;
            }
            const bool act = lane < ne;
            const u32 mark = op;                                        // all output below it is complete
            const u32 span = read_lane(incl, ne - 1);
            const bool ready = act && (is_lit || (off >= len && ostart - off + len <= mark));
            DPROF_TIME(12);                                             // tag bytes (+ whatever the wave still waits for at the batch top), decode, prefix sum, checks
            if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pf_at = SNP_D_PF ? emitted + ne : ~0u;                      // the next batch's tag bytes travel with this batch's copies
            if (pf_at < ntok) q_pf = ld64u(src + wbase + (pf_at + lane < ntok ? c_pos[pf_at + lane] : 0u));
            u8* const my = c_stage + (ostart - mark);
            const u32 s_lo = ostart - off;
            // (timing-only ablation 64: every copy source pulled to within 1 KiB below the batch -- what if far back-references cost nothing?
            //  10.9 vs 11.2 ms, profiles/r03b_decode_far_source_ablation.jsonl: they nearly do already)
            if (ready && !(SNP_D_ABLATE & 1)) lane_copy2(my, is_lit ? src + wbase + body : dst + ((SNP_D_ABLATE & 64) ? max(s_lo, max(mark, 1024u) - 1024u) : s_lo), (SNP_D_ABLATE & 128) ? min(len, 16u) : len);   // (ablation 128: one 16-byte piece per tag, whatever its length)
            u64 pend = ballot64(act && !ready);
            DPROF_ADD(0, 1);
            DPROF_ADD(1, ne);
            DPROF_ADD(5, __builtin_popcountll(pend));
            DPROF_TIME(13);                                             // first pass: source loads, stage stores
            if (pend) {
                // second lane-parallel pass: sources inside the batch that no pending tag still has to write
                bool blocked = off < len || s_lo < mark;                // pattern copies and sources straddling `mark`: finished in order
                const bool mine = (pend >> lane) & 1ull;
                if (SNP_D_P2MIN > 2 && (pend & (pend - 1)) != 0 && static_cast<u32>(__builtin_popcountll(pend)) < SNP_D_P2MIN) blocked = true;   // too few for a pass of their own
                else if ((SNP_D_ABLATE & 16) == 0 && (pend & (pend - 1))) {
                    c_busy[lane] = 0ull;
                    lanes_sync_lds();
                    if (mine) {
                        const u32 r = ostart - mark, b0 = r & 63u;
                        const u64 mk = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
                        atomicOr(reinterpret_cast<unsigned long long*>(&c_busy[r >> 6]), static_cast<unsigned long long>(mk << b0));
                        if (b0 && (mk >> (64u - b0)))
                            atomicOr(reinterpret_cast<unsigned long long*>(&c_busy[(r >> 6) + 1]), static_cast<unsigned long long>(mk >> (64u - b0)));
                    }
                    lanes_sync_lds();
                    if (mine && !blocked) {
                        const u32 lo = s_lo - mark, b0 = lo & 63u;
                        const u64 mk = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
                        const u64 w0 = c_busy[lo >> 6], w1 = c_busy[(lo >> 6) + 1];
                        blocked = ((w0 & (mk << b0)) | (b0 ? (w1 & (mk >> (64u - b0))) : 0ull)) != 0ull;
                    }
                }
                const bool ready2 = SNP_D_PASS2 && mine && !blocked;
                lanes_sync_lds();
                if (ready2) lane_copy2(my, c_stage + (s_lo - mark), len);
                pend &= ~ballot64(ready2);
                DPROF_ADD(4, __builtin_popcountll(pend));
                // The rest in order, whole wave per tag, a byte per lane.  (This loop runs ~5 times per batch and is mostly scalar
                // work -- the busiest unit of this kernel -- so the common case, a source inside the stage, is kept to one
                // LDS read and one LDS write under one exec mask; LDS operations of a wave execute in order.)
                if (SNP_D_ABLATE & 2) pend = 0;
                while (pend) {
                    const u32 f = static_cast<u32>(__builtin_ctzll(pend));
                    pend &= pend - 1;
                    const u32 f_o = read_lane(ostart, f), f_off = read_lane(off, f), f_len = read_lane(len, f);
                    u32 sidx = lane;
                    if (f_off < f_len) {
#pragma unroll
                        for (int sh = 5; sh >= 0; --sh) {
                            const u32 tt = f_off << sh;
                            sidx = min(sidx, sidx - tt);
                        }
                    }
                    const u32 rel = f_o - mark;
                    if (rel >= f_off) {                                 // the whole source lies in this batch
                        if (lane < f_len) c_stage[rel + lane] = c_stage[rel - f_off + sidx];
                    } else {                                            // it starts before the batch: those bytes are in global memory
                        const u32 spos = f_o - f_off + sidx;
                        u32 byte = 0;
                        if (lane < f_len) {
                            if (spos < mark) byte = dst[spos];
                            else byte = c_stage[spos - mark];
                        }
                        if (lane < f_len) c_stage[rel + lane] = static_cast<u8>(byte);
                    }
                }
            }
            DPROF_TIME(14);                                             // second pass + in-order finish
            // the whole run, coalesced
            lanes_sync_lds();
            u8* const g = dst + mark;
            if (!(SNP_D_ABLATE & 4)) {
            for (u32 i = lane * 16; i + 16 <= span; i += SNP_WAVE * 16)
                *reinterpret_cast<snp_u128_unaligned*>(g + i) = *reinterpret_cast<const snp_u128_unaligned*>(c_stage + i);
            const u32 tail = span & ~15u;
            if (tail + lane < span) g[tail + lane] = c_stage[tail + lane];
            }
            lanes_sync_lds();
            op += span;
            emitted += ne;
            DPROF_TIME(15);                                             // write-out, until the stores are acknowledged
        }
        DPROF_FLUSH;
        w.wv = 0x80000000u;
    }

    // ---- FRONT = 5 (experiment, -DSNP_D_PC=1 builds only): the sub-chain front end as a PRODUCER / CONSUMER pair -- a workgroup of two wavefronts
    // shares one block: wavefront 0 parses super-window k + 1 (phases A, A', R, T -> tag list in LDS, double buffered) while wavefront 1 executes the
    // batches of super-window k; one workgroup barrier per super-window.  VERDICT r3's "second, cheaper experiment".  Same results as FRONT = 3.
#if SNP_D_PC
    if (FRONT == 5) {
        constexpr u32 kR = 32;
        constexpr u32 kW = SNP_WAVE * kR;
        constexpr u32 kCap = SNP_D_CAP;
        __shared__ __attribute__((aligned(16))) u8 c_in2[2][kW];        // staged input of a super-window, afterwards its tag list (u16 each)
        __shared__ __attribute__((aligned(16))) u8 c_stage[SNP_D_STAGE + 64];   // executor
        __shared__ u64 c_busy[65];                                      // executor: pending output bytes of a batch
        __shared__ __attribute__((aligned(16))) u8 c_pscratch[512 + SNP_WAVE * (kCap / 8)];   // parser: reach flags, entries, overrun bitmaps
        __shared__ u32 c_VT[2 * SNP_WAVE];                              // parser: V and T
        __shared__ u32 c_meta[2][4];                                    // per buffer: ntok, wbase, ended
        __shared__ u32 c_stop;
        const bool parser = (threadIdx.x >> 6) == 0;
        u32* const c_V = c_VT;
        u32* const c_T = c_VT + SNP_WAVE;
        const u32 r0 = kR * lane;
        if (threadIdx.x == 0) c_stop = 0;
        __syncthreads();
        if (parser) {
            u32 wbase = ip, consumed = 0, ntok = 0;
            for (u32 k = 0; st == SNP_OK; ++k) {
                u8* const c_in = c_in2[k & 1];
                u16* const c_pos = reinterpret_cast<u16*>(c_in);
                bool ended = false;
                {
                // ---- the next super-window ----
                ip = wbase + consumed;
                if (ip + 72 > n) { ended = true; }
                else {
                wbase = ip;
                const u32 avail = n - wbase;
                const u32 L = min(kW, avail) - 8u;                      // tags may start below L: their 8 bytes lie inside the staged input
                const u8* const wsrc = src + wbase;
                {
                    // 2 x 16 bytes per lane; a piece that would cross the end of the input is pulled back inside it (avail >= 72;
                    // the bytes it rewrites are the same bytes), pieces beyond it are not needed
                    const u32 oa = lane * 16u, ob = oa + 1024u;
                    const u32 la = min(oa, avail - 16u), lb = min(ob, avail - 16u);
                    const u32x4 va = ld128u(wsrc + la), vb = ld128u(wsrc + lb);
                    st128u(c_in + la, va);
                    st128u(c_in + lb, vb);
                }
                lanes_sync_lds();
                DPROF_TIME(10);                                         // input staged
                // A: the chain from the first byte of the lane's region
                u32 p = r0, V = 0;
                [[maybe_unused]] u32 trips = 0;
                {
                    const u32 lim = min(r0 + kR, L);
#if SNP_D_A2
                    // Two tags per trip when the first is a COPY (74 % of html tags): its successor can only start 2, 3 or 5 bytes on, so those
                    // three bytes are read together with the tag byte and the second advance costs no second LDS round trip
                    // (p < L = staged - 8: the reads stay inside the staged bytes).
                    while (p < lim) {
                        const u32 c0 = c_in[p], b2 = c_in[p + 2], b3 = c_in[p + 3], b5 = c_in[p + 5];
                        V |= 1u << (p - r0);
                        const u32 t0 = c0 & 3u;
                        if (__builtin_expect((c0 & 0xf3u) == 0xf0u, 0)) {       // literal with length bytes: the plain form
                            p += tag_advance_staged(c_in + p);
                        } else if (t0 == 0) {
                            p += (c0 >> 2) + 2u;
                        } else {
                            const u32 a0 = __builtin_amdgcn_ubfe(0x05030200u, 8u * t0, 8u);
                            const u32 p1 = p + a0;
                            const u32 c1 = t0 == 1 ? b2 : t0 == 2 ? b3 : b5;
                            p = p1;
                            if (p1 < lim) {
                                V |= 1u << (p1 - r0);
                                if (__builtin_expect((c1 & 0xf3u) == 0xf0u, 0)) p = p1 + tag_advance_staged(c_in + p1);
                                else p = p1 + ((c1 & 3u) ? __builtin_amdgcn_ubfe(0x05030200u, 8u * (c1 & 3u), 8u) : (c1 >> 2) + 2u);
                            }
                        }
                        DPROF_TRIP(trips);
                    }
#else
                    while (p < lim) {
                        V |= 1u << (p - r0);
                        p += tag_advance_staged(c_in + p);
                        DPROF_TRIP(trips);
                    }
#endif
                }
                DPROF_ADD_MAX(3, trips);                                // loop trips of phase A
                c_V[lane] = V;
                c_T[lane] = 0;
                lanes_sync_lds();
                // A': on past the region until the chain lands on a position its owner visited (one loop exit: the exec-mask
                // bookkeeping of a divergent loop is scalar work, and the scalar unit is the busiest one in this kernel)
                u32 nx = 64u;                                           // 64: the chain leaves the super-window, 65: no merge within kCap bytes
                u32* const c_O = reinterpret_cast<u32*>(c_pscratch + 512) + lane * (kCap / 32);   // overrun positions, a bit each, from obase
#pragma unroll                                                          // (in the idle stage: keeping them in registers costs 20 instructions per trip)
                for (u32 w = 0; w < kCap / 32; ++w) c_O[w] = 0;
                const u32 obase = p & ~(kR - 1u);
                for (bool go = p < L; go;) {
                    const u32 v = c_V[p >> 5];
                    const u32 adv = tag_advance_staged(c_in + p);
                    const u32 rel = p - obase;
                    const bool hit = (v >> (p & 31u)) & 1u;
                    const bool stop = hit | (rel >= kCap);
                    nx = stop ? (hit ? p >> 5 : 65u) : nx;
                    atomicOr(&c_O[min(rel >> 5, kCap / 32 - 1)], stop ? 0u : 1u << (rel & 31u));   // (unconditional: no exec-mask bookkeeping)
                    p = stop ? p : p + adv;
                    go = !stop & (p < L);
                    DPROF_TRIP(trips);
                }
                const u32 m = p;                                        // where the chain merged, gave up or left
                DPROF_ADD_MAX(6, trips);                                // ... of A and A' together
                // R: the lanes on the true chain = the lanes reachable from lane 0 along nx, by pointer doubling (flags through
                // LDS: a scatter needs its senders masked); each of them tells its successor where it enters.  ~70 wave
                // instructions instead of a scalar walk of ~14 per lane on the chain (~46 of them on html).
                u64 active;
                u32 entry = 0;
                {
                    u8* const c_reach = c_pscratch;                        // (the stage is idle while a super-window is built)
                    u32* const c_entry = reinterpret_cast<u32*>(c_pscratch + SNP_WAVE);
                    u32 hop = nx;
                    bool reached = lane == 0;
                    c_reach[lane] = reached ? 1 : 0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        lanes_sync_lds();
                        if (reached && hop < 64u) c_reach[hop] = 1;
                        lanes_sync_lds();
                        reached = c_reach[lane] != 0;
                        const u32 h2 = bperm(hop, hop);
                        hop = hop < 64u ? h2 : hop;
                    }
                    if (reached && nx < 64u) c_entry[nx] = m;
                    lanes_sync_lds();
                    if (lane) entry = c_entry[lane];
                    lanes_sync_lds();
                    active = ballot64(reached);
                    const u64 ends = ballot64(reached && nx == 64u);    // the lane whose chain leaves the super-window, if the chain gets there
                    consumed = ends ? read_lane(m, static_cast<u32>(__builtin_ctzll(ends))) : 0u;
                }
                if (ballot64(((active >> lane) & 1ull) && nx == 65u)) {
                    // a chain on the true path did not merge within kCap bytes (rare: ~2 per block on html): follow the path lane by
                    // lane on the scalar unit instead, walking such a chain on, whole wave, until it merges or leaves
                    active = 0;
                    entry = 0;
                    for (u32 k = 0, e = 0;;) {
                        active |= 1ull << k;
                        entry = lane == k ? e : entry;
                        u32 mk = read_lane(m, k), nk = read_lane(nx, k);
                        if (nk == 65u) {
                            DPROF_ADD(7, 1);
                            nk = 64u;
                            while (mk < L) {
                                const u32 v = bcast_first(c_V[mk >> 5]);
                                if ((v >> (mk & 31u)) & 1u) { nk = mk >> 5; break; }
                                if (lane == 0) atomicOr(&c_T[mk >> 5], 1u << (mk & 31u));
                                mk += bcast_first(tag_advance_staged(c_in + mk));
                                DPROF_ADD(8, 1);
                            }
                        }
                        if (nk >= 64u) { consumed = mk; break; }
                        e = mk;
                        k = nk;
                    }
                }
                // T: the true tag starts
                if ((active >> lane) & 1ull) {
                    const u32 own = V & ~((1u << (entry & 31u)) - 1u);
                    const u32 w0 = obase >> 5;
                    if (own) atomicOr(&c_T[lane], own);
#pragma unroll
                    for (u32 w = 0; w < kCap / 32; ++w) {
                        const u32 ow = c_O[w];
                        if (ow && w0 + w < SNP_WAVE) atomicOr(&c_T[w0 + w], ow);
                    }
                }
                lanes_sync_lds();
                const u32 Tw = c_T[lane];
                const u32 cnt = static_cast<u32>(__builtin_popcount(Tw));
                const u32 cincl = wave_inclusive_scan(cnt);
                ntok = read_lane(cincl, 63);
                lanes_sync_lds();                                       // (every read of c_in is done: the list overwrites it)
                u32 t = cincl - cnt, bits = Tw;
                while (bits) {
                    c_pos[t++] = static_cast<u16>(r0 + static_cast<u32>(__builtin_ctz(bits)));
                    bits &= bits - 1u;
                }
                lanes_sync_lds();
                DPROF_ADD(2, 1);                                        // super-windows
                DPROF_ADD(9, __builtin_popcountll(active));             // lanes on the true chain
                DPROF_TIME(11);                                         // chains, merge, tag list
                }
                }
                if (lane == 0) { c_meta[k & 1][0] = ended ? 0u : ntok; c_meta[k & 1][1] = ended ? ip : wbase; c_meta[k & 1][2] = ended ? 1u : 0u; }
                __syncthreads();                                        // window k is ready; the executor is done with buffer (k + 1) & 1
                if (ended || c_stop) break;
            }
            return;                                                     // (the executor owns the tail, the status and the length)
        }
        {
            u32 wbase = ip, ntok = 0, emitted = 0;
            u64 q_pf = 0;
            u32 pf_at = ~0u;
            bool e_stop = false;
            DPROF_T0
            for (u32 k = 0; st == SNP_OK; ++k) {                        // (a block whose preamble failed: neither wavefront enters its loop)
                __syncthreads();                                        // the parser has published window k
                const u32 m_ntok = c_meta[k & 1][0], m_wbase = c_meta[k & 1][1], m_end = c_meta[k & 1][2];
                u16* const c_pos = reinterpret_cast<u16*>(c_in2[k & 1]);
                ip = m_wbase;
                if (m_end) break;
                if (op >= expected) { e_stop = true; }
                wbase = m_wbase;
                ntok = m_ntok;
                emitted = 0;
                pf_at = ~0u;
                while (!e_stop && emitted < ntok) {
            // ---- one batch: the next <= 64 tags of the list ----
            const u32 t = emitted + lane;
            const bool have = t < ntok;
            const u32 pos = have ? c_pos[t] : 0u;
#if SNP_D_TOPWAIT
            const u64 q = pf_at == emitted ? q_pf : ld64u(src + wbase + pos);   // pos < L (idle lanes re-read position 0)
#else
            // The tag bytes were requested a batch ago (q_pf); only the first batch of a super-window loads them here.  That load's wait
            // is kept on ITS path: merged with the prefetched value at a join, the compiler drains vmcnt in EVERY batch -- and what is
            // still in flight at that point is the previous batch's write-out, so every batch waited ~1 k cycles for its stores to be
            // acknowledged (the "write-out" that cost 13 % in the ablations was this wait, not the stores).
            u64 q = q_pf;
            if (pf_at != emitted) {
                q = ld64u(src + wbase + pos);                           // pos < L (idle lanes re-read position 0)
                asm volatile("" : "+v"(q));
            }
#endif
            const u32 c = static_cast<u32>(q) & 0xffu;
            const u32 type = c & 3u;
            const u32 hi6 = c >> 2;
            const u32 b1234 = static_cast<u32>(q >> 8);
            const bool is_lit = type == 0;
            const bool long_lit = is_lit && hi6 >= 60;
            const u32 extra = is_lit ? (long_lit ? hi6 - 59 : 0u) : (type == 3 ? 4u : type);
            const u32 trailer = extra >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * extra);
            const u32 len = (long_lit ? trailer : (hi6 & (type == 1 ? 7u : 63u))) + (type == 1 ? 4u : 1u);
            const u32 off = is_lit ? 0u : (type == 1 ? (((c >> 5) << 8) | (b1234 & 0xffu)) : trailer);
            const u32 body = pos + 1u + extra;                          // a literal's bytes, from wbase
            const u32 olen = have ? len : 0u;
            const u32 incl = wave_inclusive_scan(olen);
            const u32 ostart = op + incl - olen;
            const u32 room = n - wbase - 16u;                           // lane_copy over-reads 15 bytes
            const bool lit_ok = ((len - 1u) < room) & (body <= room - len);
            const bool copy_ok = (off - 1u) < ostart;
            const bool ok = have & ((is_lit & lit_ok) | (!is_lit & copy_ok)) & (incl + 16u <= expected - op);
            const bool big = is_lit & (len > 64u);
            const u64 okm = ballot64(ok & !big & (incl <= SNP_D_STAGE));
            const u32 ne = okm == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~okm));
            if (ne == 0) {
                const u32 f0 = read_lane((ok ? 1u : 0u) | (big ? 2u : 0u), 0);
                if (f0 != 3u) {                                         // not ours: the serial loop decides, from this tag on
                    ip = wbase + read_lane(pos, 0);
                    e_stop = true;                                      // (ip is final: the serial loop takes over; the parser is told at the next barrier)
                    break;
                }
                const u32 l0 = read_lane(len, 0);
                wave_copy(dst + op, src + wbase + read_lane(body, 0), l0, lane);
                op += l0;
                emitted += 1;
                continue;
            }
            const bool act = lane < ne;
            const u32 mark = op;                                        // all output below it is complete
            const u32 span = read_lane(incl, ne - 1);
            const bool ready = act && (is_lit || (off >= len && ostart - off + len <= mark));
            DPROF_TIME(12);                                             // tag bytes (+ whatever the wave still waits for at the batch top), decode, prefix sum, checks
            if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pf_at = SNP_D_PF ? emitted + ne : ~0u;                      // the next batch's tag bytes travel with this batch's copies
            if (pf_at < ntok) q_pf = ld64u(src + wbase + (pf_at + lane < ntok ? c_pos[pf_at + lane] : 0u));
            u8* const my = c_stage + (ostart - mark);
            const u32 s_lo = ostart - off;
            // (timing-only ablation 64: every copy source pulled to within 1 KiB below the batch -- what if far back-references cost nothing?
            //  10.9 vs 11.2 ms, profiles/r03b_decode_far_source_ablation.jsonl: they nearly do already)
            if (ready && !(SNP_D_ABLATE & 1)) lane_copy2(my, is_lit ? src + wbase + body : dst + ((SNP_D_ABLATE & 64) ? max(s_lo, max(mark, 1024u) - 1024u) : s_lo), (SNP_D_ABLATE & 128) ? min(len, 16u) : len);   // (ablation 128: one 16-byte piece per tag, whatever its length)
            u64 pend = ballot64(act && !ready);
            DPROF_ADD(0, 1);
            DPROF_ADD(1, ne);
            DPROF_ADD(5, __builtin_popcountll(pend));
            DPROF_TIME(13);                                             // first pass: source loads, stage stores
            if (pend) {
                // second lane-parallel pass: sources inside the batch that no pending tag still has to write
                bool blocked = off < len || s_lo < mark;                // pattern copies and sources straddling `mark`: finished in order
                const bool mine = (pend >> lane) & 1ull;
                if ((SNP_D_ABLATE & 16) == 0 && (pend & (pend - 1))) {
                    c_busy[lane] = 0ull;
                    lanes_sync_lds();
                    if (mine) {
                        const u32 r = ostart - mark, b0 = r & 63u;
                        const u64 mk = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
                        atomicOr(reinterpret_cast<unsigned long long*>(&c_busy[r >> 6]), static_cast<unsigned long long>(mk << b0));
                        if (b0 && (mk >> (64u - b0)))
                            atomicOr(reinterpret_cast<unsigned long long*>(&c_busy[(r >> 6) + 1]), static_cast<unsigned long long>(mk >> (64u - b0)));
                    }
                    lanes_sync_lds();
                    if (mine && !blocked) {
                        const u32 lo = s_lo - mark, b0 = lo & 63u;
                        const u64 mk = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
                        const u64 w0 = c_busy[lo >> 6], w1 = c_busy[(lo >> 6) + 1];
                        blocked = ((w0 & (mk << b0)) | (b0 ? (w1 & (mk >> (64u - b0))) : 0ull)) != 0ull;
                    }
                }
                const bool ready2 = SNP_D_PASS2 && mine && !blocked;
                lanes_sync_lds();
                if (ready2) lane_copy2(my, c_stage + (s_lo - mark), len);
                pend &= ~ballot64(ready2);
                DPROF_ADD(4, __builtin_popcountll(pend));
                // The rest in order, whole wave per tag, a byte per lane.  (This loop runs ~5 times per batch and is mostly scalar
                // work -- the busiest unit of this kernel -- so the common case, a source inside the stage, is kept to one
                // LDS read and one LDS write under one exec mask; LDS operations of a wave execute in order.)
                if (SNP_D_ABLATE & 2) pend = 0;
                while (pend) {
                    const u32 f = static_cast<u32>(__builtin_ctzll(pend));
                    pend &= pend - 1;
                    const u32 f_o = read_lane(ostart, f), f_off = read_lane(off, f), f_len = read_lane(len, f);
                    u32 sidx = lane;
                    if (f_off < f_len) {
#pragma unroll
                        for (int sh = 5; sh >= 0; --sh) {
                            const u32 tt = f_off << sh;
                            sidx = min(sidx, sidx - tt);
                        }
                    }
                    const u32 rel = f_o - mark;
                    if (rel >= f_off) {                                 // the whole source lies in this batch
                        if (lane < f_len) c_stage[rel + lane] = c_stage[rel - f_off + sidx];
                    } else {                                            // it starts before the batch: those bytes are in global memory
                        const u32 spos = f_o - f_off + sidx;
                        u32 byte = 0;
                        if (lane < f_len) {
                            if (spos < mark) byte = dst[spos];
                            else byte = c_stage[spos - mark];
                        }
                        if (lane < f_len) c_stage[rel + lane] = static_cast<u8>(byte);
                    }
                }
            }
            DPROF_TIME(14);                                             // second pass + in-order finish
            // the whole run, coalesced
            lanes_sync_lds();
            u8* const g = dst + mark;
            if (!(SNP_D_ABLATE & 4)) {
            for (u32 i = lane * 16; i + 16 <= span; i += SNP_WAVE * 16)
                *reinterpret_cast<snp_u128_unaligned*>(g + i) = *reinterpret_cast<const snp_u128_unaligned*>(c_stage + i);
            const u32 tail = span & ~15u;
            if (tail + lane < span) g[tail + lane] = c_stage[tail + lane];
            }
            lanes_sync_lds();
            op += span;
            emitted += ne;
            DPROF_TIME(15);                                             // write-out, until the stores are acknowledged
                }
                if (e_stop) {
                    if (lane == 0) c_stop = 1;
                    __syncthreads();                                    // the barrier the parser is heading for: it reads c_stop behind it and leaves
                    break;
                }
            }
            DPROF_FLUSH;
            w.wv = 0x80000000u;
        }
    }
#endif

    // ---- sub-chain parse feeding OUTPUT-granular execution through a ring of recent output in LDS (FRONT = 4) -----------------------
    // The tag-per-lane batch above moves every tag with 1-4 unaligned 16-byte vector-memory loads and unaligned LDS stores: the texture
    // path is busy 77 % of the kernel and an unaligned wide LDS access costs the pipe 1-2 cycles per LANE.  Here the lanes own OUTPUT
    // BYTES instead.  The last kRing (2 KiB) bytes of the block's output live in an LDS ring (index = output position + g0, g0 = the block's
    // misalignment in global memory, so that 16-byte units of the ring are 16-byte units of the output); a batch is still the next <= 64
    // tags of the list, but it executes in sub-steps of 64 consecutive output bytes, one byte per lane:
    //   * which tag a byte belongs to: every tag marks the byte before its first one in a bitmap of the batch's span, and a byte's tag is
    //     the popcount of the marks below it (one broadcast 8-byte LDS read + v_mbcnt per sub-step, no scan);
    //   * where the byte comes from: ONE dword per tag (`rec`): a copy inside the ring holds -offset, everything else the distance from
    //     the byte's position to its source in a virtual address space that is simply this wavefront's LDS -- the staged input (literals
    //     are read where the parse left them) and a buffer of FAR pieces (copies older than the ring and literals beyond the staged
    //     input: 16-byte pieces loaded from global memory once per batch, tag per lane, the only source loads left on the texture path);
    //   * bytes whose source lies inside their own sub-step are resolved by pointer doubling over the lanes (ds_bpermute; a lane whose
    //     source is outside the sub-step is a root; 81 % of the sub-steps of the html-like workload have no such byte and skip this);
    //   * byte-wide LDS accesses are never serialised (profiles/r04a_microbench_lds_gather.jsonl: ds_read_u8 4.1, ds_write_b8 5.3 cycles
    //     per wave-instruction per CU in run-structured gathers);
    //   * the ring leaves for global memory in aligned 16-byte units, one coalesced store per KiB, after every batch.
    // Everything irregular is left to the serial loop below exactly as in FRONT = 3, after the ring has been flushed.
    if (FRONT == 4) {
        constexpr u32 kR = 32;
        constexpr u32 kW = SNP_WAVE * kR;
        constexpr u32 kCap = 128;
        constexpr u32 kRing = SNP_D_RING;                               // bytes of recent output kept in LDS (power of two)
        constexpr u32 kSpan = SNP_D_RING_SPAN;                          // most output bytes of one batch (<= kRing - 158: a far copy's pieces then lie below the written-out frontier; multiple of 64; marks fit a dword per lane)
        constexpr u32 kIn = kRing;                                      // virtual addresses = offsets into c_all
        constexpr u32 kFar = kIn + kW + 16;
        constexpr u32 kPos = kFar + 1024;
        constexpr u32 kMisc = kPos + kW;
        constexpr u32 kTotal = kMisc + 208 * 4;
        static_assert((kRing & (kRing - 1)) == 0 && kSpan + 158 <= kRing && kSpan % 64 == 0 && kSpan + 64 <= 2048, "ring geometry");
        __shared__ __attribute__((aligned(16))) u8 c_all[kTotal];
        u8* const c_ring = c_all;
        u8* const c_in = c_all + kIn;
        u8* const c_far = c_all + kFar;                                 // batches: far pieces; while a super-window is parsed: the overrun bitmaps
        u16* const c_pos = reinterpret_cast<u16*>(c_all + kPos);
        u32* const c_misc = reinterpret_cast<u32*>(c_all + kMisc);
        u32* const c_V = c_misc;                                        // parse: V, T, entry, reach
        u32* const c_T = c_misc + 64;
        u32* const c_entry = c_misc + 128;
        u8* const c_reach = reinterpret_cast<u8*>(c_misc + 192);
        u32* const c_bits = c_misc;                                     // batches: tag-start marks (64 dwords), one rec per tag
        u32* const c_rec = c_misc + 64;
        const u32 r0 = kR * lane;
        const u32 g0 = static_cast<u32>(reinterpret_cast<uintptr_t>(dst) & 15u);
        u8* const gbase = dst - g0;                                     // biased positions: pb = output position + g0
        u32 wbase = ip, ntok = 0, emitted = 0, consumed = 0, staged = 0;
        u32 wo_b = g0;                                                  // biased position up to which the output has left for global memory
        u32 fence_b = 0;                                                // ... and is known to have arrived there (FENCED)
        // ring -> global memory: whole 16-byte units below op (final: every byte below op)
        auto write_out = [&](bool final) {
            const u32 lim = op + g0;
            if (wo_b & 15u) {                                           // the block's first unit (g0 != 0) or the unit a long literal ended in
                const u32 up = (wo_b + 15u) & ~15u;
                const u32 e = min(up, lim);
                if (!final && e != up) return;
                if (wo_b + lane < e) gbase[wo_b + lane] = c_ring[(wo_b + lane) & (kRing - 1u)];
                wo_b = e;
            }
            const u32 lim16 = lim & ~15u;
            for (u32 u = wo_b + 16u * lane; u < lim16; u += SNP_WAVE * 16u)
                *reinterpret_cast<u32x4*>(gbase + u) = *reinterpret_cast<const u32x4*>(c_ring + (u & (kRing - 1u)));
            if (lim16 > wo_b) wo_b = lim16;
            if (final && wo_b < lim) {
                if (wo_b + lane < lim) gbase[wo_b + lane] = c_ring[(wo_b + lane) & (kRing - 1u)];
                wo_b = lim;
            }
        };
        // One decoded batch: the tags first .. first + 63 of the list, their output starting at position opb.  ne = how many of them the
        // batch takes (it ends before the first tag that is malformed, a literal of more than 64 bytes, within 16 bytes of the block's
        // end, beyond kSpan output bytes or out of far units); ne == 0: f0 says what tag 0 is (bit 0 valid, bit 1 a long literal).
        // The far pieces of the batch are REQUESTED here (fp0..fp3) and stored to c_far when the batch executes.
        struct RingBatch {
            u32 ne, span, f0, pos0, len0, body0;                        // wave-uniform
            bool late;                                                  // decoded AHEAD and a far copy's source has not left the ring yet: decode again later
            u32 D, rel, len, u0;                                        // per lane (= per tag)
            bool act, farl;
            u32x4 fp0, fp1, fp2, fp3;
        };
        auto decode = [&](const u32 first, const u32 opb, RingBatch& B, const bool ahead) {
            const u32 t = first + lane;
            const bool have = t < ntok;
            const u32 pos = have ? c_pos[t] : 0u;
            u64 q;
            {
                const u32 pa = pos & ~3u;                               // two aligned dwords of the staged input (pos < staged - 8)
                const u32 d0 = *reinterpret_cast<const u32*>(c_in + pa), d1 = *reinterpret_cast<const u32*>(c_in + pa + 4u);
                q = ((static_cast<u64>(d1) << 32) | d0) >> (8u * (pos & 3u));
            }
            const u32 c = static_cast<u32>(q) & 0xffu;
            const u32 type = c & 3u;
            const u32 hi6 = c >> 2;
            const u32 b1234 = static_cast<u32>(q >> 8);
            const bool is_lit = type == 0;
            const bool long_lit = is_lit && hi6 >= 60;
            const u32 extra = is_lit ? (long_lit ? hi6 - 59 : 0u) : (type == 3 ? 4u : type);
            const u32 trailer = extra >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * extra);
            const u32 len = (long_lit ? trailer : (hi6 & (type == 1 ? 7u : 63u))) + (type == 1 ? 4u : 1u);
            const u32 off = is_lit ? 0u : (type == 1 ? (((c >> 5) << 8) | (b1234 & 0xffu)) : trailer);
            const u32 body = pos + 1u + extra;
            const u32 olen = have ? len : 0u;
            const u32 incl = wave_inclusive_scan(olen);
            const u32 ostart = opb + incl - olen;
            const u32 room = n - wbase - 16u;                           // far pieces over-read 15 bytes
            const bool lit_ok = ((len - 1u) < room) & (body <= room - len);
            const bool copy_ok = (off - 1u) < ostart;
            const bool ok = have & ((is_lit & lit_ok) | (!is_lit & copy_ok)) & (incl + 16u <= expected - opb);
            const bool big = is_lit & (len > 64u);
            const bool cand = ok & !big;
            // far tags: their bytes come from global memory, in 16-byte pieces, into c_far (64 units of 16 bytes per batch)
            const bool isfar = cand & (is_lit ? body + len > staged : off > kRing - 64u);
            const u32 units = (len + 15u) >> 4;
            const u64 f1 = ballot64(isfar), f2 = ballot64(isfar & (len > 16u)), f3 = ballot64(isfar & (len > 32u)), f4 = ballot64(isfar & (len > 48u));
            const u32 u0 = __builtin_amdgcn_mbcnt_hi(static_cast<u32>(f1 >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(f1), 0u)) +
                           __builtin_amdgcn_mbcnt_hi(static_cast<u32>(f2 >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(f2), 0u)) +
                           __builtin_amdgcn_mbcnt_hi(static_cast<u32>(f3 >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(f3), 0u)) +
                           __builtin_amdgcn_mbcnt_hi(static_cast<u32>(f4 >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<u32>(f4), 0u));
            const bool fits = !isfar | (u0 + units <= 64u);
            const u64 okm = ballot64(cand & fits & (incl <= kSpan));
            const u32 ne = okm == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~okm));
            B.ne = ne;
            B.f0 = read_lane((ok ? 1u : 0u) | (big ? 2u : 0u), 0);
            B.pos0 = read_lane(pos, 0);
            B.len0 = read_lane(len, 0);
            B.body0 = read_lane(body, 0);
            B.span = ne ? read_lane(incl, ne ? ne - 1u : 0u) : 0u;
            const bool act = lane < ne;
            const u32 rel = ostart - opb;
            const u32 pb0 = opb + g0;
            B.act = act;
            B.rel = rel;
            B.len = len;
            B.u0 = u0;
            // byte at biased position pb of this tag: ring copies read the ring at (pb + D) & (kRing - 1), D = -off; everything else
            // reads virtual address (pb + D) & 0xfffff, bit 30 of D set
            const u32 va = isfar ? kFar + 16u * u0 : kIn + body;
            B.D = (is_lit | isfar) ? (((va - rel - pb0) & 0xfffffu) | 0x40000000u) : 0u - off;
            const bool farl = act & isfar;
            B.farl = farl;
            const u8* const fsrc = is_lit ? src + wbase + body : dst + (ostart - off);
            // A far copy reads global memory this wavefront wrote.  (1) The bytes must have LEFT the ring: a batch decoded here for itself
            // reads at most kRing - 64 - 143 bytes below its own start, always below wo_b (kSpan + 158 <= kRing); a batch decoded AHEAD (while the
            // batch before it has not executed) can reach into that batch's output when both are long -- then it is not taken ahead (`late`),
            // and is decoded again when its turn comes.  (2) FENCED: the stores must also have ARRIVED (in-order vector memory is not relied on).
            const u32 need = ostart + g0 - off + len + 15u;             // biased end of the bytes the pieces read (15 over-read)
            B.late = ahead && ballot64(farl & !is_lit & (need > wo_b)) != 0ull;
            if (FENCED && !B.late) {
                if (ballot64(farl & !is_lit & (need > fence_b))) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    fence_b = wo_b;
                }
            }
            if (farl && !B.late) {
                B.fp0 = ld128u(fsrc);
                if (len > 16u) B.fp1 = ld128u(fsrc + 16);
                if (len > 32u) B.fp2 = ld128u(fsrc + 32);
                if (len > 48u) B.fp3 = ld128u(fsrc + 48);
            }
        };
        // marks, recs and far pieces of a decoded batch -> LDS (the arrays the sub-steps read)
        auto install = [&](const RingBatch& B) {
            c_bits[lane] = 0u;
            lanes_sync_lds();
            if (B.act && B.rel) atomicOr(&c_bits[(B.rel - 1u) >> 5], 1u << ((B.rel - 1u) & 31u));
            if (B.act) c_rec[lane] = B.D;
            if (B.farl) {
                u8* const fd = c_far + 16u * B.u0;
                *reinterpret_cast<u32x4*>(fd) = B.fp0;
                if (B.len > 16u) *reinterpret_cast<u32x4*>(fd + 16) = B.fp1;
                if (B.len > 32u) *reinterpret_cast<u32x4*>(fd + 32) = B.fp2;
                if (B.len > 48u) *reinterpret_cast<u32x4*>(fd + 48) = B.fp3;
            }
            lanes_sync_lds();
        };
        u32 cur_ne = 0, cur_span = 0;
        bool have_cur = false;
        DPROF_T0
        while (st == SNP_OK) {
            if (emitted == ntok) {
                // ---- the next super-window (phases A, A', R, T as in FRONT = 3; the staged input stays: literals are read from it) ----
                ip = wbase + consumed;
                if (ip + 72 > n || op >= expected) break;
                wbase = ip;
                const u32 avail = n - wbase;
                staged = min(kW, avail);
                const u32 L = staged - 8u;
                const u8* const wsrc = src + wbase;
                {
                    const u32 oa = lane * 16u, ob = oa + 1024u;
                    const u32 la = min(oa, avail - 16u), lb = min(ob, avail - 16u);
                    const u32x4 va = ld128u(wsrc + la), vb = ld128u(wsrc + lb);
                    st128u(c_in + la, va);
                    st128u(c_in + lb, vb);
                }
                lanes_sync_lds();
                DPROF_TIME(10);                                         // input staged
                u32 p = r0, V = 0;
                {
                    const u32 lim = min(r0 + kR, L);
                    while (p < lim) {
                        V |= 1u << (p - r0);
                        p += tag_advance_staged(c_in + p);
                    }
                }
                c_V[lane] = V;
                c_T[lane] = 0;
                lanes_sync_lds();
                u32 nx = 64u;
                u32* const c_O = reinterpret_cast<u32*>(c_far) + lane * (kCap / 32);
#pragma unroll
                for (u32 w = 0; w < kCap / 32; ++w) c_O[w] = 0;
                const u32 obase = p & ~(kR - 1u);
                for (bool go = p < L; go;) {
                    const u32 v = c_V[p >> 5];
                    const u32 adv = tag_advance_staged(c_in + p);
                    const u32 rel = p - obase;
                    const bool hit = (v >> (p & 31u)) & 1u;
                    const bool stop = hit | (rel >= kCap);
                    nx = stop ? (hit ? p >> 5 : 65u) : nx;
                    atomicOr(&c_O[min(rel >> 5, kCap / 32 - 1)], stop ? 0u : 1u << (rel & 31u));
                    p = stop ? p : p + adv;
                    go = !stop & (p < L);
                }
                const u32 m = p;
                u64 active;
                u32 entry = 0;
                {
                    u32 hop = nx;
                    bool reached = lane == 0;
                    c_reach[lane] = reached ? 1 : 0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        lanes_sync_lds();
                        if (reached && hop < 64u) c_reach[hop] = 1;
                        lanes_sync_lds();
                        reached = c_reach[lane] != 0;
                        const u32 h2 = bperm(hop, hop);
                        hop = hop < 64u ? h2 : hop;
                    }
                    if (reached && nx < 64u) c_entry[nx] = m;
                    lanes_sync_lds();
                    if (lane) entry = c_entry[lane];
                    lanes_sync_lds();
                    active = ballot64(reached);
                    const u64 ends = ballot64(reached && nx == 64u);
                    consumed = ends ? read_lane(m, static_cast<u32>(__builtin_ctzll(ends))) : 0u;
                }
                if (ballot64(((active >> lane) & 1ull) && nx == 65u)) {
                    active = 0;
                    entry = 0;
                    for (u32 k = 0, e = 0;;) {
                        active |= 1ull << k;
                        entry = lane == k ? e : entry;
                        u32 mk = read_lane(m, k), nk = read_lane(nx, k);
                        if (nk == 65u) {
                            nk = 64u;
                            while (mk < L) {
                                const u32 v = bcast_first(c_V[mk >> 5]);
                                if ((v >> (mk & 31u)) & 1u) { nk = mk >> 5; break; }
                                if (lane == 0) atomicOr(&c_T[mk >> 5], 1u << (mk & 31u));
                                mk += bcast_first(tag_advance_staged(c_in + mk));
                            }
                        }
                        if (nk >= 64u) { consumed = mk; break; }
                        e = mk;
                        k = nk;
                    }
                }
                if ((active >> lane) & 1ull) {
                    const u32 own = V & ~((1u << (entry & 31u)) - 1u);
                    const u32 w0 = obase >> 5;
                    if (own) atomicOr(&c_T[lane], own);
#pragma unroll
                    for (u32 w = 0; w < kCap / 32; ++w) {
                        const u32 ow = c_O[w];
                        if (ow && w0 + w < SNP_WAVE) atomicOr(&c_T[w0 + w], ow);
                    }
                }
                lanes_sync_lds();
                const u32 Tw = c_T[lane];
                const u32 cnt = static_cast<u32>(__builtin_popcount(Tw));
                const u32 cincl = wave_inclusive_scan(cnt);
                ntok = read_lane(cincl, 63);
                u32 t = cincl - cnt, bits = Tw;
                while (bits) {
                    c_pos[t++] = static_cast<u16>(r0 + static_cast<u32>(__builtin_ctz(bits)));
                    bits &= bits - 1u;
                }
                lanes_sync_lds();
                emitted = 0;
                DPROF_ADD(2, 1);
                DPROF_TIME(11);                                         // chains, merge, tag list
            }
            // ---- batches: the next <= 64 tags of the list each.  The NEXT batch is decoded, and its far pieces requested, before this
            //      batch's sub-steps run, and installed (marks, recs, far pieces -> LDS) right after them: the one global round trip of a
            //      batch is spent under the sub-steps of the batch before ----
            if (!have_cur) {
                RingBatch cur;
                write_out(false);                                       // (the batch before this one leaves the ring first: this batch's far copies may read it)
                decode(emitted, op, cur, false);
                if (cur.ne == 0) {
                    write_out(true);                                    // the ring holds the newest bytes: global memory must, too
                    if (cur.f0 != 3u) {                                 // not ours: the serial loop decides, from this tag on
                        ip = wbase + cur.pos0;
                        emitted = ntok = consumed = 0;
                        wbase = ip;
                        break;
                    }
                    // a literal of more than 64 bytes: input -> output directly, and its last bytes into the ring
                    const u32 l0 = cur.len0;
                    const u8* const ls = src + wbase + cur.body0;
                    wave_copy(dst + op, ls, l0, lane);
                    const u32 keep = min(l0, kRing);
                    const u32 pb = op + g0 + (l0 - keep);
                    for (u32 i = lane; i < keep; i += SNP_WAVE) c_ring[(pb + i) & (kRing - 1u)] = ls[l0 - keep + i];
                    lanes_sync_lds();
                    op += l0;
                    wo_b = op + g0;
                    emitted += 1;
                    continue;
                }
                install(cur);
                cur_ne = cur.ne;
                cur_span = cur.span;
                have_cur = true;
                DPROF_TIME(12);                                         // (first batch of a super-window: decode + install, far round trip exposed)
            }
            const u32 span = cur_span;
            const u32 pb0 = op + g0;
            const u32 bw = c_bits[lane];                                // the marks, a dword per lane: sub-steps pick theirs with v_readlane
            DPROF_ADD(0, 1);
            DPROF_ADD(1, cur_ne);
            write_out(false);                                           // everything before this batch
            DPROF_TIME(15);
            RingBatch nxt;
            bool have_nxt = false;
            if (emitted + cur_ne < ntok) {
                decode(emitted + cur_ne, op + span, nxt, true);
                have_nxt = nxt.ne != 0 && !nxt.late;
            }
            DPROF_TIME(13);                                             // the next batch: tag bytes, decode, prefix sum, checks, far requests
            // sub-steps of 64 output bytes
            u32 tbase = 0;                                              // tag (index in the batch) of the sub-step's first byte
            auto tag_of = [&](u32 sb) {
                const u32 w_lo = read_lane(bw, sb >> 5), w_hi = read_lane(bw, (sb >> 5) + 1u);
                const u32 ti = tbase + __builtin_amdgcn_mbcnt_hi(w_hi, __builtin_amdgcn_mbcnt_lo(w_lo, 0u));
                tbase += static_cast<u32>(__builtin_popcount(w_lo) + __builtin_popcount(w_hi));
                return ti;
            };
            u32 Dn = c_rec[tag_of(0)];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Dn));           // (nothing is in flight at the loop's head: its waits are its own)
            for (u32 sb = 0; sb < span; sb += SNP_WAVE) {
                const u32 D = Dn;
                Dn = c_rec[tag_of(min(sb + SNP_WAVE, kSpan))];          // (the next sub-step's recs travel with this one's bytes; past the span: any rec)
                const u32 pr = sb + lane;
                const bool in = pr < span;
                const u32 pb = pb0 + pr;
                const bool ring = static_cast<i32>(D) < 0;
                const u32 a = (pb + D) & (ring ? kRing - 1u : 0xfffffu);
                const i32 ptr = static_cast<i32>(lane) + static_cast<i32>(D);   // ring copies: lane - off
                const bool dep = in & ring & (ptr >= 0);                // the source byte belongs to this sub-step
                u32 byte = c_all[dep ? 0u : a];                         // (every lane reads: no exec-mask bookkeeping; idle lanes read some byte)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(byte), "+v"(Dn));   // both answers are here before the store below is issued: nothing waits for IT
                if (ballot64(dep)) {
                    // pointer doubling over the lanes: a lane whose source is outside the sub-step is a root (bit 8: its byte is known)
                    DPROF_ADD(4, 1);
                    u32 p4 = dep ? static_cast<u32>(ptr) << 2 : (lane << 2) | 256u;
                    do {
                        p4 = static_cast<u32>(__builtin_amdgcn_ds_bpermute(static_cast<int>(p4), static_cast<int>(p4)));
                        DPROF_ADD(3, 1);
                    } while (ballot64((p4 & 256u) == 0u));
                    byte = static_cast<u32>(__builtin_amdgcn_ds_bpermute(static_cast<int>(p4), static_cast<int>(byte)));
                }
                if (in) c_ring[pb & (kRing - 1u)] = static_cast<u8>(byte);
                lanes_sync_lds();
            }
            DPROF_TIME(14);                                             // sub-steps
            op += span;
            emitted += cur_ne;
            if (have_nxt) install(nxt);
            cur_ne = nxt.ne;
            cur_span = nxt.span;
            have_cur = have_nxt;
            DPROF_TIME(12);                                             // install: marks, recs, far pieces (the wait for them, if any is left)
        }
        write_out(true);
        DPROF_FLUSH;
        w.wv = 0x80000000u;
    }

    u32 fenced = 0;   // output bytes below this are known to have left the wave's store queue (FENCED only)

    // ---- tag loop  (SnappyDecompressor.cs:234-341) -----------------------------------------------------------
    while (st == SNP_OK && ip < n) {
        if (FRAG && op >= expected) break;                              // fragment full: the next tag belongs to the next one
        DPROF_ADD(9, 1);                                                // tags taken by the serial loop
        const u64 q = win_fetch(w, ip + mis, lane);
        const u32 c = static_cast<u32>(q) & 0xffu;
        const u32 type = c & 3u;
        const u32 hi6 = c >> 2;
        // bytes after the tag byte = CharTable[c] >> 11  (Constants.cs:42-76)
        const u32 extra = type == 0 ? (hi6 >= 60 ? hi6 - 59 : 0) : (type == 3 ? 4 : type);
        if (n - ip < 1 + extra) break;                                 // RefillTag: tag incomplete  :464-483
        const u32 tr_mask = extra >= 4 ? 0xffffffffu : ((1u << (8 * extra)) - 1u);
        const u32 trailer = static_cast<u32>(q >> 8) & tr_mask;         // ExtractLowBytes  Helpers.cs:72-85
        ip += 1 + extra;

        if (type == 0) {                                               // literal  :262-302
            const u64 len = hi6 >= 60 ? static_cast<u64>(trailer) + 1 : hi6 + 1;
            const u32 avail = n - ip;
            const u32 take = len < avail ? static_cast<u32>(len) : avail;   // partial literal then stop  :290-297
            if (take > expected - op) { st = SNP_ERR_TOO_LONG; break; }     // Append  :570-573
            if (FRAG && op < skip) {                                    // before the fragment: parse only
                if (take > skip - op) { st = kIrregular; break; }
                op += take;
                ip += take;
                if (take < len) break;
                continue;
            }
            if (take <= 64) {
                if (lane < take) dst[op + lane] = src[ip + lane];
            } else {
                wave_copy(dst + op, src + ip, take, lane);
            }
            op += take;
            ip += take;
            if (take < len) break;
        } else {                                                       // copy-1 / copy-2 / copy-4  :305-339
            u32 len, off;
            if (type == 1) { len = (hi6 & 7u) + 4; off = ((c >> 5) << 8) | trailer; }
            else { len = hi6 + 1; off = trailer; }
            if (FRAG && op < skip) {                                    // before the fragment: parse only
                if (len > skip - op) { st = kIrregular; break; }
                op += len;
                continue;
            }
            if (off == 0 || off > op - skip) { st = FRAG ? kIrregular : static_cast<i32>(SNP_ERR_BAD_OFFSET); break; }   // AppendFromSelf  :598-601
            if (len > expected - op) { st = SNP_ERR_TOO_LONG; break; }      // :603-606
            if (FENCED) {
                const u32 src_end = op - off + (off < len ? off : len);
                if (src_end > fenced) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    fenced = op;
                }
            }
            // IncrementalCopySlow semantics (CopyHelpers.cs:222-230): out[op+k] = out[op-off+k], serially in k,
            // i.e. out[op - off + (k mod off)].  k mod off for k < 64 by six compare-subtract steps (off < len only).
            u32 s = lane;
            if (off < len) {
#pragma unroll
                for (int sh = 5; sh >= 0; --sh) {
                    const u32 t = off << sh;
                    s = min(s, s - t);                                 // unsigned wrap: s - t is huge when s < t
                }
            }
            if (lane < len) dst[op + lane] = dst[op - off + s];
            op += len;
        }
    }
    if (st == SNP_OK && op < expected) st = SNP_ERR_INCOMPLETE;        // Snappy.cs:178-181,229-232

    if (lane == 0) {
        out_len[b] = st == SNP_OK ? op - skip : 0u;
        status[b] = st;
    }
}

#define SNP_D_PARAMS                                                                                                  \
    const u8 *__restrict__ in, const u64 *__restrict__ in_off, const u32 *__restrict__ in_len, u32 nblocks, u8 *out,  \
        const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap, u32 *__restrict__ out_len,                 \
        i32 *__restrict__ status, const u8 *__restrict__ chunk_type, const u32 *__restrict__ frag_skip, int redo_only
#define SNP_D_ARGS in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, frag_skip, redo_only

template <bool FENCED, int FRONT, bool FRAG>
__global__ __launch_bounds__(SNP_WAVE) SNP_D_OCC void k_decompress(SNP_D_PARAMS)
{
    decompress_block<FENCED, FRONT, FRAG>(SNP_D_ARGS, blockIdx.x);
}

// The sub-chain front end needs 68 VGPRs left to itself; at 64 (two spilled) it runs eight wavefronts per SIMD instead of
// seven: measured 677 -> 711 GB/s (html-like, 65 536 blocks; profiles/r02m_chains_variants.jsonl).
#ifndef SNP_D_CHAIN_WAVES
#define SNP_D_CHAIN_WAVES 8
#endif
#if SNP_D_PC
template <bool FENCED>   // (experiment: two wavefronts per block, parser + executor; see FRONT = 5)
__global__ __launch_bounds__(2 * SNP_WAVE) __attribute__((amdgpu_waves_per_eu(SNP_D_CHAIN_WAVES, SNP_D_CHAIN_WAVES))) void k_decompress_chains(SNP_D_PARAMS)
{
    decompress_block<FENCED, 5, false>(SNP_D_ARGS, blockIdx.x);
}
#else
template <bool FENCED>
__global__ __launch_bounds__(SNP_WAVE) __attribute__((amdgpu_waves_per_eu(SNP_D_CHAIN_WAVES, SNP_D_CHAIN_WAVES))) void k_decompress_chains(SNP_D_PARAMS)
{
    decompress_block<FENCED, 3, false>(SNP_D_ARGS, blockIdx.x);
}
#endif

// The same over a LIST of blocks: the blocks the small-block pre-pass (decompress_small.hip) did not finish, which it appended
// to `list` (ctl[0] = how many).  Persistent: the grid is one chip-full of wavefronts and each takes the next list entry with a
// ticket (ctl[1]) until the list is empty -- when the pre-pass finished everything (millions of small blocks) this launch costs
// microseconds instead of one empty workgroup per block (0.87 of 4.7 ms for 4 M blocks of 256 bytes), and when it finished
// nothing (64 KiB blocks) the wavefronts simply decode ~20 blocks each, balanced by the tickets.
template <bool FENCED, int FRONT>
__device__ __forceinline__ void decompress_list(
    SNP_D_PARAMS, const u32* __restrict__ list, u32* __restrict__ ctl, u32 sub_cap)
{
    // 64 sub-lists (decompress_small.hip, append_redo): lane s holds the number of entries before sub-list s.
    // (Starting the wavefronts a few microseconds apart, as a grid launch would, changes nothing: measured.)
    const u32 lane = lane_id();
    const u32 mine = __hip_atomic_load(&ctl[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 incl = wave_inclusive_scan(mine);
    const u32 count = read_lane(incl, 63);
    // tickets are taken several at a time: one counter for 2 M small blocks was the whole run time (12 ns per atomic on one
    // address: 512-byte blocks 40 GB/s); a wavefront now takes ~1/8 of its fair share per atomic, at most 64
    const u32 grab = min(max(count / (gridDim.x * 8u), 1u), 64u);
    // (a wavefront's first tickets are its by position; only the later ones come from the counter: 8 192 wavefronts hitting one
    // address at once took 82 us, the whole cost of an empty list)
    for (u32 first = blockIdx.x * grab; first < count;) {
        const u32 last = min(first + grab, count);
        for (u32 i = first; i < last; ++i) {
            const u32 sub = static_cast<u32>(__builtin_popcountll(ballot64(incl <= i)));   // the sub-list ticket i falls into
            const u32 before = read_lane(incl - mine, sub);
            decompress_block<FENCED, FRONT, false>(SNP_D_ARGS, list[static_cast<u64>(sub) * sub_cap + (i - before)]);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the next block reuses the LDS arrays)
        }
        u32 next = 0;
        if (lane == 0) next = atomicAdd(&ctl[64], grab);
        first = gridDim.x * grab + bcast_first(next);
    }
}

template <bool FENCED>
__global__ __launch_bounds__(SNP_WAVE) __attribute__((amdgpu_waves_per_eu(SNP_D_CHAIN_WAVES, SNP_D_CHAIN_WAVES))) void k_decompress_chains_list(
    SNP_D_PARAMS, const u32* __restrict__ list, u32* __restrict__ ctl, u32 sub_cap)
{
    decompress_list<FENCED, 3>(SNP_D_ARGS, list, ctl, sub_cap);
}

// FRONT = 4: output-granular execution through an LDS ring (8 KiB of LDS per wavefront: 20 wavefronts per CU).
template <bool FENCED>
__global__ __launch_bounds__(SNP_WAVE) void k_decompress_ring(SNP_D_PARAMS)
{
    decompress_block<FENCED, 4, false>(SNP_D_ARGS, blockIdx.x);
}
template <bool FENCED>
__global__ __launch_bounds__(SNP_WAVE) void k_decompress_ring_list(SNP_D_PARAMS, const u32* __restrict__ list, u32* __restrict__ ctl, u32 sub_cap)
{
    decompress_list<FENCED, 4>(SNP_D_ARGS, list, ctl, sub_cap);
}

}  // namespace

#if SNP_D_PROF
extern "C" int snp_debug_read_dprof(unsigned long long* out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dprof), sizeof(g_dprof));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_dprof), z, sizeof(z));
    }
    return static_cast<int>(e);
}
#endif


// decode_chains.hip: the round-5 default decoder (mode bit 3 without bit 6; bit 6 keeps the round-4 form of it selectable for A/B)
extern "C" hipError_t snp_launch_decode_chains(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out, const u64* out_off,
                                               const u32* out_cap, u32* out_len, i32* status, const u8* chunk_type, int fenced, int redo_only,
                                               unsigned lds_bytes, hipStream_t stream, const u32* frag_skip);
extern "C" hipError_t snp_launch_decode_chains_list(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out, const u64* out_off,
                                                    const u32* out_cap, u32* out_len, i32* status, const u8* chunk_type, int fenced,
                                                    unsigned lds_bytes, hipStream_t stream, const u32* list, u32* ctl, u32 waves, u32 sub_cap);

// The blocks of `list` (see k_decompress_chains_list): waves = wavefronts to launch (one chip-full), mode bit 0 = FENCED.
extern "C" hipError_t snp_launch_decompress_list(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                                 const u64* out_off, const u32* out_cap, u32* out_len, i32* status,
                                                 const u8* chunk_type, int mode, hipStream_t stream, const u32* list, u32* ctl,
                                                 u32 waves, u32 sub_cap)
{
    if (nblocks == 0) return hipSuccess;
    const unsigned lds_bytes = static_cast<unsigned>(mode >> 8) * 256u;
    const u32* const no_skip = nullptr;
    if (!(mode & 32) && !(mode & 64))
        return snp_launch_decode_chains_list(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, mode & 1, lds_bytes, stream,
                                             list, ctl, waves, sub_cap);
    if (mode & 32) {                                    // output-granular execution through an LDS ring
        if (mode & 1)
            hipLaunchKernelGGL((k_decompress_ring_list<true>), dim3(waves), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks,
                               out, out_off, out_cap, out_len, status, chunk_type, no_skip, 0, list, ctl, sub_cap);
        else
            hipLaunchKernelGGL((k_decompress_ring_list<false>), dim3(waves), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks,
                               out, out_off, out_cap, out_len, status, chunk_type, no_skip, 0, list, ctl, sub_cap);
        return hipGetLastError();
    }
    if (mode & 1)
        hipLaunchKernelGGL((k_decompress_chains_list<true>), dim3(waves), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks,
                           out, out_off, out_cap, out_len, status, chunk_type, no_skip, 0, list, ctl, sub_cap);
    else
        hipLaunchKernelGGL((k_decompress_chains_list<false>), dim3(waves), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks,
                           out, out_off, out_cap, out_len, status, chunk_type, no_skip, 0, list, ctl, sub_cap);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_decompress(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                            const u64* out_off, const u32* out_cap, u32* out_len, i32* status,
                                            const u8* chunk_type, int mode, hipStream_t stream, const u32* frag_skip)
{
    // mode bit 0: FENCED, bit 1: serial-only (no token-parallel front end), bit 2: batches without the execution queue, bit 3: sub-chain parse;
    // bit 4: only the blocks decompress_small.hip left marked -1; bit 5: sub-chain parse + LDS ring (FRONT = 4);  bits 8..: dynamic LDS bytes / 256 requested per wavefront purely to cap how many blocks a CU decodes at once
    if (nblocks == 0) return hipSuccess;
    const unsigned lds_bytes = static_cast<unsigned>(mode >> 8) * 256u;
#define SNP_LAUNCH_DEC(F, B)                                                                                        \
    hipLaunchKernelGGL((k_decompress<F, B, false>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off,    \
                       in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, nullptr, (mode >> 4) & 1)
#define SNP_LAUNCH_FRAG(F, B)                                                                                       \
    hipLaunchKernelGGL((k_decompress<F, B, true>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off,     \
                       in_len, nblocks, out, out_off, out_cap, out_len, status, nullptr, frag_skip, 0)
    if (frag_skip) {                                    // fragments of one large block (tag_index.hip)
        switch (mode & 7) {
            case 0: case 4: SNP_LAUNCH_FRAG(false, 2); break;
            case 1: case 5: SNP_LAUNCH_FRAG(true, 2); break;
            case 2: case 6: SNP_LAUNCH_FRAG(false, 0); break;
            default: SNP_LAUNCH_FRAG(true, 0); break;
        }
        return hipGetLastError();
    }
    if (mode & 32) {                                    // sub-chain parse + output-granular execution through an LDS ring (FRONT = 4)
        if (mode & 1)
            hipLaunchKernelGGL((k_decompress_ring<true>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len,
                               nblocks, out, out_off, out_cap, out_len, status, chunk_type, nullptr, (mode >> 4) & 1);
        else
            hipLaunchKernelGGL((k_decompress_ring<false>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len,
                               nblocks, out, out_off, out_cap, out_len, status, chunk_type, nullptr, (mode >> 4) & 1);
        return hipGetLastError();
    }
    if ((mode & 8) && !(mode & 64))
        return snp_launch_decode_chains(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, mode & 1, (mode >> 4) & 1, lds_bytes, stream, nullptr);
    if (mode & 8) {                                     // sub-chain parse
        if (mode & 1)
            hipLaunchKernelGGL((k_decompress_chains<true>), dim3(nblocks), dim3(SNP_WAVE * (SNP_D_PC ? 2 : 1)), lds_bytes, stream, in, in_off, in_len,
                               nblocks, out, out_off, out_cap, out_len, status, chunk_type, nullptr, (mode >> 4) & 1);
        else
            hipLaunchKernelGGL((k_decompress_chains<false>), dim3(nblocks), dim3(SNP_WAVE * (SNP_D_PC ? 2 : 1)), lds_bytes, stream, in, in_off, in_len,
                               nblocks, out, out_off, out_cap, out_len, status, chunk_type, nullptr, (mode >> 4) & 1);
        return hipGetLastError();
    }
    switch (mode & 7) {
        case 0: SNP_LAUNCH_DEC(false, 2); break;
        case 1: SNP_LAUNCH_DEC(true, 2); break;
        case 2: case 6: SNP_LAUNCH_DEC(false, 0); break;
        case 3: case 7: SNP_LAUNCH_DEC(true, 0); break;
        case 4: SNP_LAUNCH_DEC(false, 1); break;
        default: SNP_LAUNCH_DEC(true, 1); break;
    }
#undef SNP_LAUNCH_DEC
#undef SNP_LAUNCH_FRAG
    return hipGetLastError();
}
