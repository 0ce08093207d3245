// decode_common.h -- what every block-decoder kernel shares: the block's state, the varint preamble, and the serial tag loop that
// owns the reference's exact error semantics (Snappier/Internal/SnappyDecompressor.cs:184-347,568-611; CopyHelpers.cs:222-230).
// One 64-lane wavefront decodes one Snappy block; the front ends (decompress.hip) take the regular bulk of a block and hand whatever is
// irregular -- an error, a literal > 64 bytes that runs past the input, the block's last bytes -- to serial_tail().
#pragma once
#include "snp_device.h"

namespace {

// ---- input window of the serial loop: 512 bytes in two VGPRs per lane, tag bytes pulled out with v_readlane -----------------------
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
__device__ __forceinline__ void st64u(u8* p, u32 lo, u32 hi) { reinterpret_cast<snp_u64_unaligned*>(p)->v = lo | (static_cast<u64>(hi) << 32); }
__device__ __forceinline__ void st16u(u8* p, u32 v) { reinterpret_cast<snp_u16_unaligned*>(p)->v = static_cast<u16>(v); }

// DS operations of one wavefront execute in order; this only stops the compiler from reordering or forwarding them.
__device__ __forceinline__ void lanes_sync_lds() { asm volatile("" ::: "memory"); }

// k mod off for k = lane < 64 by six compare-subtract steps (pattern copies, off < len: CopyHelpers.cs:222-230 copies byte by byte).
__device__ __forceinline__ u32 lane_mod(u32 lane, u32 off)
{
    u32 s = lane;
#pragma unroll
    for (int sh = 5; sh >= 0; --sh) {
        const u32 t = off << sh;
        s = min(s, s - t);                                              // unsigned wrap: s - t is huge when s < t
    }
    return s;
}

// FRAG = true decodes one 64 KiB output FRAGMENT of a larger block (see tag_index.hip): the wave starts at a tag boundary at or
// before the fragment (`frag_skip[b]` output bytes early), parses the tags in between without producing them, and stops when the
// fragment is full.  A tag that straddles the fragment start or a copy that reaches back before it ends with kIrregular; the caller
// then decodes the whole block with one wavefront instead, which also owns the exact error semantics.
constexpr i32 kIrregular = 99;

#define SNP_D_PARAMS                                                                                                  \
    const u8 *__restrict__ in, const u64 *__restrict__ in_off, const u32 *__restrict__ in_len, u32 nblocks, u8 *out,  \
        const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap, u32 *__restrict__ out_len,                 \
        i32 *__restrict__ status, const u8 *__restrict__ chunk_type, const u32 *__restrict__ frag_skip, int redo_only
#define SNP_D_ARGS in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, frag_skip, redo_only

// One block being decoded.  ip / op count from the block's first compressed byte / from `skip` bytes before its output.
struct DecBlk {
    const u8* src;
    u8* dst;
    u32 n, cap, expected, skip;
    u32 ip, op;
    i32 st;
    u32 mis;
    InWindow w;
};

// Everything before the first tag: early outs (a block another kernel finished, an uncompressed framing chunk, an over-long input)
// and the varint preamble (VarIntEncoding.TryReadSlow  VarIntEncoding.Read.cs:38-79).  false = the block is done.
template <bool FRAG>
__device__ __forceinline__ bool block_begin(SNP_D_PARAMS, const u32 b, const u32 lane, DecBlk& B)
{
    if (b >= nblocks) return false;
    if (redo_only && status[b] != -1) return false;      // decompress_small.hip finished this block (it marks the others -1)
    B.src = in + in_off[b];
    B.n = bcast_first(in_len[b]);
    B.skip = FRAG ? bcast_first(frag_skip[b]) : 0u;      // output bytes parsed but not produced
    B.dst = out + out_off[b] - B.skip;                   // output offsets count from `skip` bytes early
    B.cap = bcast_first(out_cap[b]);
    if (B.n > 0x7fffffffu - 1024u) {                     // the reference's spans are int-length; keeps ip + k arithmetic below 2^32
        if (lane == 0) { out_len[b] = 0; status[b] = SNP_ERR_BAD_ARG; }
        return false;
    }
    if (!FRAG && chunk_type && chunk_type[b] == 1) {     // framing: uncompressed chunk body  SnappyStreamDecompressor.cs:137-163
        const bool fits = B.n <= B.cap;
        if (fits) wave_copy(B.dst, B.src, B.n, lane);
        if (lane == 0) {
            out_len[b] = fits ? B.n : 0u;
            status[b] = fits ? SNP_OK : SNP_ERR_OUTPUT_TOO_SMALL;
        }
        return false;
    }
    B.mis = static_cast<u32>(reinterpret_cast<uintptr_t>(B.src) & 3u);
    B.w.a0 = B.src - B.mis;
    B.w.end = B.src + B.n;
    B.w.wv = 0;
    B.w.lo = win_load(B.w, 4 * lane);
    B.w.hi = win_load(B.w, 256 + 4 * lane);
    B.st = SNP_OK;
    B.ip = 0;
    B.op = 0;
    B.expected = 0;
    if (FRAG) {
        B.expected = B.skip + B.cap;                     // no preamble: the fragment ends `cap` bytes after its start
    } else {
        const u64 q = win_fetch(B.w, B.mis, lane);
        u32 shift = 0, result = 0;
        bool done = false;
        for (u32 i = 0; i < 5 && !done; ++i) {
            if (i >= B.n) { B.st = SNP_ERR_INCOMPLETE; break; }        // NeedMoreData -> never AllDataDecompressed
            const u32 c = static_cast<u32>(q >> (8 * i)) & 0xffu;
            const u32 val = c & 0x7fu;
            if (val & ~(0xffffffffu >> shift)) { B.st = SNP_ERR_BAD_LENGTH; break; }   // LeftShiftOverflows  Helpers.cs:65-70
            result |= val << shift;
            shift += 7;
            B.ip = i + 1;
            if (c < 128) done = true;
        }
        if (B.st == SNP_OK && !done) B.st = SNP_ERR_BAD_LENGTH;        // five continuation bytes: shift >= 32  :65-69
        B.expected = result;
        if (B.st == SNP_OK && B.expected > 0x7fffffffu) B.st = SNP_ERR_BAD_LENGTH;   // (int)length < 0 in the reference
        if (B.st == SNP_OK && B.cap < B.expected) B.st = SNP_ERR_OUTPUT_TOO_SMALL;   // Snappy.cs:183-185
    }
    return true;
}

// The tag loop (SnappyDecompressor.cs:234-341), one tag per trip with the whole wave, from (ip, op) to the end of the block, and the
// block's result.  A front end must have left every output byte below `op` either stored or (FENCED = false) issued by this wave.
template <bool FENCED, bool FRAG>
__device__ __forceinline__ void serial_tail(DecBlk& B, const u32 b, const u32 lane, u32* __restrict__ out_len, i32* __restrict__ status)
{
    const u8* const src = B.src;
    u8* const dst = B.dst;
    const u32 n = B.n, expected = B.expected, skip = B.skip;
    u32 ip = B.ip, op = B.op;
    i32 st = B.st;
    u32 fenced = 0;   // output bytes below this are known to have left the wave's store queue (FENCED only)
    while (st == SNP_OK && ip < n) {
        if (FRAG && op >= expected) break;                              // fragment full: the next tag belongs to the next one
        const u64 q = win_fetch(B.w, ip + B.mis, lane);
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
            // i.e. out[op - off + (k mod off)].
            const u32 s = off < len ? lane_mod(lane, off) : lane;
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

}  // namespace
