// decompress.hip -- Snappy block decompression, one 64 KiB block per wavefront (gfx950).
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

template <bool FENCED>
__global__ __launch_bounds__(SNP_WAVE) void k_decompress(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                        const u32* __restrict__ in_len, u32 nblocks, u8* out,
                                                        const u64* __restrict__ out_off,
                                                        const u32* __restrict__ out_cap, u32* __restrict__ out_len,
                                                        i32* __restrict__ status, const u8* __restrict__ chunk_type)
{
    const u32 b = blockIdx.x;
    if (b >= nblocks) return;
    const u32 lane = lane_id();
    const u8* src = in + in_off[b];
    const u32 n = bcast_first(in_len[b]);
    u8* dst = out + out_off[b];
    const u32 cap = bcast_first(out_cap[b]);

    if (chunk_type && chunk_type[b] == 1) {     // framing: uncompressed chunk body  SnappyStreamDecompressor.cs:137-163
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
    {
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

    u32 fenced = 0;   // output bytes below this are known to have left the wave's store queue (FENCED only)

    // ---- tag loop  (SnappyDecompressor.cs:234-341) -----------------------------------------------------------
    while (st == SNP_OK && ip < n) {
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
            if (off == 0 || off > op) { st = SNP_ERR_BAD_OFFSET; break; }   // AppendFromSelf  :598-601
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
        out_len[b] = st == SNP_OK ? op : 0u;
        status[b] = st;
    }
}

}  // namespace

extern "C" hipError_t snp_launch_decompress(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                            const u64* out_off, const u32* out_cap, u32* out_len, i32* status,
                                            const u8* chunk_type, int fenced, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    if (fenced)
        hipLaunchKernelGGL(k_decompress<true>, dim3(nblocks), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks,
                           out, out_off, out_cap, out_len, status, chunk_type);
    else
        hipLaunchKernelGGL(k_decompress<false>, dim3(nblocks), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks,
                           out, out_off, out_cap, out_len, status, chunk_type);
    return hipGetLastError();
}
