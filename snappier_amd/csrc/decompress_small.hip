// decompress_small.hip -- SMALL Snappy blocks, one block per LANE (gfx950).  SURVEY.md 8(f4): the "many tiny chunks" pattern
// of Snappier.Tests/SnappyStreamTests.cs:145-192 and batches of sub-page records.
//
// decompress.hip spends a whole wavefront on a block: a 64-position window, a 64-tag execution queue.  A 256-byte block holds
// ~8 tags -- one window, one nearly empty batch -- and 4 M of them are 4 M workgroups (126 GB/s measured).  Here every lane
// decodes its own block (SnappyDecompressor.DecompressAllTags + Append / AppendFromSelf, SnappyDecompressor.cs:184-347,
// 568-611; copy semantics CopyHelpers.cs:222-230), 64 blocks per wavefront, and the loop is shaped so that one tag costs
// ONE dependent memory round trip:
//   * the next tag's bytes are requested as soon as this tag's length is known, before its copy is issued;
//   * copies are 16-byte pieces (stores exact at the block's end, may overshoot inside it: later tags overwrite the excess);
//   * an overlapping copy (offset < length) doubles the written prefix: log2(length / offset) steps instead of a byte loop.
// This kernel only ever FINISHES clean blocks of at most `small_max` declared bytes.  Anything else -- a larger block, an
// uncompressed framing chunk, and every irregularity (bad preamble, offset, length, truncated input) -- is left untouched
// and marked kRedoStatus; decompress.hip then decodes exactly those blocks and owns every error code.
#include "snp_device.h"

namespace {

constexpr i32 kRedoStatus = -1;

struct __attribute__((packed)) snp_u16_unaligned_s { u16 v; };

// len bytes from s to d, no overlap within a 16-byte piece (callers guarantee s + 16 <= d or s >= d + len per piece).
// Loads may read up to 15 bytes past s + len when `s_slop`; stores may write up to 15 bytes past d + len when `d_slop`.
__device__ __forceinline__ void copy_pieces(u8* d, const u8* s, u32 len, bool s_slop, bool d_slop)
{
    u32 i = 0;
    for (; i + 16 <= len; i += 16)
        *reinterpret_cast<snp_u128_unaligned*>(d + i) = *reinterpret_cast<const snp_u128_unaligned*>(s + i);
    const u32 r = len - i;
    if (r == 0) return;
    if (s_slop && d_slop) {                                             // one more whole piece
        *reinterpret_cast<snp_u128_unaligned*>(d + i) = *reinterpret_cast<const snp_u128_unaligned*>(s + i);
        return;
    }
    if (len >= 16) {                                                    // the last 16 bytes again, ending exactly at len
        *reinterpret_cast<snp_u128_unaligned*>(d + len - 16) = *reinterpret_cast<const snp_u128_unaligned*>(s + len - 16);
        return;
    }
    if (r & 8u) { reinterpret_cast<snp_u64_unaligned*>(d + i)->v = ld64u(s + i); i += 8; }
    if (r & 4u) { st32u(d + i, ld32u(s + i)); i += 4; }
    if (r & 2u) { reinterpret_cast<snp_u16_unaligned_s*>(d + i)->v = reinterpret_cast<const snp_u16_unaligned_s*>(s + i)->v; i += 2; }
    if (r & 1u) d[i] = s[i];
}

__global__ __launch_bounds__(SNP_WAVE) void k_decompress_small(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                              const u32* __restrict__ in_len, u32 nblocks, u8* out,
                                                              const u64* __restrict__ out_off,
                                                              const u32* __restrict__ out_cap, u32* __restrict__ out_len,
                                                              i32* __restrict__ status, const u8* __restrict__ chunk_type,
                                                              u32 small_max)
{
    const u32 b = blockIdx.x * SNP_WAVE + threadIdx.x;
    if (b >= nblocks) return;
    const u32 n = in_len[b];
    const u32 cap = out_cap[b];
    bool redo = cap > small_max || n > 2 * small_max + 64 || n < 1 || (chunk_type && chunk_type[b] == 1);
    const u8* src = in + in_off[b];
    u8* dst = out + out_off[b];
    u32 ip = 0, op = 0, expected = 0;
    if (!redo) {                                                        // varint preamble  VarIntEncoding.Read.cs:38-79
        u32 shift = 0;
        bool done = false;
        while (ip < n && ip < 5) {
            const u32 c = src[ip++];
            const u32 val = c & 0x7fu;
            if (val & ~(0xffffffffu >> shift)) break;
            expected |= val << shift;
            shift += 7;
            if (c < 128) { done = true; break; }
        }
        redo = !done || expected > cap;
    }
    // the first tag
    u64 q = 0;
    bool have = !redo && ip + 8 <= n;
    if (have) q = ld64u(src + ip);
    while (!redo && op < expected) {
        if (!have) {                                                    // the last 7 bytes of the input: bytewise
            if (ip >= n) { redo = true; break; }                        // input ends early: "Incomplete" is decompress.hip's to report
            q = 0;
            for (u32 k = 0; ip + k < n && k < 8; ++k) q |= static_cast<u64>(src[ip + k]) << (8 * k);
        }
        const u32 c = static_cast<u32>(q) & 0xffu;
        const u32 type = c & 3u;
        const u32 hi6 = c >> 2;
        const u32 extra = type == 0 ? (hi6 >= 60 ? hi6 - 59 : 0u) : (type == 3 ? 4u : type);   // CharTable[c] >> 11  Constants.cs:42-76
        if (n - ip < 1 + extra) { redo = true; break; }
        const u32 trailer = extra >= 4 ? static_cast<u32>(q >> 8) : (static_cast<u32>(q >> 8) & ((1u << (8 * extra)) - 1u));
        const u32 body = ip + 1 + extra;
        u32 len, off = 0;
        if (type == 0) len = (hi6 >= 60 ? trailer : hi6) + 1;
        else if (type == 1) { len = (hi6 & 7u) + 4; off = ((c >> 5) << 8) | trailer; }
        else { len = hi6 + 1; off = trailer; }
        const u32 nip = type == 0 ? body + len : body;
        // irregular -> leave it all to decompress.hip (TOO_LONG / BAD_OFFSET / partial literal)
        if (len > expected - op || (type == 0 ? (len > n - body) : (off == 0 || off > op))) { redo = true; break; }
        // request the next tag now: it does not depend on this tag's copy
        have = nip + 8 <= n;
        u64 qn = 0;
        if (have) qn = ld64u(src + nip);
        u8* d = dst + op;
        if (type == 0) {                                                // literal  :262-302, Append :568-589
            copy_pieces(d, src + body, len, body + len + 16 <= n, op + len + 16 <= cap);
        } else if (off >= len || off >= 16) {                           // pieces never read what they write
            copy_pieces(d, d - off, len, true, op + len + 16 <= cap);   // the source's slop is earlier output of this block
        } else {
            // offset < length and < 16: the written prefix doubles (IncrementalCopy semantics: out[op+k] = out[op-off+k])
            u32 have_b = off;                                           // bytes of the pattern run available behind op + done
            u32 done = 0;
            while (done < len) {
                const u32 m = have_b < len - done ? have_b : len - done;
                copy_pieces(d + done, d - off, m, false, false);
                done += m;
                have_b += m;
            }
        }
        op += len;
        ip = nip;
        q = qn;
    }
    // a clean block ends exactly at `expected`; trailing input beyond the last needed tag is ignored, as in the reference
    if (!redo && (op != expected || ip != n)) redo = true;               // bytes after the last needed tag: decompress.hip decides (TOO_LONG or ignored)
    if (redo) { status[b] = kRedoStatus; return; }
    out_len[b] = op;
    status[b] = SNP_OK;
}

}  // namespace

extern "C" hipError_t snp_launch_decompress_small(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                                  const u64* out_off, const u32* out_cap, u32* out_len, i32* status,
                                                  const u8* chunk_type, u32 small_max, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    const u32 grid = (nblocks + SNP_WAVE - 1) / SNP_WAVE;
    hipLaunchKernelGGL(k_decompress_small, dim3(grid), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, out, out_off,
                       out_cap, out_len, status, chunk_type, small_max);
    return hipGetLastError();
}
