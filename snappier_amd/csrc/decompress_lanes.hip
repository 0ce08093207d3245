// decompress_lanes.hip -- Snappy block decompression, one block per LANE (gfx950): the large-batch layout.
//
// Same semantics as decompress.hip (SnappyDecompressor.DecompressAllTags + Append / AppendFromSelf,
// Snappier/Internal/SnappyDecompressor.cs:184-347,568-611; copy semantics CopyHelpers.cs:222-230), but every lane
// walks the tag stream of its own block: 64 blocks per wavefront.  The tag loop is a serial dependency chain inside
// a block (each tag's position depends on the previous tag's length), so for a large batch the parallelism that
// pays is ACROSS blocks -- ~160 k independent streams keep the memory system busy where one block per wavefront
// leaves the scalar unit as the bottleneck.  Each lane reads its input with unaligned 8-byte loads, copies literals
// and back-references with 16/8/4/1-byte unaligned accesses, and resolves pattern copies (offset < length) by
// doubling the already written prefix.  Small batches keep using decompress.hip.
#include "snp_device.h"

namespace {

// Forward copy of len bytes inside the output where the source may overlap the destination (offset >= 1):
// out[op + k] = out[op - off + k] in increasing k (IncrementalCopySlow, CopyHelpers.cs:222-230).
__device__ __forceinline__ void lane_self_copy(u8* dst, u32 op, u32 off, u32 len)
{
    u8* d = dst + op;
    const u8* s = d - off;
    if (off >= 16) {                                                   // 16-byte steps never read what they wrote
        u32 i = 0;
        for (; i + 16 <= len; i += 16)
            *reinterpret_cast<snp_u128_unaligned*>(d + i) = *reinterpret_cast<const snp_u128_unaligned*>(s + i);
        for (; i + 4 <= len; i += 4) st32u(d + i, ld32u(s + i));
        for (; i < len; ++i) d[i] = s[i];
    } else if (off >= 4) {
        u32 i = 0;
        for (; i + 4 <= len; i += 4) st32u(d + i, ld32u(s + i));
        for (; i < len; ++i) d[i] = s[i];
    } else {
        for (u32 i = 0; i < len; ++i) d[i] = s[i];
    }
}

__global__ __launch_bounds__(SNP_WAVE) void k_decompress_lanes(const u8* __restrict__ in, const u64* __restrict__ in_off,
                                                              const u32* __restrict__ in_len, u32 nblocks, u8* out,
                                                              const u64* __restrict__ out_off,
                                                              const u32* __restrict__ out_cap, u32* __restrict__ out_len,
                                                              i32* __restrict__ status, const u8* __restrict__ chunk_type)
{
    const u32 b = blockIdx.x * SNP_WAVE + threadIdx.x;
    if (b >= nblocks) return;
    const u8* src = in + in_off[b];
    const u32 n = in_len[b];
    u8* dst = out + out_off[b];
    const u32 cap = out_cap[b];

    if (chunk_type && chunk_type[b] == 1) {     // framing: uncompressed chunk body  SnappyStreamDecompressor.cs:137-163
        const bool fits = n <= cap;
        if (fits) {
            u32 i = 0;
            for (; i + 16 <= n; i += 16)
                *reinterpret_cast<snp_u128_unaligned*>(dst + i) = *reinterpret_cast<const snp_u128_unaligned*>(src + i);
            for (; i < n; ++i) dst[i] = src[i];
        }
        out_len[b] = fits ? n : 0u;
        status[b] = fits ? SNP_OK : SNP_ERR_OUTPUT_TOO_SMALL;
        return;
    }

    i32 st = SNP_OK;
    u32 ip = 0, op = 0, expected = 0;
    {   // varint preamble  (VarIntEncoding.TryReadSlow  VarIntEncoding.Read.cs:38-79)
        u32 shift = 0;
        bool done = false;
        while (ip < n && ip < 5) {
            const u32 c = src[ip++];
            const u32 val = c & 0x7fu;
            if (val & ~(0xffffffffu >> shift)) { st = SNP_ERR_BAD_LENGTH; break; }
            expected |= val << shift;
            shift += 7;
            if (c < 128) { done = true; break; }
        }
        if (st == SNP_OK && !done) st = ip >= 5 ? SNP_ERR_BAD_LENGTH : SNP_ERR_INCOMPLETE;
        if (st == SNP_OK && expected > 0x7fffffffu) st = SNP_ERR_BAD_LENGTH;
        if (st == SNP_OK && cap < expected) st = SNP_ERR_OUTPUT_TOO_SMALL;
    }

    while (st == SNP_OK && ip < n) {                                   // SnappyDecompressor.cs:234-341
        // tag byte + up to 4 trailing bytes; a full 8-byte load when it stays inside the input
        u64 q;
        if (ip + 8 <= n) q = ld64u(src + ip);
        else {
            q = 0;
            for (u32 k = 0; ip + k < n; ++k) q |= static_cast<u64>(src[ip + k]) << (8 * k);
        }
        const u32 c = static_cast<u32>(q) & 0xffu;
        const u32 type = c & 3u;
        const u32 hi6 = c >> 2;
        const u32 extra = type == 0 ? (hi6 >= 60 ? hi6 - 59 : 0) : (type == 3 ? 4 : type);   // CharTable[c] >> 11
        if (n - ip < 1 + extra) break;                                 // RefillTag  :464-483
        const u32 tr_mask = extra >= 4 ? 0xffffffffu : ((1u << (8 * extra)) - 1u);
        const u32 trailer = static_cast<u32>(q >> 8) & tr_mask;
        ip += 1 + extra;
        if (type == 0) {                                               // literal  :262-302
            const u64 len = hi6 >= 60 ? static_cast<u64>(trailer) + 1 : hi6 + 1;
            const u32 avail = n - ip;
            const u32 take = len < avail ? static_cast<u32>(len) : avail;
            if (take > expected - op) { st = SNP_ERR_TOO_LONG; break; }     // :570-573
            const u8* s = src + ip;
            u8* d = dst + op;
            u32 i = 0;
            for (; i + 16 <= take; i += 16)
                *reinterpret_cast<snp_u128_unaligned*>(d + i) = *reinterpret_cast<const snp_u128_unaligned*>(s + i);
            for (; i + 4 <= take; i += 4) st32u(d + i, ld32u(s + i));
            for (; i < take; ++i) d[i] = s[i];
            op += take;
            ip += take;
            if (take < len) break;                                     // :290-297
        } else {
            u32 len, off;
            if (type == 1) { len = (hi6 & 7u) + 4; off = ((c >> 5) << 8) | trailer; }
            else { len = hi6 + 1; off = trailer; }
            if (off == 0 || off > op) { st = SNP_ERR_BAD_OFFSET; break; }   // :598-601
            if (len > expected - op) { st = SNP_ERR_TOO_LONG; break; }      // :603-606
            lane_self_copy(dst, op, off, len);
            op += len;
        }
    }
    if (st == SNP_OK && op < expected) st = SNP_ERR_INCOMPLETE;        // Snappy.cs:178-181,229-232
    out_len[b] = st == SNP_OK ? op : 0u;
    status[b] = st;
}

}  // namespace

extern "C" hipError_t snp_launch_decompress_lanes(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks,
                                                  u8* out, const u64* out_off, const u32* out_cap, u32* out_len,
                                                  i32* status, const u8* chunk_type, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    const u32 grid = (nblocks + SNP_WAVE - 1) / SNP_WAVE;
    hipLaunchKernelGGL(k_decompress_lanes, dim3(grid), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, out, out_off,
                       out_cap, out_len, status, chunk_type);
    return hipGetLastError();
}
