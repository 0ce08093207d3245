// framing.hip -- device side of the Snappy framing format (SnappyStreamCompressor.cs:18-21,194-261) and the
// segment gather used to concatenate independently compressed fragments (SnappyCompressor.cs:40-80 writes them
// back to back).  Pure data movement: 256-thread workgroups, 16 B per lane.
#include "snp_device.h"

namespace {

// Copy len bytes with a whole 256-thread workgroup (no overlap).
__device__ __forceinline__ void block_copy(u8* dst, const u8* src, u32 len, u32 tid)
{
    u32 k = tid * 16;
    for (; k + 16 <= len; k += 256 * 16) {
        snp_u128_unaligned w = *reinterpret_cast<const snp_u128_unaligned*>(src + k);
        *reinterpret_cast<snp_u128_unaligned*>(dst + k) = w;
    }
    const u32 tail = len & ~15u;
    if (tid < (len & 15u)) dst[tail + tid] = src[tail + tid];
}

__global__ __launch_bounds__(256) void k_gather(const u8* __restrict__ src, const u64* __restrict__ src_off,
                                               const u32* __restrict__ seg_len, u8* __restrict__ dst,
                                               const u64* __restrict__ dst_off, u32 nseg)
{
    const u32 s = blockIdx.x;
    if (s >= nseg) return;
    block_copy(dst + dst_off[s], src + src_off[s], seg_len[s], threadIdx.x);
}

// Chunk table of a raw stream cut into 65536-byte chunks (SnappyStreamCompressor.CompressInput  :166-192)
__global__ void k_frame_chunks(u64 n, u32 nchunks, u64 comp_stride, u64* in_off, u32* in_len, u64* comp_off)
{
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const u64 o = static_cast<u64>(c) * SNP_BLOCK_SIZE;
    in_off[c] = o;
    in_len[c] = static_cast<u32>(n - o < SNP_BLOCK_SIZE ? n - o : SNP_BLOCK_SIZE);
    comp_off[c] = static_cast<u64>(c) * comp_stride;
}

// Per chunk: compressed-vs-raw decision (CompressBlock  :212-229) and the exclusive scan of encoded sizes.
// Single 1024-thread workgroup; dst_off has nchunks + 1 entries, the last one is the total encoded length.
__global__ __launch_bounds__(1024) void k_frame_plan(const u32* __restrict__ in_len, const u32* __restrict__ comp_len,
                                                    u32 nchunks, u8* __restrict__ type, u32* __restrict__ payload,
                                                    u64* __restrict__ dst_off, u64* __restrict__ total)
{
    __shared__ u64 wave_sum[16];
    __shared__ u64 carry;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) carry = SNP_STREAM_HEADER_LEN;
    __syncthreads();
    for (u32 base = 0; base < nchunks; base += 1024) {
        const u32 i = base + tid;
        u64 v = 0;
        if (i < nchunks) {
            const bool shrink = comp_len[i] < in_len[i];                // :212
            const u32 pl = shrink ? comp_len[i] : in_len[i];
            type[i] = shrink ? 0 : 1;
            payload[i] = pl;
            v = SNP_CHUNK_HEADER_LEN + pl;
        }
        u64 x = v;
        for (u32 d = 1; d < 64; d <<= 1) {
            const u64 y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_sum[wave] = x;
        __syncthreads();
        if (wave == 0) {
            u64 s = lane < 16 ? wave_sum[lane] : 0;
            for (u32 d = 1; d < 16; d <<= 1) {
                const u64 y = __shfl_up(s, d, 64);
                if (lane >= d) s += y;
            }
            if (lane < 16) wave_sum[lane] = s;
        }
        __syncthreads();
        const u64 excl = carry + (wave ? wave_sum[wave - 1] : 0) + x - v;
        if (i < nchunks) dst_off[i] = excl;
        __syncthreads();
        if (tid == 1023) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0) {
        dst_off[nchunks] = carry;
        *total = carry;
    }
}

__constant__ u8 k_stream_header[SNP_STREAM_HEADER_LEN] = {0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59};   // :18-21

__global__ void k_frame_header_only(u8* dst, u64* total)
{
    if (threadIdx.x < SNP_STREAM_HEADER_LEN) dst[threadIdx.x] = k_stream_header[threadIdx.x];
    if (threadIdx.x == 0) *total = SNP_STREAM_HEADER_LEN;
}

// One workgroup per chunk: [type:1][len:3 LE = payload + 4][masked crc:4 LE][payload]   (:233-261)
__global__ __launch_bounds__(256) void k_frame_emit(const u8* __restrict__ raw, const u64* __restrict__ in_off,
                                                   const u8* __restrict__ comp, const u64* __restrict__ comp_off,
                                                   const u8* __restrict__ type, const u32* __restrict__ payload,
                                                   const u32* __restrict__ crc, const u64* __restrict__ dst_off,
                                                   u8* __restrict__ dst, u64 cap, u32 nchunks)
{
    const u32 c = blockIdx.x, tid = threadIdx.x;
    if (c >= nchunks) return;
    if (dst_off[nchunks] > cap) return;                                 // caller reports OUTPUT_TOO_SMALL from the total
    if (c == 0 && tid < SNP_STREAM_HEADER_LEN) dst[tid] = k_stream_header[tid];   // EnsureStreamHeaderWritten  :148-157
    const u64 o = dst_off[c];
    const u32 pl = payload[c];
    const u32 t = type[c];
    if (tid < 8) {
        const u32 bs = pl + 4;                                          // :236,251
        const u32 v = tid == 0 ? t : tid < 4 ? (bs >> (8 * (tid - 1))) : (crc[c] >> (8 * (tid - 4)));
        dst[o + tid] = static_cast<u8>(v);
    }
    const u8* src = t == 0 ? comp + comp_off[c] : raw + in_off[c];
    block_copy(dst + o + SNP_CHUNK_HEADER_LEN, src, pl, tid);
}


// ---- device-side chunk-header walk (SnappyStreamDecompressor.ReadChunkHeader / Decompress  :53-199,215-289) -----
// A framed stream carries no index: every header gives the position of the next one, so the walk is a serial chain of
// ~64 KiB hops (one 16-byte load per chunk: type, 24-bit size, masked CRC and the first bytes of the block preamble).
// One lane walks; the table it writes is exactly what the host walk in capi_frame.hip produces.  Entries past the last data
// chunk are filled as empty uncompressed chunks so that the decode and CRC launches can run over max_chunks without
// knowing the count on the host.
constexpr u32 kEmptyMaskedCrc = 0xa282ead8u;      // crc32c_mask(crc32c of no bytes = 0)

__global__ __launch_bounds__(SNP_WAVE) void k_frame_scan(const u8* __restrict__ in, u64 n, u64 cap, u32 max_chunks,
                                                        u8* __restrict__ type, u64* __restrict__ body_off,
                                                        u32* __restrict__ body_len, u32* __restrict__ crc,
                                                        u64* __restrict__ out_off, u32* __restrict__ out_cap,
                                                        u64* __restrict__ hdr /* total, tail status, chunks */)
{
    __shared__ u64 s_total;
    __shared__ u32 s_nc;
    if (threadIdx.x == 0) {
        u64 ip = 0, total = 0;
        u32 nc = 0;
        i32 tail = SNP_OK;
        while (ip < n) {
            if (n - ip < 4) { tail = SNP_ERR_TRUNCATED_STREAM; break; }
            u32 b[4] = {0, 0, 0, 0};                                    // 16 bytes at ip (fewer at the very end)
            if (n - ip >= 16) {
                const snp_u128_unaligned q = *reinterpret_cast<const snp_u128_unaligned*>(in + ip);
                b[0] = q.v[0]; b[1] = q.v[1]; b[2] = q.v[2]; b[3] = q.v[3];
            } else {
                for (u32 i = 0; i < n - ip; ++i) b[i >> 2] |= static_cast<u32>(in[ip + i]) << (8 * (i & 3));
            }
            const u32 t = b[0] & 0xffu;
            const u32 size = b[0] >> 8;                                 // :64-65
            ip += 4;
            if (n - ip < size) { tail = SNP_ERR_TRUNCATED_STREAM; break; }
            if (t <= 1) {
                if (size < 4) { tail = SNP_ERR_TRUNCATED_STREAM; break; }
                u32 dec = size - 4;
                if (t == 0) {                                           // block preamble  VarIntEncoding.Read.cs:38-79
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
                    if (bad || !done || result > 0x7fffffffu) { tail = SNP_ERR_BAD_LENGTH; break; }
                    dec = result;
                    // no tag expands more than 3 bytes -> 64: such a chunk can only end "Incomplete Snappy block." (capi_frame.hip scan_chunks)
                    if (static_cast<u64>(dec) > (static_cast<u64>(size - 4 - (shift / 7)) / 3 + 1) * 64) { tail = SNP_ERR_INCOMPLETE; break; }
                }
                if (nc == max_chunks) { tail = SNP_ERR_OUTPUT_TOO_SMALL; break; }   // chunk table full
                type[nc] = static_cast<u8>(t);
                body_off[nc] = ip + 4;
                body_len[nc] = size - 4;
                crc[nc] = b[1];                                         // ReadChunkCrc  :260-289
                out_off[nc] = total;
                out_cap[nc] = dec;
                total += dec;
                ++nc;
            } else if (t < 0x80) {                                      // :182-185
                tail = SNP_ERR_CHUNK_TYPE;
                break;
            }                                                           // 0x80..0xff skipped unvalidated  :187-196
            ip += size;
        }
        if (total > cap) { tail = SNP_ERR_OUTPUT_TOO_SMALL; nc = 0; total = 0; }   // nothing is decoded
        hdr[0] = total;
        hdr[1] = static_cast<u64>(static_cast<u32>(tail));
        hdr[2] = nc;
        s_total = total;
        s_nc = nc;
    }
    __syncthreads();
    const u64 total = s_total;
    for (u32 k = s_nc + threadIdx.x; k < max_chunks; k += SNP_WAVE) {
        type[k] = 1;
        body_off[k] = 0;
        body_len[k] = 0;
        crc[k] = kEmptyMaskedCrc;
        out_off[k] = total;
        out_cap[k] = 0;
    }
}

// The stream's verdict: the first failing chunk in stream order (as the sequential reference would throw), else the
// error that ended the header walk, else OK with the byte count.
__global__ __launch_bounds__(256) void k_frame_result(const i32* __restrict__ status, const u64* __restrict__ hdr,
                                                     u64* __restrict__ result)
{
    __shared__ u32 s_first;
    if (threadIdx.x == 0) s_first = 0xffffffffu;
    __syncthreads();
    const u32 nc = static_cast<u32>(hdr[2]);
    u32 first = 0xffffffffu;
    for (u32 k = threadIdx.x; k < nc; k += 256)
        if (status[k] != SNP_OK) { first = k; break; }
    if (first != 0xffffffffu) atomicMin(&s_first, first);
    __syncthreads();
    if (threadIdx.x == 0) {
        i32 st = static_cast<i32>(hdr[1]);
        if (s_first != 0xffffffffu) st = status[s_first];
        result[0] = st == SNP_OK ? hdr[0] : 0;
        result[1] = static_cast<u64>(static_cast<u32>(st));
    }
}

}  // namespace

extern "C" hipError_t snp_launch_gather(const u8* src, const u64* src_off, const u32* seg_len, u8* dst,
                                        const u64* dst_off, u32 nseg, hipStream_t stream)
{
    if (nseg == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather, dim3(nseg), dim3(256), 0, stream, src, src_off, seg_len, dst, dst_off, nseg);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_frame_chunks(u64 n, u32 nchunks, u64 comp_stride, u64* in_off, u32* in_len,
                                              u64* comp_off, hipStream_t stream)
{
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_frame_chunks, dim3((nchunks + 255) / 256), dim3(256), 0, stream, n, nchunks, comp_stride,
                       in_off, in_len, comp_off);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_frame_plan(const u32* in_len, const u32* comp_len, u32 nchunks, u8* type, u32* payload,
                                            u64* dst_off, u64* total, hipStream_t stream)
{
    hipLaunchKernelGGL(k_frame_plan, dim3(1), dim3(1024), 0, stream, in_len, comp_len, nchunks, type, payload, dst_off,
                       total);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_frame_header_only(u8* dst, u64* total, hipStream_t stream)
{
    hipLaunchKernelGGL(k_frame_header_only, dim3(1), dim3(64), 0, stream, dst, total);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_frame_emit(const u8* raw, const u64* in_off, const u8* comp, const u64* comp_off,
                                            const u8* type, const u32* payload, const u32* crc, const u64* dst_off,
                                            u8* dst, u64 cap, u32 nchunks, hipStream_t stream)
{
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_frame_emit, dim3(nchunks), dim3(256), 0, stream, raw, in_off, comp, comp_off, type, payload,
                       crc, dst_off, dst, cap, nchunks);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_frame_scan(const u8* in, u64 n, u64 cap, u32 max_chunks, u8* type, u64* body_off,
                                            u32* body_len, u32* crc, u64* out_off, u32* out_cap, u64* hdr,
                                            hipStream_t stream)
{
    hipLaunchKernelGGL(k_frame_scan, dim3(1), dim3(SNP_WAVE), 0, stream, in, n, cap, max_chunks, type, body_off, body_len,
                       crc, out_off, out_cap, hdr);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_frame_result(const i32* status, const u64* hdr, u64* result, hipStream_t stream)
{
    hipLaunchKernelGGL(k_frame_result, dim3(1), dim3(256), 0, stream, status, hdr, result);
    return hipGetLastError();
}
