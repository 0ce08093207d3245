// capi_frame.hip -- the framing format end to end (SnappyStreamCompressor.cs:18-21,166-261; SnappyStreamDecompressor.cs:38-208): orchestration of
// compress + masked CRC + raw-vs-compressed decision + scan + emit on the way in, header walk + decode + CRC verify on the way out.  The kernels are in
// framing.hip, frame_scan.hip, crc32c.hip and the codec files; this file only sequences them and moves host buffers.
#include "capi_internal.h"

// workspace layout for snp_frame_encode_device (all sub-arrays 16-byte aligned)
struct FrameWork {
    u64 *in_off, *comp_off, *dst_off, *total;
    u32 *in_len, *comp_len, *payload, *crc;
    i32* status;
    u8 *type, *comp;
    u64 bytes;
};
static FrameWork frame_work_layout(void* base, u64 n)
{
    const u64 nc = (n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE;
    u8* p = static_cast<u8*>(base);
    u64 o = 0;
    FrameWork w{};
    auto take = [&](u64 bytes) { u8* r = p ? p + o : nullptr; o += snp_align_up(bytes, 16); return r; };
    w.in_off = reinterpret_cast<u64*>(take(nc * 8));
    w.comp_off = reinterpret_cast<u64*>(take(nc * 8));
    w.dst_off = reinterpret_cast<u64*>(take((nc + 1) * 8));
    w.total = reinterpret_cast<u64*>(take(8));
    w.in_len = reinterpret_cast<u32*>(take(nc * 4));
    w.comp_len = reinterpret_cast<u32*>(take(nc * 4));
    w.payload = reinterpret_cast<u32*>(take(nc * 4));
    w.crc = reinterpret_cast<u32*>(take(nc * 4));
    w.status = reinterpret_cast<i32*>(take(nc * 4));
    w.type = take(nc);
    w.comp = take(nc * kSnpCompStride);
    w.bytes = o;
    return w;
}


// host_in != nullptr: the raw stream is still in host memory; it is uploaded into d_in in slices that overlap the compressor
snp_status snp_frame_encode_impl(snp_ctx* c, const uint8_t* d_in, const uint8_t* host_in, uint64_t n, uint8_t* d_out,
                                    uint64_t cap, uint64_t* d_written, void* d_work)
{
    if (cap < SNP_STREAM_HEADER_LEN) return SNP_ERR_OUTPUT_TOO_SMALL;
    const u32 nc = static_cast<u32>((n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE);
    hipStream_t s = c->stream;
    if (nc == 0)
        return c->check(snp_launch_frame_header_only(d_out, d_written, s), "frame header") ? SNP_OK : SNP_ERR_DEVICE;
    const FrameWork w = frame_work_layout(d_work, n);
    bool ok = c->check(snp_launch_frame_chunks(n, nc, kSnpCompStride, w.in_off, w.in_len, w.comp_off, s), "frame chunks");
    // CompressBlock: TryCompress(chunk) = varint + one fragment  (SnappyStreamCompressor.cs:206)
    if (host_in) ok = ok && c->upload_and_compress(host_in, n, nc, w.in_off, w.in_len, w.comp, w.comp_off, w.comp_len, w.status, 1);
    else ok = ok && c->launch_compress(d_in, w.in_off, w.in_len, nc, w.comp, w.comp_off, w.comp_len, w.status, 1);
    // masked CRC-32C of the RAW chunk  (:243-245,258-260)
    ok = ok && c->check(snp_launch_crc32c(d_in, w.in_off, w.in_len, nc, 1 | c->crc_bits(), w.crc, nullptr, nullptr, s), "frame crc");
    ok = ok && c->check(snp_launch_frame_plan(w.in_len, w.comp_len, nc, w.type, w.payload, w.dst_off, d_written, s),
                        "frame plan");
    ok = ok && c->check(snp_launch_frame_emit(d_in, w.in_off, w.comp, w.comp_off, w.type, w.payload, w.crc, w.dst_off,
                                              d_out, cap, nc, s), "frame emit");
    return ok ? SNP_OK : SNP_ERR_DEVICE;
}


extern "C" {

uint64_t snp_frame_encode_workspace(uint64_t n) { return frame_work_layout(nullptr, n).bytes; }

snp_status snp_frame_encode_device(snp_ctx* c, const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap,
                                   uint64_t* d_written, void* d_work)
{
    if (!c || !d_out || !d_written || !d_work || (n && !d_in)) return SNP_ERR_BAD_ARG;
    if (n > 0xffffffffull * SNP_BLOCK_SIZE) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    return snp_frame_encode_impl(c, d_in, nullptr, n, d_out, cap, d_written, d_work);
}

snp_status snp_frame_decode_chunks_device(snp_ctx* c, const uint8_t* d_in, const uint8_t* chunk_type,
                                          const uint64_t* body_off, const uint32_t* body_len,
                                          const uint32_t* chunk_crc, uint32_t nchunks, uint8_t* d_out,
                                          const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len,
                                          int32_t* status)
{
    if (!c || (nchunks && (!d_in || !chunk_type || !body_off || !body_len || !chunk_crc || !d_out || !out_off ||
                           !out_cap || !out_len || !status)))
        return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    bool ok = c->launch_decompress(d_in, body_off, body_len, nchunks, d_out, out_off, out_cap, out_len, status, chunk_type);
    // CRC over the produced bytes, compared with the chunk's stored masked CRC  (SnappyStreamDecompressor.cs:117-131)
    ok = ok && c->check(snp_launch_crc32c(d_out, out_off, out_len, nchunks, 1 | c->crc_bits(), nullptr, chunk_crc, status, s),
                        "frame crc verify");
    return ok ? SNP_OK : SNP_ERR_DEVICE;
}

// ---- framed stream without a chunk table: header walk on the device --------------------------------
// workspace: 64-byte header {total, tail status, chunks} ; body_off, out_off (u64) ; body_len, crc, out_cap, out_len (u32) ;
// status (i32) ; type (u8)
uint64_t snp_frame_decode_workspace(uint32_t max_chunks)
{
    return 64 + snp_align_up(static_cast<u64>(max_chunks) * (8 * 2 + 4 * 5 + 1), 16) + 16;
}

snp_status snp_frame_decode_device(snp_ctx* c, const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t cap,
                                   uint32_t max_chunks, void* d_work, uint64_t* d_result)
{
    if (!c || !d_work || !d_result || (n && !d_in) || (cap && !d_out)) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    u64* hdr = static_cast<u64*>(d_work);
    u64* body_off = hdr + 8;
    u64* out_off = body_off + max_chunks;
    u32* body_len = reinterpret_cast<u32*>(out_off + max_chunks);
    u32* crc = body_len + max_chunks;
    u32* out_cap = crc + max_chunks;
    u32* out_len = out_cap + max_chunks;
    i32* status = reinterpret_cast<i32*>(out_len + max_chunks);
    u8* type = reinterpret_cast<u8*>(status + max_chunks);
    bool ok;
    if (c->frame_scan == 1) {
        ok = c->check(snp_launch_frame_scan(d_in, n, cap, max_chunks, type, body_off, body_len, crc, out_off, out_cap, hdr, s), "frame scan");
    } else {
        // per-span candidate tables live in context scratch (a few MB per 10 GiB of stream; grows on first use only)
        if (!c->ensure(c->scan, snp_frame_scan_workspace(n), "hipMalloc(frame scan)")) return SNP_ERR_DEVICE;
        ok = c->check(snp_launch_frame_scan_spans(d_in, n, cap, max_chunks, type, body_off, body_len, crc, out_off, out_cap, hdr,
                                                  c->scan.p, s), "frame scan (spans)");
    }
    if (ok && max_chunks && n) {                     // n == 0: no chunk, nothing to launch
        const snp_status st = snp_frame_decode_chunks_device(c, d_in, type, body_off, body_len, crc, max_chunks, d_out, out_off,
                                                             out_cap, out_len, status);
        if (st != SNP_OK) return st;
    }
    ok = ok && c->check(snp_launch_frame_result(status, hdr, d_result, s), "frame result");
    return ok ? SNP_OK : SNP_ERR_DEVICE;
}

snp_status snp_frame_encode(snp_ctx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    if (!c || !written || (n && !in) || (cap && !out)) return SNP_ERR_BAD_ARG;
    *written = 0;
    if (snp_ranges_overlap(in, n, out, cap)) return SNP_ERR_OVERLAP;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    const u64 max_out = static_cast<u64>(snp_frame_max_encoded_length(static_cast<int64_t>(n)));
    const u64 wbytes = snp_frame_encode_workspace(n);
    if (!c->ensure(c->in, n + 16, "hipMalloc(in)") || !c->ensure(c->out, max_out + 16, "hipMalloc(out)") ||
        !c->ensure(c->work, wbytes + 16, "hipMalloc(work)") || !c->ensure(c->meta, 64, "hipMalloc(meta)"))
        return SNP_ERR_DEVICE;
    bool ok = true;
    if (n > 0xffffffffull * SNP_BLOCK_SIZE) return SNP_ERR_BAD_ARG;
    snp_status st = snp_frame_encode_impl(c, static_cast<const u8*>(c->in.p), n ? in : nullptr, n, static_cast<u8*>(c->out.p), max_out,
                                      static_cast<u64*>(c->meta.p), c->work.p);
    if (st != SNP_OK) return st;
    u64 total = 0;
    ok = c->check(hipMemcpyAsync(&total, c->meta.p, 8, hipMemcpyDeviceToHost, s), "D2H total") &&
         c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    if (total > cap) return SNP_ERR_OUTPUT_TOO_SMALL;
    ok = c->d2h(out, c->out.p, total, "D2H output") &&
         c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    *written = total;
    return SNP_OK;
}

// Host-side walk over chunk headers only (SnappyStreamDecompressor.ReadChunkHeader  :215-254): 4 bytes per chunk.
struct ChunkScan {
    std::vector<u8> type;
    std::vector<u64> body_off, out_off;
    std::vector<u32> body_len, crc, out_cap;
    u64 total = 0;
    snp_status tail = SNP_OK;   // error met after the chunks listed above (they are still decoded and checked first)
};
static inline u64 snp_max_expansion(u64 body_bytes) { return (body_bytes / 3 + 1) * 64; }
static void scan_chunks(const u8* in, size_t n, ChunkScan& cs)
{
    size_t ip = 0;
    while (ip < n) {
        if (n - ip < 4) { cs.tail = SNP_ERR_TRUNCATED_STREAM; return; }
        const u32 type = in[ip];
        const u32 size = in[ip + 1] | (in[ip + 2] << 8) | (static_cast<u32>(in[ip + 3]) << 16);   // :64-65
        ip += 4;
        if (n - ip < size) { cs.tail = SNP_ERR_TRUNCATED_STREAM; return; }
        if (type == 0x00 || type == 0x01) {
            if (size < 4) { cs.tail = SNP_ERR_TRUNCATED_STREAM; return; }
            u32 crc;
            memcpy(&crc, in + ip, 4);                                    // ReadChunkCrc  :260-289
            u32 dec = size - 4, hb = 0;
            if (type == 0x00 && snp_get_uncompressed_length(in + ip + 4, size - 4, &dec, &hb) != SNP_OK) {
                cs.tail = SNP_ERR_BAD_LENGTH;
                return;
            }
            if (dec > 0x7fffffffu) { cs.tail = SNP_ERR_BAD_LENGTH; return; }
            // No tag expands more than 3 bytes -> 64 (a copy-2 of length 64): a chunk that declares more than its body can
            // possibly produce is "Incomplete Snappy block." whatever its tags say -- and must not size any allocation.
            if (type == 0x00 && dec > snp_max_expansion(size - 4 - hb)) { cs.tail = SNP_ERR_INCOMPLETE; return; }
            cs.type.push_back(static_cast<u8>(type));
            cs.body_off.push_back(ip + 4);
            cs.body_len.push_back(size - 4);
            cs.crc.push_back(crc);
            cs.out_off.push_back(cs.total);
            cs.out_cap.push_back(dec);
            cs.total += dec;
        } else if (type < 0x80) {                                        // :182-185
            cs.tail = SNP_ERR_CHUNK_TYPE;
            return;
        }                                                                // 0x80..0xff skipped unvalidated  :187-196
        ip += size;
    }
}

snp_status snp_frame_decoded_length(const uint8_t* in, size_t n, uint64_t* out_len)
{
    if (!out_len || (n && !in)) return SNP_ERR_BAD_ARG;
    ChunkScan cs;
    scan_chunks(in, n, cs);
    *out_len = cs.total;
    return cs.tail;
}

snp_status snp_frame_decode(snp_ctx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    if (!c || !written || (n && !in) || (cap && !out)) return SNP_ERR_BAD_ARG;
    *written = 0;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    ChunkScan cs;
    scan_chunks(in, n, cs);
    const u32 nc = static_cast<u32>(cs.type.size());
    if (cs.total > cap) return SNP_ERR_OUTPUT_TOO_SMALL;
    if (nc == 0) return cs.tail;
    hipStream_t s = c->stream;
    // meta: body_off, out_off (u64) ; body_len, crc, out_cap, out_len (u32) ; status (i32) ; type (u8)
    const u64 meta_bytes = static_cast<u64>(nc) * (8 * 2 + 4 * 5 + 1) + 64;
    if (!c->ensure(c->in, n + 16, "hipMalloc(in)") || !c->ensure(c->out, cs.total + 16, "hipMalloc(out)") ||
        !c->ensure(c->meta, meta_bytes, "hipMalloc(meta)"))
        return SNP_ERR_DEVICE;
    u64* d_body_off = static_cast<u64*>(c->meta.p);
    u64* d_out_off = d_body_off + nc;
    u32* d_body_len = reinterpret_cast<u32*>(d_out_off + nc);
    u32* d_crc = d_body_len + nc;
    u32* d_out_cap = d_crc + nc;
    u32* d_out_len = d_out_cap + nc;
    i32* d_status = reinterpret_cast<i32*>(d_out_len + nc);
    u8* d_type = reinterpret_cast<u8*>(d_status + nc);
    bool ok = c->h2d(c->in.p, in, n, "H2D input");
    ok = ok && c->check(hipMemcpyAsync(d_body_off, cs.body_off.data(), nc * 8ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_out_off, cs.out_off.data(), nc * 8ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_body_len, cs.body_len.data(), nc * 4ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_crc, cs.crc.data(), nc * 4ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_out_cap, cs.out_cap.data(), nc * 4ull, hipMemcpyHostToDevice, s), "H2D meta");
    ok = ok && c->check(hipMemcpyAsync(d_type, cs.type.data(), nc, hipMemcpyHostToDevice, s), "H2D meta");
    if (!ok) return SNP_ERR_DEVICE;
    snp_status st = snp_frame_decode_chunks_device(c, static_cast<const u8*>(c->in.p), d_type, d_body_off, d_body_len,
                                                   d_crc, nc, static_cast<u8*>(c->out.p), d_out_off, d_out_cap, d_out_len,
                                                   d_status);
    if (st != SNP_OK) return st;
    std::vector<i32> status(nc);
    ok = c->check(hipMemcpyAsync(status.data(), d_status, nc * 4ull, hipMemcpyDeviceToHost, s), "D2H status") &&
         c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    for (u32 i = 0; i < nc; ++i)                                          // first failing chunk in stream order wins,
        if (status[i] != SNP_OK) return static_cast<snp_status>(status[i]);   // as the sequential reference would throw
    if (cs.tail != SNP_OK) return cs.tail;
    if (cs.total) {
        ok = c->d2h(out, c->out.p, cs.total, "D2H output") &&
             c->check(hipStreamSynchronize(s), "sync");
        if (!ok) return SNP_ERR_DEVICE;
    }
    *written = cs.total;
    return SNP_OK;
}

}  // extern "C"
