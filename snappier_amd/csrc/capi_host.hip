// capi_host.hip -- the host-pointer calls of the C-ABI: snp_try_compress / snp_try_decompress (+ the segment forms for ReadOnlySequence<byte>)
// and snp_crc32c -- what a C# shim P/Invokes in place of the bodies of Snappy.TryCompress / TryDecompress / DecompressToMemory (Snappy.cs:55-67,
// 172-186,223-235).  Upload (sliced, overlapping the kernels), launch, download; one large block is decoded a wavefront per 64 KiB fragment via
// the tag index (tag_index.hip).
#include "capi_internal.h"

// Host input of nf 64 KiB fragments -> this->in, compressed into d_out.  Large inputs go up in slices on a copy stream and
// slice i is compressed while slice i+1 crosses PCIe (the fragments are independent; one launch per slice, small enough
// for the LDS-table kernel).  1 GiB: 57 -> 44 ms (profiles/r02i_host_api_rates.jsonl).
bool snp_ctx::upload_and_compress(const u8* host_in, size_t n, u32 nf, const u64* d_in_off, const u32* d_in_len, u8* d_out,
                         const u64* d_out_off, u32* d_out_len, i32* d_status, int emit_varint)
{
    if (nf < 4096 || !copy_stream_ready())
        return h2d(in.p, host_in, n, "H2D input") &&
               launch_compress(static_cast<const u8*>(in.p), d_in_off, d_in_len, nf, d_out, d_out_off, d_out_len, d_status, emit_varint);
    const u32 per = nf >= 16384 ? 4096u : (nf + 3) / 4;
    // the copy stream overwrites this->in: everything already queued on `stream` that reads it goes first (correctness must not
    // rest on the previous call having synchronised)
    bool ok = check(hipEventRecord(copy_ev[0], stream), "event") && check(hipStreamWaitEvent(copy_stream, copy_ev[0], 0), "wait");
    u32 k = 0;
    for (u32 first = 0; first < nf && ok; first += per, ++k) {
        const u32 cnt = nf - first < per ? nf - first : per;
        const size_t off = static_cast<size_t>(first) * SNP_BLOCK_SIZE;
        const size_t len = n - off < static_cast<size_t>(cnt) * SNP_BLOCK_SIZE ? n - off : static_cast<size_t>(cnt) * SNP_BLOCK_SIZE;
        hipEvent_t ev = copy_ev[k & 1];
        ok = check(hipMemcpyAsync(static_cast<u8*>(in.p) + off, host_in + off, len, hipMemcpyHostToDevice, copy_stream), "H2D input") &&
             check(hipEventRecord(ev, copy_stream), "event") && check(hipStreamWaitEvent(stream, ev, 0), "wait") &&
             launch_compress(static_cast<const u8*>(in.p), d_in_off + first, d_in_len + first, cnt, d_out, d_out_off + first,
                             d_out_len + first, d_status + first, emit_varint);
    }
    // a failed step returns to the caller, who may free or reuse host_in at once: no copy from it may still be in flight
    if (!ok) (void)hipStreamSynchronize(copy_stream);
    return ok;
}


namespace {
// The input of a host-pointer call: one span (snp_try_compress / snp_try_decompress) or the segments of a ReadOnlySequence
// (snp_try_*_segments).  Segments go up one after the other into ONE device buffer: the managed side never flattens them.
struct HostSpans {
    const uint8_t* const* ptr;
    const size_t* len;
    uint32_t count;
    bool valid(size_t* total) const
    {
        size_t t = 0;
        for (uint32_t i = 0; i < count; ++i) {
            if (len[i] && !ptr[i]) return false;
            if (len[i] > 0xffffffffull || t + len[i] > 0xffffffffull) { *total = ~size_t{0}; return true; }
            t += len[i];
        }
        *total = t;
        return true;
    }
    bool upload(snp_ctx* c, void* dev) const
    {
        size_t at = 0;
        for (uint32_t i = 0; i < count; ++i) {
            if (len[i] && !c->h2d(static_cast<u8*>(dev) + at, ptr[i], len[i], "H2D input segment")) return false;
            at += len[i];
        }
        return true;
    }
    void head(uint8_t* dst, size_t want) const                            // the first `want` bytes (the varint preamble)
    {
        size_t got = 0;
        for (uint32_t i = 0; i < count && got < want; ++i)
            for (size_t k = 0; k < len[i] && got < want; ++k) dst[got++] = ptr[i][k];
    }
};

snp_status compress_spans(snp_ctx* c, const HostSpans& in_spans, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    const uint8_t* in = in_spans.count == 1 ? in_spans.ptr[0] : nullptr;
    *written = 0;
    if (n > 0xffffffffull) return SNP_ERR_BAD_ARG;                       // SnappyCompressor.cs:88-91
    if (cap == 0) return SNP_ERR_OUTPUT_TOO_SMALL;                        // Snappy.cs:57-62
    for (uint32_t i = 0; i < in_spans.count; ++i)
        if (snp_ranges_overlap(in_spans.ptr[i], in_spans.len[i], out, cap)) return SNP_ERR_OVERLAP;   // SnappyCompressor.cs:27-30
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;

    u8 hdr[SNP_VARINT_MAX];                                               // VarIntEncoding.TryWrite  :34-37
    u32 hb = 0;
    for (u32 v = static_cast<u32>(n);;) {
        if (v < 128) { hdr[hb++] = static_cast<u8>(v); break; }
        hdr[hb++] = static_cast<u8>(v | 0x80);
        v >>= 7;
    }
    if (cap < hb) return SNP_ERR_OUTPUT_TOO_SMALL;
    const u32 nf = static_cast<u32>((n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE);
    if (nf == 0) { memcpy(out, hdr, hb); *written = hb; return SNP_OK; }

    hipStream_t s = c->stream;
    // meta: in_off, comp_off, dst_off (u64) ; in_len, comp_len (u32) ; status (i32)
    const u64 meta_bytes = static_cast<u64>(nf) * (8 * 3 + 4 * 3);
    if (!c->ensure(c->in, n, "hipMalloc(in)") || !c->ensure(c->work, nf * kSnpCompStride, "hipMalloc(work)") ||
        !c->ensure(c->meta, meta_bytes, "hipMalloc(meta)"))
        return SNP_ERR_DEVICE;
    u64* d_in_off = static_cast<u64*>(c->meta.p);
    u64* d_comp_off = d_in_off + nf;
    u64* d_dst_off = d_comp_off + nf;
    u32* d_in_len = reinterpret_cast<u32*>(d_dst_off + nf);
    u32* d_comp_len = d_in_len + nf;
    i32* d_status = reinterpret_cast<i32*>(d_comp_len + nf);

    bool ok = c->check(snp_launch_frame_chunks(n, nf, kSnpCompStride, d_in_off, d_in_len, d_comp_off, s), "fragment table");
    if (in)
        ok = ok && c->upload_and_compress(in, n, nf, d_in_off, d_in_len, static_cast<u8*>(c->work.p), d_comp_off, d_comp_len, d_status, 0);
    else
        ok = ok && in_spans.upload(c, c->in.p) &&
             c->launch_compress(static_cast<const u8*>(c->in.p), d_in_off, d_in_len, nf, static_cast<u8*>(c->work.p), d_comp_off, d_comp_len, d_status, 0);
    std::vector<u32> comp_len(nf);
    ok = ok && c->check(hipMemcpyAsync(comp_len.data(), d_comp_len, nf * 4ull, hipMemcpyDeviceToHost, s), "D2H lengths");
    ok = ok && c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;

    std::vector<u64> dst_off(nf);
    u64 total = 0;
    for (u32 f = 0; f < nf; ++f) { dst_off[f] = total; total += comp_len[f]; }
    if (cap - hb < total) return SNP_ERR_OUTPUT_TOO_SMALL;                 // SnappyCompressor.cs:63-68
    if (!c->ensure(c->out, total, "hipMalloc(out)")) return SNP_ERR_DEVICE;
    ok = c->check(hipMemcpyAsync(d_dst_off, dst_off.data(), nf * 8ull, hipMemcpyHostToDevice, s), "H2D offsets");
    ok = ok && c->check(snp_launch_gather(static_cast<const u8*>(c->work.p), d_comp_off, d_comp_len,
                                          static_cast<u8*>(c->out.p), d_dst_off, nf, s), "gather");
    ok = ok && c->d2h(out + hb, c->out.p, total, "D2H output");
    ok = ok && c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    memcpy(out, hdr, hb);
    *written = hb + total;
    return SNP_OK;
}

snp_status decompress_spans(snp_ctx* c, const HostSpans& in_spans, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    *written = 0;
    if (n > 0x7fffffffull) return SNP_ERR_BAD_ARG;                        // the reference's spans are int-length
    uint8_t in[8] = {0};                                                  // the preamble is read on the host
    in_spans.head(in, n < 5 ? n : 5);
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    const u32 cap32 = cap > 0x7fffffffull ? 0x7fffffffu : static_cast<u32>(cap);
    if (!c->ensure(c->in, n + 16, "hipMalloc(in)") || !c->ensure(c->out, static_cast<size_t>(cap32) + 16, "hipMalloc(out)") ||
        !c->ensure(c->meta, 64, "hipMalloc(meta)"))
        return SNP_ERR_DEVICE;
    struct Meta { u64 in_off, out_off; u32 in_len, out_cap, out_len; i32 status; } h{0, 0, static_cast<u32>(n), cap32, 0, 0};
    u8* m = static_cast<u8*>(c->meta.p);
    bool ok = c->check(hipMemcpyAsync(m, &h, sizeof(h), hipMemcpyHostToDevice, s), "H2D meta");

    // A large block: one wavefront per 64 KiB output fragment, fragment starts from the tag index (tag_index.hip).
    // Taken only for a clean preamble that fits the output; any fragment that does not come back OK (foreign streams
    // whose copies cross fragments, malformed data) sends the whole block to the single-wavefront decoder below.
    u32 expected = 0, hb = 0, shift = 0;
    bool clean = false;
    for (u32 i = 0; i < 5 && i < n; ++i) {                                // VarIntEncoding.Read.cs:38-79
        const u32 ch = in[i], val = ch & 0x7fu;
        if (val & ~(0xffffffffu >> shift)) break;
        expected |= val << shift;
        shift += 7;
        hb = i + 1;
        if (ch < 128) { clean = true; break; }
    }
    const bool large = clean && c->par_min && expected >= c->par_min && expected <= cap32 && n > hb;
    const u32 nf = large ? (expected + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE : 0;
    // fragment table: in_off, out_off (u64) ; in_len, out_cap, skip, out_len (u32) ; status (i32)
    if (large && (!c->ensure(c->work, snp_tag_index_workspace_bytes(static_cast<u32>(n), hb), "hipMalloc(tag index)") ||
                  !c->ensure(c->fragtab, static_cast<size_t>(nf) * (8 * 2 + 4 * 5), "hipMalloc(fragment table)")))
        return SNP_ERR_DEVICE;
    // The upload.  One large stream from one host buffer goes up in slices on the copy stream, and the tag index's per-chunk pass (most of its
    // time, and independent chunk by chunk) runs on each slice as it lands.
    bool indexed = false;
    if (ok && n && large && in_spans.count == 1 && n >= (8u << 20) && !snp_tag_index_look_back_only(static_cast<u32>(n), expected) && c->copy_stream_ready()) {
        const u8* const host_in = in_spans.ptr[0];
        const size_t slice = n / 8 > (4u << 20) ? (n / 8 + 4095) / 4096 * 4096 : (4u << 20);
        ok = c->check(snp_launch_tag_index_begin(static_cast<u64*>(c->work.p), static_cast<u32>(n), hb, 0, s), "tag index") &&
             c->check(hipEventRecord(c->copy_ev[0], s), "event") && c->check(hipStreamWaitEvent(c->copy_stream, c->copy_ev[0], 0), "wait");
        u32 done_chunks = 0, k = 0;
        for (size_t off = 0; off < n && ok; off += slice, ++k) {
            const size_t len = n - off < slice ? n - off : slice;
            hipEvent_t ev = c->copy_ev[k & 1];
            const u32 ready = snp_tag_index_chunks_ready(static_cast<u32>(n), hb, off + len);
            ok = c->check(hipMemcpyAsync(static_cast<u8*>(c->in.p) + off, host_in + off, len, hipMemcpyHostToDevice, c->copy_stream), "H2D input") &&
                 c->check(hipEventRecord(ev, c->copy_stream), "event") && c->check(hipStreamWaitEvent(s, ev, 0), "wait") &&
                 c->check(snp_launch_tag_index_chunks(static_cast<const u8*>(c->in.p), static_cast<u32>(n), hb, static_cast<u64*>(c->work.p),
                                                      done_chunks, ready - done_chunks, s), "tag index");
            done_chunks = ready;
        }
        if (!ok) (void)hipStreamSynchronize(c->copy_stream);              // (the caller may free its buffer at once: no copy from it may be in flight)
        indexed = ok;
    } else if (n) {
        ok = ok && in_spans.upload(c, c->in.p);
    }
    {
        if (ok && large) {
            const u32 nent = snp_tag_index_entries(static_cast<u32>(n), hb);
            u64* f_in_off = static_cast<u64*>(c->fragtab.p);
            u64* f_out_off = f_in_off + nf;
            u32* f_in_len = reinterpret_cast<u32*>(f_out_off + nf);
            u32* f_out_cap = f_in_len + nf;
            u32* f_skip = f_out_cap + nf;
            u32* f_out_len = f_skip + nf;
            i32* f_status = reinterpret_cast<i32*>(f_out_len + nf);
            ok = indexed ? c->check(snp_launch_tag_index_finish(static_cast<const u8*>(c->in.p), static_cast<u32>(n), hb, expected,
                                                                static_cast<u64*>(c->work.p), f_in_off, f_in_len, f_out_off, f_out_cap, f_skip, s),
                                    "tag index")
                         : c->check(snp_launch_tag_index(static_cast<const u8*>(c->in.p), static_cast<u32>(n), hb, expected,
                                                         static_cast<u64*>(c->work.p), f_in_off, f_in_len, f_out_off, f_out_cap, f_skip, s),
                                    "tag index");
            ok = ok && c->check(snp_launch_decompress(static_cast<const u8*>(c->in.p), f_in_off, f_in_len, nf,
                                                      static_cast<u8*>(c->out.p), f_out_off, f_out_cap, f_out_len, f_status,
                                                      nullptr, c->fenced | ((c->dec_lds / 256) << 8), s, f_skip),
                                "decompress fragments");
            std::vector<i32> st(nf);
            u32 looked_back = 0;
            ok = ok && c->check(hipMemcpyAsync(st.data(), f_status, nf * 4ull, hipMemcpyDeviceToHost, s), "D2H status");
            ok = ok && c->check(hipMemcpyAsync(&looked_back, static_cast<const u8*>(c->work.p) + snp_tag_index_fallback_offset(static_cast<u32>(n), hb), 4,
                                               hipMemcpyDeviceToHost, s), "D2H tag-index flag");
            ok = ok && c->check(hipStreamSynchronize(s), "sync");
            if (!ok) return SNP_ERR_DEVICE;
            bool all_ok = true;
            for (u32 f = 0; f < nf; ++f) all_ok = all_ok && st[f] == SNP_OK;
            ++c->counters[all_ok ? 0 : 1];
            if (looked_back) ++c->counters[6];
            if (!all_ok && SNP_GETENV("SNAPPIER_HIP_DEBUG")) {
                std::vector<u64> ent(nent), fo(nf);
                std::vector<u32> sk(nf), il(nf);
                (void)hipMemcpy(ent.data(), c->work.p, nent * 8ull, hipMemcpyDeviceToHost);
                (void)hipMemcpy(fo.data(), f_in_off, nf * 8ull, hipMemcpyDeviceToHost);
                (void)hipMemcpy(sk.data(), f_skip, nf * 4ull, hipMemcpyDeviceToHost);
                (void)hipMemcpy(il.data(), f_in_len, nf * 4ull, hipMemcpyDeviceToHost);
                fprintf(stderr, "[snappier] fragment decode fell back: n=%zu hb=%u expected=%u nent=%u last=(ip %u, op %u)\n", n, hb,
                        expected, nent, static_cast<u32>(ent[nent - 1]), static_cast<u32>(ent[nent - 1] >> 32) & 0x7fffffffu);
                for (u32 i = 0; i < nent && i < 12; ++i)
                    fprintf(stderr, "   entry %u: ip %u op %u\n", i, static_cast<u32>(ent[i]), static_cast<u32>(ent[i] >> 32) & 0x7fffffffu);
                u32 shown = 0;
                for (u32 f = 0; f < nf && shown < 8; ++f)
                    if (st[f] != SNP_OK) { fprintf(stderr, "   fragment %u: status %d in_off %llu in_len %u skip %u\n", f, st[f], (unsigned long long)fo[f], il[f], sk[f]); ++shown; }
            }
            if (all_ok) {
                ok = c->d2h(out, c->out.p, expected, "D2H output") &&
                     c->check(hipStreamSynchronize(s), "sync");
                if (!ok) return SNP_ERR_DEVICE;
                *written = expected;
                return SNP_OK;
            }
        }
    }
    ok = ok && c->check(snp_launch_decompress(static_cast<const u8*>(c->in.p), reinterpret_cast<u64*>(m + offsetof(Meta, in_off)),
                                              reinterpret_cast<u32*>(m + offsetof(Meta, in_len)), 1,
                                              static_cast<u8*>(c->out.p), reinterpret_cast<u64*>(m + offsetof(Meta, out_off)),
                                              reinterpret_cast<u32*>(m + offsetof(Meta, out_cap)),
                                              reinterpret_cast<u32*>(m + offsetof(Meta, out_len)),
                                              reinterpret_cast<i32*>(m + offsetof(Meta, status)), nullptr, c->fenced, s, nullptr),
                        "decompress");
    ok = ok && c->check(hipMemcpyAsync(&h, m, sizeof(h), hipMemcpyDeviceToHost, s), "D2H meta");
    ok = ok && c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    if (h.status != SNP_OK) return static_cast<snp_status>(h.status);
    if (h.out_len) {
        ok = c->d2h(out, c->out.p, h.out_len, "D2H output") &&
             c->check(hipStreamSynchronize(s), "sync");
        if (!ok) return SNP_ERR_DEVICE;
    }
    *written = h.out_len;
    return SNP_OK;
}
}  // namespace

extern "C" {

snp_status snp_try_compress(snp_ctx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    if (!c || !written || (n && !in) || (cap && !out)) return SNP_ERR_BAD_ARG;
    const HostSpans one{&in, &n, 1};
    return compress_spans(c, one, n, out, cap, written);
}

snp_status snp_try_decompress(snp_ctx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* written)
{
    if (!c || !written || (n && !in) || (cap && !out)) return SNP_ERR_BAD_ARG;
    const HostSpans one{&in, &n, 1};
    return decompress_spans(c, one, n, out, cap, written);
}

snp_status snp_try_compress_segments(snp_ctx* c, const uint8_t* const* seg, const size_t* seg_len, uint32_t nseg, uint8_t* out, size_t cap,
                                     size_t* written)
{
    if (!c || !written || (nseg && (!seg || !seg_len)) || (cap && !out)) return SNP_ERR_BAD_ARG;
    *written = 0;
    const HostSpans spans{seg, seg_len, nseg};
    size_t n = 0;
    if (!spans.valid(&n)) return SNP_ERR_BAD_ARG;
    if (n == ~size_t{0}) return SNP_ERR_BAD_ARG;                          // >= 2^32 bytes in all  SnappyCompressor.cs:88-91
    return compress_spans(c, spans, n, out, cap, written);
}

snp_status snp_try_decompress_segments(snp_ctx* c, const uint8_t* const* seg, const size_t* seg_len, uint32_t nseg, uint8_t* out, size_t cap,
                                       size_t* written)
{
    if (!c || !written || (nseg && (!seg || !seg_len)) || (cap && !out)) return SNP_ERR_BAD_ARG;
    *written = 0;
    const HostSpans spans{seg, seg_len, nseg};
    size_t n = 0;
    if (!spans.valid(&n) || n == ~size_t{0}) return SNP_ERR_BAD_ARG;
    return decompress_spans(c, spans, n, out, cap, written);
}

snp_status snp_crc32c(snp_ctx* c, const uint8_t* in, size_t n, int masked, uint32_t* out_crc)
{
    if (!c || !out_crc || (n && !in)) return SNP_ERR_BAD_ARG;
    if (n > 0xffffffffull) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t s = c->stream;
    if (!c->ensure(c->in, n + 16, "hipMalloc(in)") || !c->ensure(c->meta, 64, "hipMalloc(meta)")) return SNP_ERR_DEVICE;
    struct Meta { u64 off; u32 len, crc; } h{0, static_cast<u32>(n), 0};
    u8* m = static_cast<u8*>(c->meta.p);
    bool ok = c->check(hipMemcpyAsync(m, &h, sizeof(h), hipMemcpyHostToDevice, s), "H2D meta");
    if (n) ok = ok && c->h2d(c->in.p, in, n, "H2D input");
    ok = ok && c->check(snp_launch_crc32c(static_cast<const u8*>(c->in.p), reinterpret_cast<u64*>(m), reinterpret_cast<u32*>(m + 8),
                                          1, (masked ? 1 : 0) | c->crc_bits(), reinterpret_cast<u32*>(m + 12), nullptr, nullptr, s), "crc32c");
    ok = ok && c->check(hipMemcpyAsync(&h, m, sizeof(h), hipMemcpyDeviceToHost, s), "D2H meta");
    ok = ok && c->check(hipStreamSynchronize(s), "sync");
    if (!ok) return SNP_ERR_DEVICE;
    *out_crc = h.crc;
    return SNP_OK;
}

}  // extern "C"
