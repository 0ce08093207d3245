// capi_ctx.hip -- contexts of the C-ABI (include/snappier_hip.h): create / destroy / stream / options / counters, the context's scratch
// buffers and transfers, and the host-only arithmetic (varint, MaxCompressedLength).  See capi_internal.h for how the host side is split.
#include "capi_internal.h"

bool snp_ctx::check(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

bool snp_ctx::ensure(DevBuf& b, size_t bytes, const char* what)
{
    if (bytes <= b.cap) return true;
    if (stream_is_capturing()) {                                    // hipFree / hipMalloc would invalidate the capture: refuse and leave it intact
        err = std::string(what) + ": a workspace would have to grow while the stream is being captured -- make the same call once before the capture (or snp_ctx_reserve_compress)";
        return false;
    }
    if (b.p) {
        if (was_captured) kept.push_back(b.p);                      // a graph captured earlier may still hold this address: kept until snp_ctx_destroy
        else (void)hipFree(b.p);
    }
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    if (!check(hipMalloc(&b.p, want), what)) return false;
    b.cap = want;
    return true;
}

// Host <-> device transfers of the host-pointer entry points: the caller's buffers are pageable (the reference's Span API)
// and the runtime's own pageable path moves them at ~40 GB/s; a pinned-slice pipeline inside the library measured slower
// (31 GB/s, profiles/r02b_host_api_rates.jsonl) and was removed.
bool snp_ctx::h2d(void* dev, const void* host, size_t n, const char* what)
{
    return check(hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, stream), what);
}

bool snp_ctx::d2h(void* host, const void* dev, size_t n, const char* what)
{
    return check(hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, stream), what) && check(hipStreamSynchronize(stream), what);
}

bool snp_ctx::stream_is_capturing()
{
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(stream, &cap_st) != hipSuccess || cap_st != hipStreamCaptureStatusNone;
    if (capturing) { (void)hipGetLastError(); was_captured = true; }
    return capturing;
}

// The context's scratch (hash tables, staging) is ordered by the stream it runs on.  Moving the context to another
// stream: everything already queued on the old stream must finish before the new stream touches the scratch.
bool snp_ctx::rebind(hipStream_t next)
{
    if (next == stream) return true;
    if (!order_ev && !check(hipEventCreateWithFlags(&order_ev, hipEventDisableTiming), "hipEventCreate")) return false;
    return check(hipEventRecord(order_ev, stream), "hipEventRecord") &&
           check(hipStreamWaitEvent(next, order_ev, 0), "hipStreamWaitEvent");
}

bool snp_ctx::copy_stream_ready()
{
    if (copy_state) return copy_state > 0;
    copy_state = -1;
    if (hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); copy_stream = nullptr; return false; }
    for (auto& e : copy_ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    copy_state = 1;
    return true;
}

#ifdef SNAPPIER_HIP_DEBUG_ENV
// The SNAPPIER_HIP_* knobs of the A/B scripts (scripts/): initial values of a context in VARIANT builds (scripts/build_variant.sh, LAB=1).
static void snp_apply_debug_env(snp_ctx* c)
{
    const char* f = SNP_GETENV("SNAPPIER_HIP_FENCED");
    c->fenced = (f && f[0] == '0') ? 0 : 1;
    const char* m = SNP_GETENV("SNAPPIER_HIP_DECODE");
    if (m && strcmp(m, "serial") == 0) c->fenced |= 2;
    if (m && strcmp(m, "batched") == 0) c->fenced |= 4;                 // token-parallel batches without the execution queue
    if (m && strcmp(m, "ring") == 0) c->fenced |= 8 | 32;              // sub-chain parse + output-granular execution through an LDS ring (FRONT = 4)
    else if (m && strcmp(m, "chains_r04") == 0) c->fenced |= 8 | 64;   // the round-4 form of the default decoder (A/B)
    else if (!m || strcmp(m, "chains") == 0 || (strcmp(m, "queued") != 0 && strcmp(m, "serial") != 0 && strcmp(m, "batched") != 0))
        c->fenced |= 8;                                                 // default: sub-chain parse (decompress.hip, FRONT = 3); "queued" = 64-byte windows + queue
    c->decode_layout = (m && (strcmp(m, "serial") == 0 || strcmp(m, "batched") == 0)) ? 1 : 0;
    // SNAPPIER_HIP_DEC_LDS=<bytes>: dynamic LDS per decode wavefront, an occupancy throttle (160 KiB / bytes blocks per CU)
    const char* dl = SNP_GETENV("SNAPPIER_HIP_DEC_LDS");
    c->dec_lds = dl ? (atoi(dl) / 256) * 256 : 0;
    // SNAPPIER_HIP_COMPRESS=win|lanes pins the compressor layout (default: by batch size)
    const char* cm = SNP_GETENV("SNAPPIER_HIP_COMPRESS");
    c->compress_mode = (cm && strcmp(cm, "lanes") == 0) ? 2 : (cm && strcmp(cm, "wing") == 0) ? 4 : (cm && strncmp(cm, "win", 3) == 0) ? 3 : 0;
    const char* wg = SNP_GETENV("SNAPPIER_HIP_WIN_GTAB_MIN");
    if (wg) c->win_gtab_min = static_cast<u32>(strtoul(wg, nullptr, 10));
    const char* wn = SNP_GETENV("SNAPPIER_HIP_WIN_NP");
    if (wn) c->win_np = atoi(wn) == 2 ? 2 : 1;
    const char* fs = SNP_GETENV("SNAPPIER_HIP_FRAME_SCAN");
    c->frame_scan = (fs && strcmp(fs, "serial") == 0) ? 1 : 0;
    const char* sm = SNP_GETENV("SNAPPIER_HIP_SMALL_MAX");
    if (sm) c->small_max = static_cast<u32>(strtoul(sm, nullptr, 10));
    const char* sl = SNP_GETENV("SNAPPIER_HIP_SMALL");
    c->small_lanes = sl && strcmp(sl, "lanes") == 0;
    c->small_team_log = (sl && strcmp(sl, "team4") == 0) ? 2 : (sl && strcmp(sl, "team8") == 0) ? 3 : (sl && strcmp(sl, "team16") == 0) ? 4 : 0;
    const char* rg = SNP_GETENV("SNAPPIER_HIP_REDO");
    c->redo_grid = rg && strcmp(rg, "grid") == 0;
    c->redo_list = rg && strcmp(rg, "list") == 0;
    const char* sn = SNP_GETENV("SNAPPIER_HIP_SMALL_MIN");
    if (sn) c->small_min_blocks = static_cast<u32>(strtoul(sn, nullptr, 10));
    const char* sf = SNP_GETENV("SNAPPIER_HIP_SLICE");
    if (sf && atoi(sf) >= 4096) c->slice_fragments = static_cast<u32>(atoi(sf));
    const char* wm = SNP_GETENV("SNAPPIER_HIP_WIN_MAX");
    if (wm) c->win_max = static_cast<u32>(strtoul(wm, nullptr, 10));
    // SNAPPIER_HIP_PARALLEL_MIN=<bytes>: declared length from which snp_try_decompress splits ONE block into 64 KiB
    // fragments decoded in parallel (tag_index.hip); 0 = always one wavefront per block
    const char* tt = SNP_GETENV("SNAPPIER_HIP_TABLE_TRIES");
    if (tt) { c->table_tries = atoi(tt) < 1 ? 1 : atoi(tt) > 24 ? 24 : atoi(tt); c->table_tries_set = true; }
    const char* pm = SNP_GETENV("SNAPPIER_HIP_PARALLEL_MIN");
    if (pm) c->par_min = static_cast<u32>(strtoul(pm, nullptr, 10));
}
#endif

extern "C" {

// ---- context ---------------------------------------------------------------------------------------------

snp_status snp_ctx_create(int device, int hash_variant, void* stream, snp_ctx** out_ctx)
{
    if (!out_ctx || (hash_variant != SNP_HASH_CRC32C && hash_variant != SNP_HASH_MUL)) return SNP_ERR_BAD_ARG;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return SNP_ERR_DEVICE;
    snp_ctx* c = new (std::nothrow) snp_ctx();
    if (!c) return SNP_ERR_DEVICE;
    c->device = device;
    c->variant = hash_variant;
    DevGuard dg(c);
    if (!dg.ok) { delete c; return SNP_ERR_DEVICE; }
    if (stream) c->stream = static_cast<hipStream_t>(stream);
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SNP_ERR_DEVICE; }
        c->own_stream = true;
    }
    c->pool = snp_pool_of(device);
    if (!c->pool) { if (c->own_stream) (void)hipStreamDestroy(c->stream); delete c; return SNP_ERR_DEVICE; }   // (an ordinal beyond the pool table: no two devices share a workspace)
    {
        std::lock_guard<std::mutex> g(c->pool->mu);
        ++c->pool->users;
    }
    // FENCED (a wavefront drains vmcnt before it reads output bytes it stored itself) is the default: measured 0.9 % slower than relying on
    // in-order vector memory (17.22 vs 17.38 ms per 10 GiB, profiles/r02c_fenced_ab.jsonl); SNP_OPT_FENCED = 0 turns it off.
    c->fenced = 1 | 8;                                                  // (bit 3: the default front end, decode_chains.hip; see capi_internal.h)
#ifdef SNAPPIER_HIP_DEBUG_ENV
    snp_apply_debug_env(c);                                             // variant builds only: the product library reads no environment
#endif
    *out_ctx = c;
    return SNP_OK;
}

uint64_t snp_ctx_counter(const snp_ctx* c, int which) { return (c && which >= 0 && which < 7) ? c->counters[which] : 0; }

snp_status snp_ctx_reserve_compress(snp_ctx* c, uint32_t nfragments)
{
    if (!c) return SNP_ERR_BAD_ARG;
    if (nfragments == 0) return SNP_OK;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    if (!c->borrow_tables(nfragments < c->slice_fragments ? nfragments : c->slice_fragments, true)) return SNP_ERR_DEVICE;
    c->return_tables();
    return SNP_OK;
}

snp_status snp_ctx_set_option(snp_ctx* c, int option, int64_t v)
{
    if (!c) return SNP_ERR_BAD_ARG;
    switch (option) {
        case SNP_OPT_DECODE_LAYOUT:
            if (v < 0 || v > 6) return SNP_ERR_BAD_ARG;
            c->no_prepass = v == 1;
            c->small_lanes = v == 2;
            c->small_team_log = (v >= 3 && v <= 5) ? static_cast<u32>(v - 1) : 0u;   // 3 / 4 / 5 -> teams of 4 / 8 / 16 lanes
            c->fenced = (c->fenced & ~2) | (v == 6 ? 2 : 0);                          // 6: the serial kernel (the reference's loop as it stands), every block
            c->decode_layout = v == 6 ? 1 : 0;
            return SNP_OK;
        case SNP_OPT_SMALL_BLOCK_MAX:
            if (v < 0 || v > 0x0fffffff) return SNP_ERR_BAD_ARG;
            c->small_max = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_SMALL_BLOCK_MIN_BATCH:
            if (v < 1 || v > 0xffffffffll) return SNP_ERR_BAD_ARG;
            c->small_min_blocks = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_LAYOUT:
            if (v != 0 && v != 2 && v != 3 && v != 4 && v != 5) return SNP_ERR_BAD_ARG;
            c->compress_mode = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_MAX_BATCH:
            if (v < 0 || v > 0xffffffffll) return SNP_ERR_BAD_ARG;
            c->win_max = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_TABLE_PROBE_TRIES:
            if (v < 1 || v > 24) return SNP_ERR_BAD_ARG;
            c->table_tries = static_cast<int>(v);
            c->table_tries_set = true;
            return SNP_OK;
        case SNP_OPT_TABLE_PROBE_MAX_BYTES:
            if (v < 0) return SNP_ERR_BAD_ARG;
            c->table_probe_max_bytes = static_cast<uint64_t>(v);
            return SNP_OK;
        case SNP_OPT_PARALLEL_DECODE_MIN:
            if (v < 0 || v > 0x7fffffff) return SNP_ERR_BAD_ARG;
            c->par_min = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_FENCED:
            if (v != 0 && v != 1) return SNP_ERR_BAD_ARG;
            c->fenced = (c->fenced & ~1) | static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_DECODE_LEFTOVERS:
            if (v < 0 || v > 2) return SNP_ERR_BAD_ARG;
            c->redo_grid = v == 1;
            c->redo_list = v == 2;
            return SNP_OK;
        case SNP_OPT_CRC_KERNEL:
            if (v < 0 || v > 2) return SNP_ERR_BAD_ARG;
            c->crc_kernel = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_POSITIONS:
            if (v != 1 && v != 2) return SNP_ERR_BAD_ARG;
            c->win_np = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_GLOBAL_MIN_BATCH:
            if (v < 0 || v > 0xffffffffll) return SNP_ERR_BAD_ARG;
            c->win_gtab_min = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_LANE_STORES:
            if (v < -1 || v > 255) return SNP_ERR_BAD_ARG;
            c->lane_tune.opts = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_LANE_PROBES:
            if (v < 0 || v > 4) return SNP_ERR_BAD_ARG;
            c->lane_tune.probes = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_LANES_PER_WAVEFRONT:
            if (v != 0 && v != 8 && v != 16 && v != 32 && v != 64) return SNP_ERR_BAD_ARG;
            c->lane_tune.lanes_per_wave = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_SLICE:
            if (v < 4096 || v > 0x7fffffff) return SNP_ERR_BAD_ARG;
            c->slice_fragments = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_SMALL_INPUT_LDS:
            if (v < -1 || v > 2048) return SNP_ERR_BAD_ARG;
            c->lane_tune.small_bytes = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_SMALL_INPUT_LANES:
            if (v != 0 && v != 16 && v != 32 && v != 64) return SNP_ERR_BAD_ARG;
            c->lane_tune.small_lanes = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_FRAME_SCAN:
            if (v != 0 && v != 1) return SNP_ERR_BAD_ARG;
            c->frame_scan = static_cast<int>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_GLOBAL_SLOTS:
            if (v < 0 || v > 65536) return SNP_ERR_BAD_ARG;
            c->win_gslots = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_DUAL_MIN_BATCH:
            if (v < 0 || v > 0xffffffffll) return SNP_ERR_BAD_ARG;
            c->win_dual_min = static_cast<u32>(v);
            return SNP_OK;
        case SNP_OPT_DECODE_LDS_THROTTLE:
            if (v < 0 || v > 65536) return SNP_ERR_BAD_ARG;
            c->dec_lds = static_cast<int>(v / 256 * 256);
            return SNP_OK;
        default:
            return SNP_ERR_BAD_ARG;
    }
}

snp_status snp_ctx_get_option(const snp_ctx* c, int option, int64_t* out)
{
    if (!c || !out) return SNP_ERR_BAD_ARG;
    switch (option) {
        case SNP_OPT_DECODE_LAYOUT: *out = (c->fenced & 2) ? 6 : c->no_prepass ? 1 : c->small_lanes ? 2 : c->small_team_log ? c->small_team_log + 1 : 0; return SNP_OK;
        case SNP_OPT_SMALL_BLOCK_MAX: *out = c->small_max; return SNP_OK;
        case SNP_OPT_SMALL_BLOCK_MIN_BATCH: *out = c->small_min_blocks; return SNP_OK;
        case SNP_OPT_COMPRESS_LAYOUT: *out = c->compress_mode; return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_MAX_BATCH: *out = c->win_max; return SNP_OK;
        case SNP_OPT_TABLE_PROBE_TRIES: *out = c->table_tries; return SNP_OK;
        case SNP_OPT_TABLE_PROBE_MAX_BYTES: *out = static_cast<int64_t>(c->table_probe_max_bytes); return SNP_OK;
        case SNP_OPT_PARALLEL_DECODE_MIN: *out = c->par_min; return SNP_OK;
        case SNP_OPT_FENCED: *out = c->fenced & 1; return SNP_OK;
        case SNP_OPT_DECODE_LEFTOVERS: *out = c->redo_grid ? 1 : c->redo_list ? 2 : 0; return SNP_OK;
        case SNP_OPT_CRC_KERNEL: *out = c->crc_kernel; return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_POSITIONS: *out = c->win_np; return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_GLOBAL_MIN_BATCH: *out = c->win_gtab_min; return SNP_OK;
        case SNP_OPT_COMPRESS_LANE_STORES: *out = c->lane_tune.opts; return SNP_OK;
        case SNP_OPT_COMPRESS_LANE_PROBES: *out = c->lane_tune.probes; return SNP_OK;
        case SNP_OPT_COMPRESS_LANES_PER_WAVEFRONT: *out = c->lane_tune.lanes_per_wave; return SNP_OK;
        case SNP_OPT_COMPRESS_SLICE: *out = c->slice_fragments; return SNP_OK;
        case SNP_OPT_COMPRESS_SMALL_INPUT_LDS: *out = c->lane_tune.small_bytes; return SNP_OK;
        case SNP_OPT_COMPRESS_SMALL_INPUT_LANES: *out = c->lane_tune.small_lanes; return SNP_OK;
        case SNP_OPT_FRAME_SCAN: *out = c->frame_scan; return SNP_OK;
        case SNP_OPT_DECODE_LDS_THROTTLE: *out = c->dec_lds; return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_GLOBAL_SLOTS: *out = c->win_gslots; return SNP_OK;
        case SNP_OPT_COMPRESS_WINDOW_DUAL_MIN_BATCH: *out = c->win_dual_min; return SNP_OK;
        default: return SNP_ERR_BAD_ARG;
    }
}

void snp_ctx_destroy(snp_ctx* c)
{
    if (!c) return;
    {
        DevGuard dg(c);
        (void)hipStreamSynchronize(c->stream);
        for (DevBuf* b : {&c->in, &c->out, &c->meta, &c->work, &c->fragtab, &c->scan, &c->small, &c->redo, &c->win_tables, &c->own_tables})
            if (b->p) (void)hipFree(b->p);
        for (void* q : c->kept) (void)hipFree(q);
        if (c->pool) {                                    // the device's last context takes the table pool with it
            std::lock_guard<std::mutex> g(c->pool->mu);
            if (--c->pool->users == 0) c->pool->destroy();
        }
        if (c->order_ev) (void)hipEventDestroy(c->order_ev);
        if (c->hint_ev) (void)hipEventDestroy(c->hint_ev);
        if (c->chint_ev) (void)hipEventDestroy(c->chint_ev);
        if (c->chint) (void)hipHostFree(c->chint);
        if (c->hint) (void)hipHostFree(c->hint);
        if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
        if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
        for (auto& e : c->side_ev) if (e) (void)hipEventDestroy(e);
        for (auto& e : c->copy_ev) if (e) (void)hipEventDestroy(e);
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

snp_status snp_ctx_set_stream(snp_ctx* c, void* stream)
{
    if (!c) return SNP_ERR_BAD_ARG;
    DevGuard dg(c);
    if (!dg.ok) return SNP_ERR_DEVICE;
    hipStream_t next = static_cast<hipStream_t>(stream);
    if (c->own_stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
        c->own_stream = false;
    } else if (!c->rebind(next)) {
        return SNP_ERR_DEVICE;
    }
    c->stream = next;
    return SNP_OK;
}

const char* snp_ctx_last_error(const snp_ctx* c) { return c ? c->err.c_str() : "null context"; }

snp_status snp_ctx_synchronize(snp_ctx* c)
{
    if (!c) return SNP_ERR_BAD_ARG;
    return c->check(hipStreamSynchronize(c->stream), "hipStreamSynchronize") ? SNP_OK : SNP_ERR_DEVICE;
}

const char* snp_status_string(int s)
{
    switch (s) {
        case SNP_OK: return "ok";
        case SNP_ERR_OUTPUT_TOO_SMALL: return "Output buffer is too small.";
        case SNP_ERR_BAD_OFFSET: return "Invalid copy offset";
        case SNP_ERR_TOO_LONG: return "Data too long";
        case SNP_ERR_INCOMPLETE: return "Incomplete Snappy block.";
        case SNP_ERR_BAD_LENGTH: return "Invalid stream length";
        case SNP_ERR_CRC_MISMATCH: return "Chunk CRC mismatch.";
        case SNP_ERR_CHUNK_TYPE: return "Unknown chunk type";
        case SNP_ERR_OVERLAP: return "Input and output spans must not overlap.";
        case SNP_ERR_BAD_ARG: return "bad argument";
        case SNP_ERR_DEVICE: return "HIP device error";
        case SNP_ERR_TRUNCATED_STREAM: return "truncated framed stream";
        default: return "unknown status";
    }
}

const char* snp_version(void) { return "snappier_hip 0.1 (gfx950)"; }

// ---- host-only arithmetic --------------------------------------------------------------------------------

int64_t snp_max_fragment_compressed_length(int64_t n)   // Helpers.MaxCompressedLength  Helpers.cs:17-46
{
    if (n < 0) return -1;
    return 32 + n + n / 6 + 1;
}

int64_t snp_max_compressed_length(int64_t n)            // Snappy.GetMaxCompressedLength  Snappy.cs:20-24
{
    if (n < 0) return -1;
    const int64_t v = snp_max_fragment_compressed_length(n) + SNP_VARINT_MAX;
    return v > 0x7fffffffLL ? -1 : v;
}

// VarIntEncoding.TryReadSlow  VarIntEncoding.Read.cs:38-79 ; anything but Done is "Invalid stream length" (:16-24)
snp_status snp_get_uncompressed_length(const uint8_t* in, size_t n, uint32_t* out_len, uint32_t* out_header_bytes)
{
    if (!in && n) return SNP_ERR_BAD_ARG;
    u32 result = 0;
    int shift = 0;
    for (size_t i = 0; i < n; ++i) {
        const u8 c = in[i];
        const u32 val = c & 0x7fu;
        if (val & ~(0xffffffffu >> shift)) return SNP_ERR_BAD_LENGTH;
        result |= val << shift;
        shift += 7;
        if (c < 128) {
            if (out_len) *out_len = result;
            if (out_header_bytes) *out_header_bytes = static_cast<u32>(i + 1);
            return SNP_OK;
        }
        if (shift >= 32) return SNP_ERR_BAD_LENGTH;
    }
    return SNP_ERR_BAD_LENGTH;
}

int64_t snp_frame_max_encoded_length(int64_t n)
{
    if (n < 0) return -1;
    const int64_t chunks = (n + SNP_BLOCK_SIZE - 1) / SNP_BLOCK_SIZE;
    return SNP_STREAM_HEADER_LEN + chunks * SNP_CHUNK_HEADER_LEN + n;   // a chunk never grows (type 0x01 fallback)
}

}  // extern "C"
