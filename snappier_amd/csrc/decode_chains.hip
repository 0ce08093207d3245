// decode_chains.hip -- k_decode_chains: the default block decoder, one Snappy block per wavefront (gfx950).
//
// Replaces the tag loop of SnappyDecompressor.DecompressAllTags + Append / AppendFromSelf
// (Snappier/Internal/SnappyDecompressor.cs:184-347,568-611; copy semantics CopyHelpers.cs:222-230) for whole blocks.
//
// The kernel is bound by a wavefront's own chain of latencies: what shapes the code is which waits sit on that chain.
//
//   PARSE, one SUPER-WINDOW of 64 x 32 = 2 KiB of compressed input at a time.  Where do the tags start?  Every lane walks a chain of tags
//   through its own 32-byte region, starting blindly at the region's first byte.  A chain that starts inside a tag reads garbage, but a
//   garbage chain and the true chain that land on the same byte are the same chain from there on, and they meet within a few tags:
//     S   the input is staged into LDS not as bytes but as a table of tag ADVANCES -- tab[p] = bytes from a tag that starts at p to the
//         next tag (2 / 3 / 5 for copies, 2..61 for literals, 62..65 = a literal with length bytes) -- computed four bytes per instruction
//         on the dwords as they arrive from global memory; the walks below are then one LDS byte read per tag and no decode;
//     A   lane k walks region k from its first byte, marks what it visits (bit 7 of the table entry; its own 32-bit mask V_k);
//     A'  it walks on past the region's end until it lands on a marked entry (the chains have merged: m_k, next lane nx_k), recording
//         these overrun positions in a per-lane bitmap (kCap = 128 bytes far at most);
//     R   lane 0's chain is the true one (the super-window starts at a tag): the lanes reachable from lane 0 along nx are the lanes
//         whose chains are true from their entry on (pointer doubling over the lanes);
//     T   true tag starts = each such lane's V_k from its entry on, plus its overrun positions: a 2048-bit map whose popcount prefix
//         numbers the tags; the positions are written out as a u16 list (over the table).
//
//   EXECUTE, 64 tags at a time, one per lane, TWO BATCHES IN FLIGHT.  A batch's bytes are assembled in a 2 KiB LDS stage (+ 64 bytes of
//   history below it) and leave as one coalesced write; the batch in the stage is HELD while the next one starts:
//     top     the new batch: tag bytes (a 4-byte gather requested a batch ahead) -> decode -> DPP prefix sum of the output lengths -> where
//             does each tag's source lie?  literals and FAR copies (source ends at or below the held batch's first byte: global memory):
//             their 16-byte pieces are requested now;
//     finish  the HELD batch's waiting tags (source inside their own batch, pattern copies), in order, whole wave per tag, a byte per
//             lane (lanes beyond the tag's length repeat its last byte: no EXEC mask) -- the new batch's round trip to memory runs underneath;
//     write   the new batch's pieces are waited for (they are back), then the held batch leaves: stage -> global memory, 16 bytes per lane;
//     stage   NEAR copies of the new batch (source inside the held batch or the history) read their pieces from the stage; the last 64
//             bytes of what the stage holds move below its first byte (the new batch's history); every piece is stored; the new batch
//             is now the held one.
//   A literal of 65..128 bytes takes TWO slots of a batch (its first 64 bytes, the rest).  Nothing is held across a super-window build (it uses
//   the stage as scratch), a literal > 128 bytes (copied by the whole wave) or an exit.
//   A batch ends before the first tag it cannot take (malformed, a literal > 128 bytes, the last 16 output bytes); anything irregular falls
//   to serial_tail(), which owns the reference's error semantics.
#include "decode_common.h"

// Phase markers for scripts/isa_budget.py (comments in the assembly: no instructions, no barriers beyond `volatile`).
#define SNP_MARK(name) asm volatile("; MARK " #name)

#ifndef SNP_DC_FINISH_MIN
#define SNP_DC_FINISH_MIN 1      // the in-order finish without an EXEC mask (round 6: html 9.25 -> 9.02 ms per 10 GiB, same-process A/B, profiles/r06a_ab_decode.jsonl); 0 = the round-5 loop
#endif

namespace {

constexpr u32 kR = 32;                  // input bytes per lane region
constexpr u32 kW = SNP_WAVE * kR;       // the super-window
constexpr u32 kCap = 128;               // a chain may overrun its region by this much before the wave takes over (multiple of 32)
constexpr u32 kModeBatch = 0, kModeWindow = 1, kModeLongLiteral = 2, kModeLeave = 3;
constexpr u32 kHist = 64;                // the stage keeps the output bytes just before the batch it holds: a copy of <= 64 bytes that starts there finds its source in LDS
constexpr u32 kStage = 2048;            // output bytes of a batch (64 tags of <= 64 bytes could span 4096: a batch is cut at this)

// Advance of the tag that starts with byte c, for the four bytes of a dword at once (Constants.cs:42-76: tag byte, 0..4 trailer bytes,
// the body of a literal): copy-1 / 2 / 4 -> 2 / 3 / 5; literal of h + 1 <= 60 bytes -> h + 2; literal with 1..4 length bytes -> 62..65.
__device__ __forceinline__ u32 adv4(u32 x)
{
    const u32 k1 = 0x01010101u;
    const u32 t = x & 0x03030303u;
    const u32 t1 = t >> 1;
    const u32 hp = ((x >> 2) & 0x3f3f3f3fu) + 0x02020202u;             // literals: h + 2 (<= 65: no carry between the bytes)
    const u32 ca = t + k1 + (t1 & t & k1);                              // copies: 1 -> 2, 2 -> 3, 3 -> 5
    const u32 nz = ((t | t1) & k1) * 255u;                              // 0xff in the bytes whose tag is a copy
    return (ca & nz) | (hp & ~nz);
}

__device__ __forceinline__ u32x4 adv16(u32x4 v) { return u32x4{adv4(v.x), adv4(v.y), adv4(v.z), adv4(v.w)}; }

// The advance of a literal with length bytes (table value a = 62..65) at input position `at` (read from global memory: rare).
__device__ __forceinline__ u32 long_literal_advance(const u8* at, u32 a)
{
    const u32 ex = a - 61u;                                             // 1..4 length bytes
    const u32 b1234 = ld32u(at + 1);
    const u32 tr = ex >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * ex);
    return 2u + ex + min(tr, 0x3fffffffu);                              // saturates so that positions stay below 2^31
}

// Stores into the stage are exact; LOADS are not: a lane reads whole 16-byte pieces, so one memory round trip serves every size class and
// the 8/4/2/1-byte stores of a short tag are cut out of the registers.  The caller guarantees that reading up to 15 bytes past a source is
// safe.  Every vector-memory instruction costs the texture path ~40 cycles whatever its lane count: none is issued needlessly.
// The first 16-byte piece p0 of a tag of len bytes into the stage at d: whole when len >= 16, else as exact 8 / 4 / 2 / 1-byte pieces, each
// cut from the front of what is left of the 16 bytes.
__device__ __forceinline__ void store_first_piece(u8* d, u32x4 p0, u32 len)
{
    if (len >= 16u) {
        st128u(d, p0);
    } else {
        const bool c8 = (len & 8u) != 0, c4 = (len & 4u) != 0, c2 = (len & 2u) != 0;
        const u32 a0 = c8 ? p0.z : p0.x;
        const u32 a1 = c8 ? p0.w : p0.y;
        const u32 b0 = c4 ? a1 : a0;
        const u32 c0 = c2 ? b0 >> 16 : b0;
        if (c8) st64u(d, p0.x, p0.y);
        if (c4) st32u(d + (len & 8u), a0);
        if (c2) st16u(d + (len & 12u), b0);
        if (len & 1u) d[len & 14u] = static_cast<u8>(c0);
    }
}

// One lane copies len (1..64) bytes into the stage at d in 16-byte pieces: first and last 16 bytes (every tag of <= 32 bytes), then the two middle
// pieces of a longer one.  The LOADS (from global memory, or -- near copies -- from the stage while it still holds the previous batch) and the
// STORES are apart: a batch's loads are in flight while the batch before it is finished.  `any_mid`: some lane of the wave has len > 32
// (wave-uniform: the middle pieces are skipped by a scalar branch otherwise).
__device__ __forceinline__ void load_pieces(const u8* s, u32 len, bool any_mid, u32x4& p0, u32x4& p1, u32x4& p2, u32x4& p3)
{
    p0 = ld128u(s);
    if (len > 16u) p1 = ld128u(s + len - 16);
    if (any_mid) {
        if (len > 32u) {
            p2 = ld128u(s + 16);
            p3 = ld128u(s + min(32u, len - 16u));
        }
    }
}

__device__ __forceinline__ void store_pieces(u8* d, u32 len, bool any_mid, const u32x4& p0, const u32x4& p1, const u32x4& p2, const u32x4& p3)
{
    store_first_piece(d, p0, len);
    if (len > 16u) st128u(d + len - 16, p1);
    if (any_mid) {
        if (len > 32u) {
            st128u(d + 16, p2);
            if (len > 48u) st128u(d + 32, p3);
        }
    }
}

// FRAG: one 64 KiB fragment of a larger block (decode_common.h): tags that end at or before the fragment's start (`dead`) are parsed, never
// produced; a tag that straddles the start, or a copy that reaches back before it, is not taken (serial_tail answers kIrregular).
template <bool FENCED, bool FRAG>
__device__ __forceinline__ void chains_front(DecBlk& B, const u32 lane)
{
    __shared__ __attribute__((aligned(16))) u8 c_tab[kW + 64];          // the advance table (+ 64 sentinels); afterwards the tag positions (u16 each, <= kW / 2 of them)
    __shared__ __attribute__((aligned(16))) u8 c_stage_h[kHist + kStage + 64];    // a batch's output; while a super-window is built: flags, entries, overrun bitmaps
    __shared__ u64 c_busy[65];                                          // batches: pending output bytes; while a super-window is built: the tag-start map
    u8* const c_stage = c_stage_h + kHist;                              // (the batch's first byte; up to h_hist bytes of history below it)
    u32* const c_T = reinterpret_cast<u32*>(c_busy);
    u16* const c_pos = reinterpret_cast<u16*>(c_tab);
    const u8* const src = B.src;
    u8* const dst = B.dst;
    const u32 n = B.n, expected = B.expected;
    const u32 skip = FRAG ? B.skip : 0u;
    u32 op = B.op;
    const u32 r0 = kR * lane;
    u32 wbase = B.ip, ntok = 0, emitted = 0, consumed = 0;
    u32 q_pf = 0;                                                       // tag bytes of the batch that starts at list index pf_at,
    u32 pf_at = ~0u;                                                    // requested while the batch before it executes
    // The batch held back: its bytes are in the stage (every piece stored), its waiting tags and its write-out run one trip later, under the next
    // batch's loads.
    bool h_valid = false;
    u64 h_pend = 0;
    u32 h_pk = 0, h_op = 0, h_span = 0, h_hist = 0;             // h_hist: valid history bytes below the held batch in the stage
    // The held batch's waiting tags, in order (`fence`: no vmcnt(0) was waited for on the way here, and the slow form reads global memory).
    auto finish_held = [&](const bool fence) __attribute__((always_inline)) {
        lanes_sync_lds();
        if (h_pend) {
            if (FENCED && fence) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the slow form reads global memory)
            const u32 vbase = lane + static_cast<u32>(reinterpret_cast<uintptr_t>(c_stage));   // LDS address of this lane's byte of a tag at stage offset 0
            const u32 vbase0 = vbase - lane;
            (void)vbase0;
            u64 pend = h_pend;
            while (pend) {
                u32 f, k;
                // The plain form, hand-laid: 16 instructions per tag since round 6 (18 with the EXEC mask of round 5; the compiler's structured
                // version of the same loop: 25).  Pops tags off `pend` until it is empty or the popped tag (f, k < 0) needs the slow form below.
                // The LDS operations of a wavefront execute in order, so a tag reads what the tag before it wrote.
#if SNP_DC_FINISH_MIN
                // No EXEC mask: lanes at or beyond the tag's length repeat the copy of its LAST byte (same source, same destination, same value):
                // v_min replaces v_cmp + s_and_saveexec + s_mov exec -- 16 instructions per tag, 9 of them scalar.  (k carries len - 1 here.)
                {
                    u32 t0, t1, t2, va, vb, vi;
                    asm volatile(
                        "1:\n\t"
                        "s_ff1_i32_b64 %[f], %[pend]\n\t"
                        "v_readlane_b32 %[k], %[pk], %[f]\n\t"
                        "s_bitset0_b64 %[pend], %[f]\n\t"
                        "s_cmp_lt_i32 %[k], 0\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_bfe_u32 %[t0], %[k], 0x7000b\n\t"
                        "s_lshr_b32 %[t1], %[k], 18\n\t"
                        "s_and_b32 %[t2], %[k], 0x7ff\n\t"
                        "v_min_u32_e32 %[vi], %[t0], %[lane]\n\t"
                        "v_add3_u32 %[va], %[vi], %[t1], %[vsrc]\n\t"
                        "ds_read_u8 %[vb], %[va]\n\t"
                        "v_add3_u32 %[va], %[vi], %[t2], %[vbase]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "ds_write_b8 %[va], %[vb]\n\t"
                        "s_cmp_lg_u64 %[pend], 0\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "2:"
                        : [pend] "+s"(pend), [f] "=&s"(f), [k] "=&s"(k), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2),
                          [va] "=&v"(va), [vb] "=&v"(vb), [vi] "=&v"(vi)
                        : [pk] "v"(h_pk), [lane] "v"(lane), [vbase] "v"(vbase0), [vsrc] "v"(vbase0 - kHist)
                        : "scc", "memory");
#else
                {
                    u32 t0, t1, t2, va, vb;
                    u64 sv;
                    asm volatile(
                        "1:\n\t"
                        "s_ff1_i32_b64 %[f], %[pend]\n\t"
                        "v_readlane_b32 %[k], %[pk], %[f]\n\t"
                        "s_bitset0_b64 %[pend], %[f]\n\t"
                        "s_cmp_lt_i32 %[k], 0\n\t"
                        "s_cbranch_scc1 2f\n\t"
                        "s_bfe_u32 %[t0], %[k], 0x7000b\n\t"
                        "s_lshr_b32 %[t1], %[k], 18\n\t"
                        "s_and_b32 %[t2], %[k], 0x7ff\n\t"
                        "v_cmp_gt_u32_e32 vcc, %[t0], %[lane]\n\t"
                        "s_and_saveexec_b64 %[sv], vcc\n\t"
                        "v_add_u32_e32 %[va], %[t1], %[vsrc]\n\t"
                        "ds_read_u8 %[vb], %[va]\n\t"
                        "v_add_u32_e32 %[va], %[t2], %[vbase]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "ds_write_b8 %[va], %[vb]\n\t"
                        "s_mov_b64 exec, %[sv]\n\t"
                        "s_cmp_lg_u64 %[pend], 0\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "2:"
                        : [pend] "+s"(pend), [f] "=&s"(f), [k] "=&s"(k), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [sv] "=&s"(sv),
                          [va] "=&v"(va), [vb] "=&v"(vb)
                        : [pk] "v"(h_pk), [lane] "v"(lane), [vbase] "v"(vbase), [vsrc] "v"(vbase - kHist)
                        : "vcc", "scc", "memory");
#endif
                    if (static_cast<i32>(k) >= 0) break;            // (pend is empty)
                }
                // the slow form: a pattern copy (off < len: CopyHelpers.cs:222-230 copies byte by byte), or a source that starts below the batch
                const u32 f_d = k & 0x7ffu, f_len = ((k >> 11) & 0x7fu) + (SNP_DC_FINISH_MIN ? 1u : 0u);
                const u32 f_off = (k >> 18) & 0x1fffu;
                const u32 sidx = f_off < f_len ? lane_mod(lane, f_off) : lane;
                const u32 spos = f_d + sidx - f_off;                // from the batch's first byte; wraps when below it
                // the source: the stage (the batch, or the history below it) -- or, when the history is shorter than the reach (a batch of
                // < 64 bytes before this one: rare), global memory, on a scalar branch of its own so that only THAT path waits on vmcnt
                u32 byte = 0;
                if (__builtin_expect(static_cast<i32>(f_d - f_off) >= -static_cast<i32>(h_hist), 1)) {
                    if (lane < f_len) byte = c_stage[static_cast<i32>(spos)];
                } else {
                    const bool below = static_cast<i32>(spos) < -static_cast<i32>(h_hist);
                    if ((lane < f_len) & !below) byte = c_stage[static_cast<i32>(spos)];
                    if ((lane < f_len) & below) byte = dst[h_op + spos];
                    asm volatile("" : "+v"(byte));                      // (the vmcnt wait stays in this block: left pending, the compiler would
                }                                                       //  wait at the loop's head, in every trip -- with the next batch's loads in flight)
                if (lane < f_len) c_stage[f_d + lane] = static_cast<u8>(byte);
                lanes_sync_lds();
            }
        }
    };
    auto write_out_held = [&]() __attribute__((always_inline)) {
        // the whole run, coalesced, in 16-byte units: the last one may carry up to 15 stale bytes past the run -- every batch ends at least
        // 16 bytes short of the block's end (`ok`), and what follows (the next batch, the serial loop) stores over them in order
        lanes_sync_lds();
        u8* const g = dst + h_op;
        if (FRAG && h_op < skip) {                                  // the batch the fragment starts in: nothing before its first byte is written
            for (u32 i = skip - h_op + lane; i < h_span; i += SNP_WAVE) g[i] = c_stage[i];
        } else {
            for (u32 i = lane * 16; i < h_span; i += SNP_WAVE * 16)
                *reinterpret_cast<snp_u128_unaligned*>(g + i) = *reinterpret_cast<const snp_u128_unaligned*>(c_stage + i);
        }
    };
    for (;;) {
        // What this trip does: builds the next super-window (nothing may be held then), or takes the next batch of the list.
        u32 mode = emitted == ntok ? kModeWindow : kModeBatch;
        // the new batch (kModeBatch): what outlives the held batch's finish
        u32 len = 0, off = 0, drel = 0, ne = 0, incl = 0, body = 0, s_first = 0, s_len0 = 0, s_body0 = 0, taken = 0;
        bool is_lit = false, live = true;
        if (mode == kModeBatch) {
SNP_MARK(B_top);
            // ---- one batch: the next <= 64 tags of the list ----
            const u32 t = emitted + lane;
            bool have = t < ntok;
            const u32 pos = c_pos[have ? t : emitted];                  // (idle lanes re-read the batch's first position)
            // The tag bytes were requested a batch ago (q_pf); only the first batch of a super-window loads them here, and that load's wait
            // stays on ITS path: merged at a join, the compiler would drain vmcnt in EVERY batch.
            // (FOUR bytes per tag: a dword gather costs the texture path half of what the 8-byte one did, and only a copy-4 or a literal
            //  with four length bytes -- which no 64 KiB-fragment compressor emits -- has a fifth byte: fetched below, when one shows up)
            u32 q = q_pf;
            if (pf_at != emitted) {
                q = ld32u(src + wbase + pos);
                asm volatile("" : "+v"(q));
            }
            const u32 c = q & 0xffu;
            const u32 type = c & 3u;
            const u32 hi6 = c >> 2;
            u32 b1234 = q >> 8;
            if (__builtin_expect(ballot64(have & ((type == 3u) | (c == 0xfcu))) != 0ull, 0)) {
                if ((type == 3u) | (c == 0xfcu)) b1234 |= static_cast<u32>(src[wbase + pos + 4u]) << 24;
            }
            is_lit = type == 0;
            const bool long_lit = is_lit && hi6 >= 60;
            const u32 extra = is_lit ? (long_lit ? hi6 - 59 : 0u) : (type == 3 ? 4u : type);
            const u32 trailer = extra >= 4 ? b1234 : __builtin_amdgcn_ubfe(b1234, 0u, 8 * extra);
            len = (long_lit ? trailer : (hi6 & (type == 1 ? 7u : 63u))) + (type == 1 ? 4u : 1u);
            off = is_lit ? 0u : (type == 1 ? (((c >> 5) << 8) | (b1234 & 0xffu)) : trailer);
            body = pos + 1u + extra;                          // a literal's bytes, from wbase
            // A literal of 65..128 bytes takes TWO slots of the batch -- its first 64 bytes, the rest -- instead of ending it (plain text and
            // low-entropy data are full of them, and each one used to cost a drain of the held batch): the tags are dealt out to the lanes
            // again through the LDS words that are idle between super-window builds.  `last`: the slot is the last (or only) one of its tag.
            u64 last = ~0ull;
            u32 len_tag0 = 0;
            bool dealt = false;                                         // (wave-uniform: the slots were dealt out again)
            if (!FRAG && __builtin_expect(ballot64((q & 0xc0ffu) == 0x40f0u) != 0ull, 0)) {   // (tag 0xf0, length byte 64..127: every compressor's form of such a literal;
                                                                                           //  idle lanes repeat the batch's first tag: no need to mask them)
                const bool two = have && is_lit && len > 64u && len <= 128u;
                dealt = true;
                len_tag0 = read_lane(len, 0);
                const u32 cnt = have ? (two ? 2u : 1u) : 0u;
                const u32 sincl = wave_inclusive_scan(cnt);
                const u32 s0 = sincl - cnt;
                const bool fits = have && sincl <= SNP_WAVE;               // (a prefix of the lanes: the tags this batch's 64 slots hold)
                const u32 nslots = read_lane(sincl, static_cast<u32>(__builtin_popcountll(ballot64(fits))) - 1u);
                // slot word: len (8; a literal > 128 bytes keeps 255: it ends the batch as before) | last (1) | literal (1) | body (16) | offset (32)
                if (fits) c_busy[s0] = (two ? 64u : min(len, 255u)) | (two ? 0u : 0x100u) | (is_lit ? 0x200u : 0u) | (static_cast<u64>(body) << 10) | (static_cast<u64>(off) << 26);
                if (fits && two) c_busy[s0 + 1u] = (len - 64u) | 0x100u | 0x200u | (static_cast<u64>(body + 64u) << 10);
                lanes_sync_lds();
                const u64 w = c_busy[lane];
                lanes_sync_lds();
                have = lane < nslots;
                len = static_cast<u32>(w) & 0xffu;
                is_lit = (w & 0x200u) != 0;
                body = static_cast<u32>(w >> 10) & 0xffffu;
                off = static_cast<u32>(w >> 26);
                last = ballot64(have && (w & 0x100u) != 0) | ~ballot64(have);
            }
            const u32 olen = have ? len : 0u;
            incl = wave_inclusive_scan(olen);
            drel = incl - olen;                                         // the tag's first output byte, from the batch's
            const u32 room = n - wbase - 16u;                           // load_pieces over-reads 15 bytes
            const bool lit_ok = ((len - 1u) < room) & (body <= room - len);
            const bool dead = FRAG && op + incl <= skip;                // ends at or before the fragment's start: parsed only
            live = !FRAG || op + drel >= skip;
            const bool copy_ok = (off - 1u) < op + drel - skip;
            const bool ok = have & (dead ? (!is_lit | lit_ok) : live & ((is_lit & lit_ok) | (!is_lit & copy_ok))) & (incl + 16u <= expected - op);
            const bool big = is_lit & (len > 64u);
            const u64 okm = ballot64(ok & !big & (incl <= kStage));
            ne = okm == ~0ull ? 64u : static_cast<u32>(__builtin_ctzll(~okm));
            taken = ne;                                                 // tags, not slots:
            if (dealt) {
                if (ne && !((last >> (ne - 1u)) & 1ull)) ne -= 1u;      // never between the two slots of a literal
                taken = static_cast<u32>(__builtin_popcountll(last & (ne == 64u ? ~0ull : (1ull << ne) - 1ull)));
            }
SNP_MARK(B_ne0);
            if (ne == 0) {
                // not a batch: a literal > 64 bytes goes by the whole wave, anything else to the serial loop -- after the held batch
                // (slots dealt out again: slot 0 may be the first half of a literal whose second half cannot be taken, or carry a clamped
                //  length -- the WHOLE first tag is judged then, by the checks `ok` makes)
                const u32 body0 = read_lane(body, 0);
                const bool long0 = dealt && read_lane(is_lit ? 1u : 0u, 0) != 0u && len_tag0 > 64u;
                const bool whole_ok = (len_tag0 - 1u) < room && body0 <= room - len_tag0 && static_cast<u64>(len_tag0) + 16u <= expected - op;
                const u32 f0 = long0 ? (whole_ok ? 3u : 0u) : read_lane((ok ? 1u : 0u) | (big ? 2u : 0u), 0);
                mode = f0 == 3u ? kModeLongLiteral : kModeLeave;
                s_first = read_lane(pos, 0);
                s_len0 = dealt ? len_tag0 : read_lane(len, 0);         // (slot 0 is tag 0: its full length, whatever the slots say)
                s_body0 = read_lane(body, 0) | (FRAG && read_lane(dead ? 1u : 0u, 0) ? 0x80000000u : 0u);
            }
        }
        if (mode != kModeBatch) {
            // Nothing is held across a window build (it uses the stage), a long literal or an exit.
            if (h_valid) {
                finish_held(true);
                write_out_held();
                h_valid = false;
            }
            if (mode == kModeLeave) {                                   // not ours: the serial loop decides, from this tag on
                wbase += s_first;
                consumed = 0;
                break;
            }
            if (mode == kModeLongLiteral) {
                if (!FRAG || static_cast<i32>(s_body0) >= 0) wave_copy(dst + op, src + wbase + (s_body0 & 0x7fffffffu), s_len0, lane);
                op += s_len0;
                emitted += 1;
                continue;
            }
            // ---- the next super-window ----
SNP_MARK(S_stage_table);
            wbase += consumed;
            consumed = 0;
            if (wbase + 72 > n || op >= expected) break;
            const u32 avail = n - wbase;
            const u32 L = min(kW, avail) - 8u;                          // tags may start below L: their 8 bytes lie inside the input
            const u8* const wsrc = src + wbase;
            {
                // S: 2 x 16 bytes per lane; a piece that would cross the end of the input is pulled back inside it (avail >= 72; the entries
                // it rewrites are the same entries), pieces beyond it are not needed
                const u32 oa = lane * 16u, ob = oa + 1024u;
                const u32 la = min(oa, avail - 16u), lb = min(ob, avail - 16u);
                const u32x4 va = ld128u(wsrc + la), vb = ld128u(wsrc + lb);
                st128u(c_tab + la, adv16(va));
                st128u(c_tab + lb, adv16(vb));
                lanes_sync_lds();
                c_tab[L + lane] = 64;                                   // sentinels: a chain that gets here has left the window (entries < 62 + 64 bytes on)
            }
            lanes_sync_lds();
SNP_MARK(A_walk);
            // A: the chain from the first byte of the lane's region.  A literal with length bytes (a >= 62) ends the walk whatever its
            // real length -- p + 62 is past the region -- and is sorted out below, off the loop.
            u32 p = r0, V = 0, a = 0;
            const u32 lim = min(r0 + kR, L);
            while (p < lim) {
                a = c_tab[p];
                c_tab[p] = static_cast<u8>(a | 0x80u);
                V |= 1u << (p & 31u);
                p += a;
            }
SNP_MARK(Aprime_walk);
            c_T[lane] = 0;
            u32* const c_O = reinterpret_cast<u32*>(c_stage + 512) + lane * (kCap / 32);   // overrun positions, a bit each, from obase
#pragma unroll
            for (u32 w = 0; w < kCap / 32; ++w) c_O[w] = 0;
            lanes_sync_lds();
            // A': on past the region until the chain lands on a marked entry, on a literal with length bytes, kCap bytes away, or on
            // the sentinels past the window.  A literal with length bytes is where a chain STOPS in both walks (its length bytes are
            // not in the table): if it lies on the true path, the wave follows it below (`unresolved`).  One loop exit, at the top.
            const bool stopA = (p > r0) & (a >= 62u);                   // the region walk ended on one: the chain stands there, marked
            p = stopA ? p - a : min(p, L + 63u);                        // (an idle region past the input: onto a sentinel -- what lies beyond them is stale)
            const u32 obase = p & ~(kR - 1u);
            u32* const c_Ob = c_O - (obase >> 5);                       // word of position q: c_Ob[q >> 5]
            a = c_tab[p];
            u32 x = a | ((p - obase) & ~(kCap - 1u));                   // >= 62: stop (kCap is a power of two)
            while (x < 62u) {
                atomicOr(&c_Ob[p >> 5], 1u << (p & 31u));
                p += a;
                a = c_tab[p];
                x = a | ((p - obase) & ~(kCap - 1u));
            }
            // where the chain merged (nx = that lane), left the super-window (64), or stopped unresolved (65)
            const u32 nx = stopA ? 65u : (a & 0x80u) ? p >> 5 : (p >= L ? 64u : 65u);
SNP_MARK(R_reach);
            const u32 m = p;                                            // where the chain merged, gave up or left
            // R: the lanes on the true chain = the lanes reachable from lane 0 along nx, by pointer doubling (flags through LDS: a
            // scatter needs its senders masked); each of them tells its successor where it enters.
            u64 active;
            u32 entry = 0;
            {
                u8* const c_reach = c_stage;                            // (the stage is idle while a super-window is built)
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
                const u64 ends = ballot64(reached && nx == 64u);        // the lane whose chain leaves the super-window, if the chain gets there
                consumed = ends ? read_lane(m, static_cast<u32>(__builtin_ctzll(ends))) : 0u;
            }
            if (ballot64(((active >> lane) & 1ull) && nx == 65u)) {
                // a chain on the true path did not merge within kCap bytes (rare: ~2 per block on html): follow the path lane by lane on the
                // scalar unit instead, walking such a chain on, whole wave, until it merges or leaves (tag bytes from global memory)
                active = 0;
                entry = 0;
                for (u32 k = 0, e = 0;;) {
                    active |= 1ull << k;
                    entry = lane == k ? e : entry;
                    u32 mk = read_lane(m, k), nk = read_lane(nx, k);
                    if (nk == 65u) {
                        nk = 64u;
                        for (bool first = true; mk < L; first = false) {
                            u32 a = bcast_first(static_cast<u32>(c_tab[mk]));
                            if (!first && a >= 128u) { nk = mk >> 5; break; }   // (the entry it stands on may be its own mark)
                            a &= 0x7fu;
                            if (lane == 0) atomicOr(&c_T[mk >> 5], 1u << (mk & 31u));
                            if (a >= 62u) a = bcast_first(long_literal_advance(wsrc + mk, a));
                            mk += a;
                        }
                    }
                    if (nk >= 64u) { consumed = mk; break; }
                    e = mk;
                    k = nk;
                }
            }
SNP_MARK(T_list);
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
            lanes_sync_lds();                                           // (every read of the table is done: the list overwrites it)
            u32 t = cincl - cnt, bits = Tw;
            while (bits) {
                c_pos[t++] = static_cast<u16>(r0 + static_cast<u32>(__builtin_ctz(bits)));
                bits &= bits - 1u;
            }
            lanes_sync_lds();
            emitted = 0;
            pf_at = ~0u;
            if (ntok == 0) {                                            // (cannot happen: position 0 is always a tag start; guards the loop)
                consumed = 0;
                break;
            }
            continue;
    }
SNP_MARK(B_pass1);
        const bool act = lane < ne;
        const u32 span = read_lane(incl, ne - 1);
        // Where a copy's source lies (the held batch's bytes are in the stage, not yet in global memory):
        //   far    ends at or below the held batch's first byte: global memory, loaded NOW -- the round trip runs under the held batch's finish;
        //   near   inside the held batch: read from the stage once that batch is finished, before the stage is overwritten;
        //   else   overlaps its own batch (or straddles the held batch's first byte: rare, the finish's slow form): waits, in order.
        const u32 gap = h_valid ? op - h_op : 0u;
        const u32 reach = drel + len;                           // off >= reach: the source ends at or below this batch's first byte
        const u32 back = gap + (h_valid ? h_hist : 0u);                 // the stage holds this many bytes below this batch's first
        const bool far = !is_lit & (off >= reach + gap);
        const bool near = act & live & !is_lit & !far & (off >= reach) & (off <= drel + back);
        const bool early = act & live & (is_lit | far);
        const bool waits = act & live & !early & !near;
        if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the write-outs before the held batch's are visible to this wave's loads
        pf_at = emitted + taken;                                // the next batch's tag bytes travel with this batch's loads
        if (pf_at < ntok) q_pf = ld32u(src + wbase + c_pos[pf_at + lane < ntok ? pf_at + lane : pf_at]);
        const bool any_mid = ballot64(act & (len > 32u)) != 0ull;
        const u32 sofs_mine = is_lit ? wbase + body : op + drel - off;   // the source, from src (literals) or dst (copies)
        u32x4 p0, p1, p2, p3;                                           // read only where they were loaded
        if (early) load_pieces((is_lit ? src : dst) + sofs_mine, len, any_mid, p0, p1, p2, p3);
SNP_MARK(B_finish);
        if (h_valid) finish_held(false);
SNP_MARK(B_writeout);
        // This batch's loads are back by now (they had the finish to travel in): waited for HERE, before the write-out's stores join the
        // queue behind them -- past this point the compiler sees fresh values and has no reason to drain the stores with them.
        asm volatile("; loads of the new batch are back"
                     :: "v"(p0.x), "v"(p0.y), "v"(p0.z), "v"(p0.w), "v"(p1.x), "v"(p1.y), "v"(p1.z), "v"(p1.w), "v"(p2.x), "v"(p2.y), "v"(p2.z),
                        "v"(p2.w), "v"(p3.x), "v"(p3.y), "v"(p3.z), "v"(p3.w), "v"(q_pf)
                     : "memory");
        if (h_valid) write_out_held();
SNP_MARK(B_stage);
        // ---- the new batch into the stage (the held batch has left it): near copies read their source there first ----
        if (near) load_pieces(c_stage + static_cast<i32>(drel + gap - off), len, any_mid, p0, p1, p2, p3);
        // the history for THIS batch: the last 64 bytes of what the stage holds (history + held batch), moved below the batch's first byte
        const u32 hist = h_valid ? min(kHist, h_hist + gap) : 0u;
        if (hist) {
            const u8 hb = c_stage[static_cast<i32>(gap + lane - kHist)];
            if (lane >= kHist - hist) c_stage[static_cast<i32>(lane - kHist)] = hb;
        }
        // (the near pieces and the history byte above are READ from the stage bytes the stores below overwrite, through differently typed unaligned
        //  accesses: the compiler must not move a store above them -- the hardware keeps a wavefront's DS operations in order: ADVICE r5)
        lanes_sync_lds();
        if (early | near) store_pieces(c_stage + drel, len, any_mid, p0, p1, p2, p3);
        {
            // The rest waits for the next trip, in order, whole wave per tag, a byte per lane.  One packed word per tag (v_readlane): destination,
            // length, and either the source inside the stage or a flag for the slow form (pattern copy, source that starts below the batch).
            const bool plain = (off >= len) & (off <= drel + hist);     // not a pattern copy, source inside the batch or the history below it
            h_pend = ballot64(waits);
            h_pk = drel | ((len - (SNP_DC_FINISH_MIN ? 1u : 0u)) << 11) | (plain ? (drel - off + kHist) << 18 : 0x80000000u | (off << 18));   // (plain: the source, counted from the
                                                                        //  history's first byte; slow: the offset -- a waiting tag's is < drel + len + 64)
            h_op = op;
            h_span = span;
            h_hist = hist;
            h_valid = true;
        }
        op += span;
        emitted += taken;
    }
SNP_MARK(Z_exit);
    B.ip = wbase;                                                       // (every exit leaves consumed = 0)
    B.op = op;
    B.w.wv = 0x80000000u;                                               // the serial loop re-seats its window
}

template <bool FENCED, bool FRAG>
__device__ __forceinline__ void decode_block_chains(SNP_D_PARAMS, const u32 b)
{
    const u32 lane = lane_id();
    DecBlk B;
    if (!block_begin<FRAG>(SNP_D_ARGS, b, lane, B)) return;
    if (B.st == SNP_OK) chains_front<FENCED, FRAG>(B, lane);
    serial_tail<FENCED, FRAG>(B, b, lane, out_len, status);
}

// 64 VGPRs: eight wavefronts per SIMD.
#ifdef SNP_DC_VGPRS
#define SNP_DC_ATTR __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_vgpr(SNP_DC_VGPRS)))
#else
#define SNP_DC_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
template <bool FENCED>
__global__ __launch_bounds__(SNP_WAVE) SNP_DC_ATTR void k_decode_chains(SNP_D_PARAMS)
{
    decode_block_chains<FENCED, false>(SNP_D_ARGS, blockIdx.x);
}

// One wavefront per 64 KiB output fragment of one large block (tag_index.hip found where each begins).
template <bool FENCED>
__global__ __launch_bounds__(SNP_WAVE) SNP_DC_ATTR void k_decode_chains_frag(SNP_D_PARAMS)
{
    decode_block_chains<FENCED, true>(SNP_D_ARGS, blockIdx.x);
}

// The same over a LIST of blocks: the blocks the small-block pre-pass (decompress_small.hip) did not finish, which it appended to
// `list` in 64 sub-lists (ctl[s] = length of sub-list s).  Persistent: the grid is one chip-full of wavefronts and each takes list
// entries by ticket (ctl[64]) until the list is empty -- when the pre-pass finished everything (millions of small blocks) this launch
// costs microseconds instead of one empty workgroup per block, and when it finished nothing the wavefronts decode ~20 blocks each.
template <bool FENCED>
__global__ __launch_bounds__(SNP_WAVE) SNP_DC_ATTR void k_decode_chains_list(
    SNP_D_PARAMS, const u32* __restrict__ list, u32* __restrict__ ctl, u32 sub_cap)
{
    const u32 lane = lane_id();
    const u32 mine = __hip_atomic_load(&ctl[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u32 incl = wave_inclusive_scan(mine);
    const u32 count = read_lane(incl, 63);
    // tickets are taken several at a time (one counter for 2 M small blocks was the whole run time); a wavefront's first tickets are
    // its by position, only the later ones come from the counter
    const u32 grab = min(max(count / (gridDim.x * 8u), 1u), 64u);
    for (u32 first = blockIdx.x * grab; first < count;) {
        const u32 last = min(first + grab, count);
        for (u32 i = first; i < last; ++i) {
            const u32 sub = static_cast<u32>(__builtin_popcountll(ballot64(incl <= i)));   // the sub-list ticket i falls into
            const u32 before = read_lane(incl - mine, sub);
            decode_block_chains<FENCED, false>(SNP_D_ARGS, list[static_cast<u64>(sub) * sub_cap + (i - before)]);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the next block reuses the LDS arrays)
        }
        u32 next = 0;
        if (lane == 0) next = atomicAdd(&ctl[64], grab);
        first = gridDim.x * grab + bcast_first(next);
    }
}

}  // namespace

// lds_bytes: dynamic LDS requested per wavefront purely to cap how many blocks a CU decodes at once (0 = no cap).
extern "C" hipError_t snp_launch_decode_chains(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out, const u64* out_off,
                                               const u32* out_cap, u32* out_len, i32* status, const u8* chunk_type, int fenced, int redo_only,
                                               unsigned lds_bytes, hipStream_t stream, const u32* frag_skip)
{
    if (nblocks == 0) return hipSuccess;
    if (frag_skip) {                                    // fragments of one large block (tag_index.hip)
        if (fenced)
            hipLaunchKernelGGL((k_decode_chains_frag<true>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks, out, out_off,
                               out_cap, out_len, status, nullptr, frag_skip, 0);
        else
            hipLaunchKernelGGL((k_decode_chains_frag<false>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks, out, out_off,
                               out_cap, out_len, status, nullptr, frag_skip, 0);
        return hipGetLastError();
    }
    const u32* const no_skip = nullptr;
    if (fenced)
        hipLaunchKernelGGL((k_decode_chains<true>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks, out, out_off,
                           out_cap, out_len, status, chunk_type, no_skip, redo_only);
    else
        hipLaunchKernelGGL((k_decode_chains<false>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks, out, out_off,
                           out_cap, out_len, status, chunk_type, no_skip, redo_only);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_decode_chains_list(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out, const u64* out_off,
                                                    const u32* out_cap, u32* out_len, i32* status, const u8* chunk_type, int fenced,
                                                    unsigned lds_bytes, hipStream_t stream, const u32* list, u32* ctl, u32 waves, u32 sub_cap)
{
    if (nblocks == 0) return hipSuccess;
    const u32* const no_skip = nullptr;
    if (fenced)
        hipLaunchKernelGGL((k_decode_chains_list<true>), dim3(waves), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks, out, out_off,
                           out_cap, out_len, status, chunk_type, no_skip, 0, list, ctl, sub_cap);
    else
        hipLaunchKernelGGL((k_decode_chains_list<false>), dim3(waves), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len, nblocks, out, out_off,
                           out_cap, out_len, status, chunk_type, no_skip, 0, list, ctl, sub_cap);
    return hipGetLastError();
}
