// decompress_small.hip -- SMALL Snappy blocks (<= 512 declared bytes), several per wavefront (gfx950).  SURVEY.md 8(f4): the "many
// tiny chunks" pattern of Snappier.Tests/SnappyStreamTests.cs:145-192 and batches of sub-page records.
//
// decompress.hip spends a whole wavefront on a block: a 2 KiB super-window, 64-tag execution batches.  A 256-byte block holds
// ~10 tags -- one nearly empty window, one nearly empty batch -- and 4 M of them are 4 M workgroups (126-148 GB/s measured).
// The kernels here are a PRE-PASS over a batch: they finish every clean small block (SnappyDecompressor.DecompressAllTags +
// Append / AppendFromSelf, SnappyDecompressor.cs:184-347, 568-611; copy semantics CopyHelpers.cs:222-230) and leave anything
// else -- a larger block, an uncompressed framing chunk, and every irregularity (bad preamble, offset, length, truncated input)
// -- untouched, marked kRedoStatus and appended to a list; decompress.hip decodes exactly those and owns every error code.
//   k_decompress_small      one block per LANE, global memory: the best layout for blocks of <= ~48 bytes (64 blocks per wavefront)
//   k_decompress_teams<T>   one block per TEAM of T = 4 / 8 / 16 lanes, compressed and decoded bytes in LDS, coalesced I/O
//   k_sample_caps           what a batch that skipped the pre-pass looked like (the host's policy, capi_batch.hip launch_decompress)
// Which one runs, and with how much LDS per wavefront, is decided per batch from the previous batch's read-back (HISTORY.md §4.1b).
//
// k_decompress_small: every lane decodes its own block, and the loop is shaped so that one tag costs ONE dependent memory round trip:
//   * the next tag's bytes are requested as soon as this tag's length is known, before its copy is issued;
//   * copies are 16-byte pieces (stores exact at the block's end, may overshoot inside it: later tags overwrite the excess);
//   * an overlapping copy (offset < length) doubles the written prefix: log2(length / offset) steps instead of a byte loop.
#include "snp_device.h"

namespace {

constexpr i32 kRedoStatus = -1;
constexpr u32 kDefaultTeamLog = 3;                                     // lanes per block of k_decompress_teams: 2^this

// The blocks a pre-pass kernel leaves to decompress.hip go on a list -- 64 sub-lists, each with its own counter, because one
// counter for all wavefronts was the bottleneck (131 k same-address atomics for 1 M blocks: 1.4 of 4.9 ms): wavefront w appends to
// sub-list w mod 64 with one atomic (ctl[s] = its length, entries at list[s * sub_cap ...]).  list == nullptr: count only.
// `mine` = this lane reports block b (one lane per block).  Called by every lane still running, exactly once.
__device__ __forceinline__ void append_redo(bool mine, u32 b, u32* __restrict__ list, u32* __restrict__ ctl, u32 sub_cap)
{
    if (!ctl) return;
    const u64 m = __ballot(mine);
    if (m == 0) return;
    const u32 lane = lane_id();
    const u32 sub = blockIdx.x & 63u;
    u32 base = 0;
    if (lane == static_cast<u32>(__builtin_ctzll(m))) base = atomicAdd(&ctl[sub], static_cast<u32>(__builtin_popcountll(m)));
    base = __builtin_amdgcn_readlane(base, __builtin_ctzll(m));
    if (mine && list) list[static_cast<u64>(sub) * sub_cap + base + static_cast<u32>(__builtin_popcountll(m & lanes_below(lane)))] = b;
}

// Every 64th wavefront of a pre-pass kernel also reports the capacities it saw (ctl[66] += their sum, ctl[67] += how many): the
// host reads them back with the sub-list lengths and sizes the NEXT batch's pre-pass by the mean block size (capi_batch.hip).
__device__ __forceinline__ void sample_sizes(bool live, u32 cap, u32* __restrict__ ctl)
{
    if (!ctl || (blockIdx.x & 63u) != 0) return;
    u32 sum = live ? (cap < 0x100000u ? cap : 0x100000u) : 0u, cnt = live ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sum += static_cast<u32>(__shfl_xor(static_cast<int>(sum), d, SNP_WAVE));
        cnt += static_cast<u32>(__shfl_xor(static_cast<int>(cnt), d, SNP_WAVE));
    }
    if (lane_id() == 0) {
        atomicAdd(&ctl[66], sum >> 4);                                  // (in units of 16 bytes, as k_sample_caps)
        atomicAdd(&ctl[67], cnt);
    }
}

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
                                                              u32 small_max, u32* __restrict__ list, u32* __restrict__ ctl, u32 sub_cap)
{
    const u32 b = blockIdx.x * SNP_WAVE + threadIdx.x;
    if (b >= nblocks) return;                                           // (the last wavefront: append_redo's ballot sees the live lanes only)
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
        // (len - 1 >= ...: a long literal whose 4-byte length is 0xFFFFFFFF makes len wrap to 0 -- the reference sees 2^32 bytes and fails)
        if (len - 1u >= expected - op || (type == 0 ? (len > n - body) : (off == 0 || off > op))) { redo = true; break; }
        // request the next tag now: it does not depend on this tag's copy
        have = nip + 8 <= n;
        u64 qn = 0;
        if (have) qn = ld64u(src + nip);
        u8* d = dst + op;
        if (type == 0) {                                                // literal  :262-302, Append :568-589
            copy_pieces(d, src + body, len, body + len + 16 <= n, op + len + 16 <= expected);
        } else if (off >= len || off >= 16) {                           // pieces never read what they write
            copy_pieces(d, d - off, len, true, op + len + 16 <= expected);   // the source's slop is earlier output of this block; stores stay below `expected`
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
    if (redo) status[b] = kRedoStatus;
    else {
        out_len[b] = op;
        status[b] = SNP_OK;
    }
    append_redo(redo, b, list, ctl, sub_cap);
    sample_sizes(true, cap, ctl);
}


// ---- several blocks per wavefront, a TEAM of 4 / 8 / 16 lanes per block ---------------------------------------------------------
// The lane-per-block kernel above runs at ~200 GB/s whatever the block size: every lane's 8- or 16-byte access is its own
// memory transaction.  Here a block belongs to a team of adjacent lanes: they copy its compressed bytes into LDS with
// coalesced 16-byte loads, run the reference's tag loop together -- every lane of the team decodes the same tag (three
// aligned LDS dwords + v_alignbyte: an LDS access at an odd address is serialised per lane,
// profiles/r02n_microbench_lds_unaligned.jsonl), then literal and copy alike move TEAM bytes per step, one per lane, LDS to
// LDS: out[op + k] = source[k - dist], where dist is the copy's offset (doubled while it is shorter than what the tag has
// produced, so a step never reads what it writes and a run-length pattern needs log steps) -- and write the finished block
// out with coalesced 16-byte stores.  The teams of a wavefront loop independently (exec mask); a tag costs a few LDS round
// trips, not a dependent global one.  As above, only clean blocks are finished; the rest is marked kRedoStatus.
// LDS holds every block in flight (compressed + decoded), so a CU decodes 160 KiB / ~1.2 x block size blocks at a time
// whatever the team size; smaller teams mean fewer wavefronts doing more per instruction.
// LDS is handed out by need, not by the largest block the kernel accepts: a team takes align16(compressed bytes) + align16(its
// capacity) + 32, the teams of a wavefront share kTeamBudget bytes (sized for 32 wavefronts per CU), and teams that do not fit
// in one round take the next one (512-byte blocks: two rounds of 5 + 3; 256-byte blocks and smaller: one).
constexpr u32 kTeamBudget = 4608;                                       // default; the launcher may hand a wavefront more (fewer wavefronts per CU)
constexpr u32 kTeamOutMax = 512;                                        // declared bytes a team accepts at most

#define SNP_T_PARAMS                                                                                                    \
    const u8 *__restrict__ in, const u64 *__restrict__ in_off, const u32 *__restrict__ in_len, u32 nblocks, u8 *out,    \
        const u64 *__restrict__ out_off, const u32 *__restrict__ out_cap, u32 *__restrict__ out_len,                   \
        i32 *__restrict__ status, const u8 *__restrict__ chunk_type, u32 small_max, u32 *__restrict__ list,            \
        u32 *__restrict__ ctl, u32 sub_cap, u32 budget
#define SNP_T_ARGS in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, small_max, list, ctl, sub_cap, budget

template <u32 TEAM>
__device__ __forceinline__ void decompress_teams(SNP_T_PARAMS)
{
    constexpr u32 kTeams = SNP_WAVE / TEAM;
    extern __shared__ __attribute__((aligned(16))) u8 t_buf[];          // `budget` bytes (dynamic LDS)
    const u32 lane = threadIdx.x, tl = lane & (TEAM - 1), team = lane / TEAM;
    const u32 b = blockIdx.x * kTeams + team;
    const bool live = b < nblocks;
    u32 n = 0, cap = 0;
    const u8* src = in;
    u8* dst = out;
    if (live) {
        n = in_len[b];
        cap = out_cap[b];
        src = in + in_off[b];
        dst = out + out_off[b];
    }
    const u32 in_room = (n + 15u) & ~15u;
    const u32 need = in_room + 16u + ((cap + 15u) & ~15u) + 16u;        // [compressed | slack | decoded | slack]
    // (n bounded first: (n + 15) & ~15 wraps for n >= 0xFFFFFFF1 and the block would be admitted with a tiny `need`)
    bool redo = live && (cap > small_max || cap > kTeamOutMax || n < 1 || n > 2 * kTeamOutMax + 64 || need > budget || (chunk_type && chunk_type[b] == 1));
    bool waiting = live && !redo;                                       // teams that have not had their round yet
    u32 op = 0;
    while (__ballot(waiting)) {
        // this round: the waiting teams, in order, as long as they fit
        const u32 ask = (waiting && tl == 0) ? need : 0u;
        u32 incl = ask;                                                 // inclusive prefix over the lanes (only team leaders ask)
#pragma unroll
        for (int d = 1; d < SNP_WAVE; d <<= 1) {
            const u32 up = static_cast<u32>(__shfl_up(static_cast<int>(incl), d, SNP_WAVE));
            incl += lane >= static_cast<u32>(d) ? up : 0u;
        }
        const u32 end = static_cast<u32>(__shfl(static_cast<int>(incl), static_cast<int>(team * TEAM), SNP_WAVE));   // my team's slot ends here
        const bool now = waiting && end <= budget;
        waiting = waiting && !now;
        u8* const tin = t_buf + (end - need);
        const u32 kOut = in_room + 16u;                                 // the decoded bytes start here
        bool bad = false;
        if (now) {                                                      // the block's bytes, 16 per lane per step; the last piece is pulled back inside
            if (n >= 16) {
                for (u32 o = tl * 16; o < n; o += TEAM * 16) {
                    const u32 o2 = o < n - 16 ? o : n - 16;
                    *reinterpret_cast<snp_u128_unaligned*>(tin + o2) = *reinterpret_cast<const snp_u128_unaligned*>(src + o2);
                }
            } else {
                for (u32 o = tl; o < n; o += TEAM) tin[o] = src[o];
            }
        }
        asm volatile("" ::: "memory");                                  // (LDS operations of a wave execute in order)
        u32 ip = 0, expected = 0;
        op = 0;
        if (now) {                                                      // varint preamble  VarIntEncoding.Read.cs:38-79
            u32 shift = 0;
            bool done = false;
            while (ip < n && ip < 5) {
                const u32 c = tin[ip++];
                const u32 val = c & 0x7fu;
                if (val & ~(0xffffffffu >> shift)) break;
                expected |= val << shift;
                shift += 7;
                if (c < 128) { done = true; break; }
            }
            bad = !done || expected > cap;
        }
        const bool run = now && !bad;
        // (the three dwords of the NEXT tag are requested before this tag's bytes are moved: one LDS round trip less per tag)
        u32 d0 = 0, d1 = 0, d2 = 0;
        if (run) {
            const u32 a = ip & ~3u;
            d0 = *reinterpret_cast<const u32*>(tin + a), d1 = *reinterpret_cast<const u32*>(tin + a + 4), d2 = *reinterpret_cast<const u32*>(tin + a + 8);
        }
        while (run && op < expected) {                                  // SnappyDecompressor.DecompressAllTags :234-341, one tag per trip
            if (ip >= n) { bad = true; break; }                         // input ends early: decompress.hip reports it
            const u32 lo = __builtin_amdgcn_alignbyte(d1, d0, ip & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, ip & 3u);
            const u32 c = lo & 0xffu;
            const u32 type = c & 3u;
            const u32 hi6 = c >> 2;
            const u32 extra = type == 0 ? (hi6 >= 60 ? hi6 - 59 : 0u) : (type == 3 ? 4u : type);   // CharTable[c] >> 11  Constants.cs:42-76
            if (n - ip < 1 + extra) { bad = true; break; }
            const u32 b1234 = (lo >> 8) | (hi << 24);
            const u32 trailer = extra >= 4 ? b1234 : (b1234 & ((1u << (8 * extra)) - 1u));
            const u32 body = ip + 1 + extra;
            u32 len, off = 0;
            if (type == 0) len = (hi6 >= 60 ? trailer : hi6) + 1;
            else if (type == 1) { len = (hi6 & 7u) + 4; off = ((c >> 5) << 8) | trailer; }
            else { len = hi6 + 1; off = trailer; }
            // irregular -> leave it all to decompress.hip (TOO_LONG / BAD_OFFSET / partial literal); len = 0 is a 2^32-byte literal
            if (len - 1u >= expected - op || (type == 0 ? (len > n - body) : (off == 0 || off > op))) { bad = true; break; }
            const u32 nip = type == 0 ? body + len : body;
            {
                const u32 a = nip & ~3u;                                // (nip <= n: inside the slot's slack)
                d0 = *reinterpret_cast<const u32*>(tin + a), d1 = *reinterpret_cast<const u32*>(tin + a + 4), d2 = *reinterpret_cast<const u32*>(tin + a + 8);
            }
            // literal (Append :568-589): source = the input;  copy (AppendFromSelf :591-611): source = dist bytes back in the output
            u32 from = type == 0 ? body : kOut + op - off;              // slot offset the tag's first byte comes from
            u32 dist = type == 0 ? 0xffffu : off;
            const u32 to = kOut + op;
            if (dist >= len) {
                // the tag reads nothing it writes (every literal, most copies): its steps are independent -- four in flight at a
                // time, so that a 30-byte literal costs one LDS round trip instead of four
                for (u32 k0 = tl; k0 < len; k0 += 4 * TEAM) {
                    const u32 k1 = k0 + TEAM, k2 = k1 + TEAM, k3 = k2 + TEAM;
                    u8 v0 = tin[from + k0], v1 = 0, v2 = 0, v3 = 0;
                    if (k1 < len) v1 = tin[from + k1];
                    if (k2 < len) v2 = tin[from + k2];
                    if (k3 < len) v3 = tin[from + k3];
                    tin[to + k0] = v0;
                    if (k1 < len) tin[to + k1] = v1;
                    if (k2 < len) tin[to + k2] = v2;
                    if (k3 < len) tin[to + k3] = v3;
                }
            } else {
                for (u32 done = 0; done < len;) {
                    const u32 w = min(min(dist, TEAM), len - done);
                    if (tl < w) tin[to + done + tl] = tin[from + done + tl];
                    done += w;
                    if (dist <= done && dist < TEAM) {                  // a pattern: twice the distance is the same bytes
                        from -= dist;
                        dist *= 2;
                    }
                }
            }
            ip = nip;
            op += len;
        }
        // a clean block ends exactly at `expected`; bytes after the last needed tag: decompress.hip decides (TOO_LONG or ignored)
        if (run && !bad && (op != expected || ip != n)) bad = true;
        redo = redo || (now && bad);
        asm volatile("" ::: "memory");
        if (now && !bad) {
            for (u32 o = tl * 16; o + 16 <= expected; o += TEAM * 16)
                *reinterpret_cast<snp_u128_unaligned*>(dst + o) = *reinterpret_cast<const snp_u128_unaligned*>(tin + kOut + o);
            for (u32 o = (expected & ~15u) + tl; o < expected; o += TEAM) dst[o] = tin[kOut + o];
            if (tl == 0) {
                out_len[b] = op;
                status[b] = SNP_OK;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the next round reuses the buffer)
    }
    append_redo(live && redo && tl == 0, b, list, ctl, sub_cap);
    sample_sizes(live && tl == 0, cap, ctl);
    if (live && redo && tl == 0) status[b] = kRedoStatus;
}

// Two builds of the same body.  Left to itself the kernel takes 88 VGPRs = five wavefronts per SIMD; small blocks need little LDS per
// wavefront (<= 5 KiB: 32 wavefronts per CU fit), and then the registers are what caps the blocks in flight: held to 64 VGPRs (19 spilled
// to scratch) it runs eight per SIMD -- 64-byte blocks 480 -> 565 GB/s, 128 B 462 -> 546, 256 B 365 -> 419 (profiles/r03x_team_occupancy.txt).
// With more LDS per wavefront (384-512-byte blocks: 6.75-9 KiB) LDS caps the CU at 17-23 wavefronts anyway and the spills only cost
// (512 B: 349 -> 333; a six-per-SIMD build for 384-byte blocks measured no gain), so those launches keep the uncapped build.
template <u32 TEAM>
__global__ __launch_bounds__(SNP_WAVE) void k_decompress_teams(SNP_T_PARAMS)
{
    decompress_teams<TEAM>(SNP_T_ARGS);
}
template <u32 TEAM>
__global__ __launch_bounds__(SNP_WAVE) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_decompress_teams_dense(SNP_T_PARAMS)
{
    decompress_teams<TEAM>(SNP_T_ARGS);
}

// When a batch skips the pre-pass (the previous one was all large blocks), this looks at <= 16 384 evenly spaced capacities so
// that the host still learns what the batch was like: ctl[65] += sampled blocks of at most small_max bytes, ctl[66] += the sum of
// the sampled capacities, ctl[67] += how many were sampled.
__global__ __launch_bounds__(SNP_WAVE) void k_sample_caps(const u32* __restrict__ out_cap, u32 nblocks, u32 stride, u32 small_max,
                                                         u32* __restrict__ ctl)
{
    const u64 b = static_cast<u64>(blockIdx.x * SNP_WAVE + threadIdx.x) * stride;
    const bool live = b < nblocks;
    const u32 cap = live ? (out_cap[b] < 0x100000u ? out_cap[b] : 0x100000u) : 0u;
    u32 sum = cap, cnt = live ? 1u : 0u, small = (live && cap <= small_max) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        sum += static_cast<u32>(__shfl_xor(static_cast<int>(sum), d, SNP_WAVE));
        cnt += static_cast<u32>(__shfl_xor(static_cast<int>(cnt), d, SNP_WAVE));
        small += static_cast<u32>(__shfl_xor(static_cast<int>(small), d, SNP_WAVE));
    }
    if (threadIdx.x == 0 && cnt) {
        atomicAdd(&ctl[65], small);
        atomicAdd(&ctl[66], sum >> 4);                                  // (in units of 16 bytes: 16 384 x 64 KiB would overflow)
        atomicAdd(&ctl[67], cnt);
    }
}

}  // namespace

extern "C" hipError_t snp_launch_sample_caps(const u32* out_cap, u32 nblocks, u32 small_max, u32* ctl, hipStream_t stream)
{
    if (nblocks == 0) return hipSuccess;
    const u32 samples = nblocks < 16384u ? nblocks : 16384u;
    const u32 stride = nblocks / samples;
    hipLaunchKernelGGL(k_sample_caps, dim3((samples + SNP_WAVE - 1) / SNP_WAVE), dim3(SNP_WAVE), 0, stream, out_cap, nblocks, stride,
                       small_max, ctl);
    return hipGetLastError();
}

extern "C" hipError_t snp_launch_decompress_small(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                                  const u64* out_off, const u32* out_cap, u32* out_len, i32* status,
                                                  const u8* chunk_type, u32 small_max, hipStream_t stream, u32* list, u32* ctl, u32 sub_cap,
                                                  u32 team_budget)
{
    if (nblocks == 0) return hipSuccess;
    // small_max bit 31: the block-per-lane kernel (kept for A/B); bits 28-30: log2 of the team size (0 = default)
    const u32 lim = small_max & 0x0fffffffu, tlog = (small_max >> 28) & 7u;
    if (small_max & 0x80000000u) {
        const u32 grid = (nblocks + SNP_WAVE - 1) / SNP_WAVE;
        hipLaunchKernelGGL(k_decompress_small, dim3(grid), dim3(SNP_WAVE), 0, stream, in, in_off, in_len, nblocks, out, out_off,
                           out_cap, out_len, status, chunk_type, lim, list, ctl, sub_cap);
        return hipGetLastError();
    }
    const char* be = SNP_GETENV("SNAPPIER_HIP_TEAM_BUDGET");                // LDS bytes per wavefront (experiments; default below)
    const u32 budget = be && atoi(be) >= 1024 && atoi(be) <= 65536 ? static_cast<u32>(atoi(be)) / 16 * 16 : (team_budget ? team_budget : kTeamBudget);
    const bool dense = budget <= 5120;                                  // 32 wavefronts per CU fit by LDS: let the registers allow them too
#define SNP_LAUNCH_TEAMS(T)                                                                                             \
    do {                                                                                                                \
        if (dense)                                                                                                      \
            hipLaunchKernelGGL((k_decompress_teams_dense<T>), dim3((nblocks + SNP_WAVE / T - 1) / (SNP_WAVE / T)), dim3(SNP_WAVE), budget, \
                               stream, in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, lim, list, ctl, sub_cap, budget); \
        else                                                                                                            \
            hipLaunchKernelGGL((k_decompress_teams<T>), dim3((nblocks + SNP_WAVE / T - 1) / (SNP_WAVE / T)), dim3(SNP_WAVE), budget, \
                               stream, in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, lim, list, ctl, sub_cap, budget); \
    } while (0)
    switch (tlog ? tlog : kDefaultTeamLog) {
        case 2: SNP_LAUNCH_TEAMS(4); break;
        case 3: SNP_LAUNCH_TEAMS(8); break;
        default: SNP_LAUNCH_TEAMS(16); break;
    }
#undef SNP_LAUNCH_TEAMS
    return hipGetLastError();
}
