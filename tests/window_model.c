/*
 * window_model.c -- executable CPU model of the ALGORITHM of snappier_amd/csrc/compress_win.hip (the "window"
 * compressor: one <= 64 KiB fragment per wavefront, u16 hash table in LDS, multi-token speculative rounds).
 *
 * TEST INFRASTRUCTURE ONLY (tests/test_window_model.py compares it with the oracle byte for byte on CPU; the HIP
 * kernel cannot run in the build container).  It models the kernel's decisions -- what a round speculates, which
 * prefix of it is accepted, what is written to the table -- position by position; it does not model registers.
 *
 * The reference parse (SnappyCompressor.CompressFragment, Snappier/Internal/SnappyCompressor.cs:174-415) as a state
 * machine over probe EVENTS in position order:
 *   state  (S, pos, pend, next_emit):   S = start of the current scan (the OUTER iteration's ip+1, :198-199),
 *          pos = next position to probe, pend = "insert pos-1 first" (:393-394), pos == S-1 = the post-copy probe (:395-398)
 *   probe at p is legal iff p == S-1, or p + bb <= limit with bb = (32 + p - S) >> 5   (skip = 32 + distance, :227,319-323;
 *          the unrolled 16-probe section :230-313 follows the same rule)
 *   miss:  p == S-1 -> pos = S ; else pos = p + bb          hit: literal [next_emit, p), copy, ip = p + len;
 *          ip >= limit -> remainder ; else S = ip+1, pos = ip, pend
 * A DENSE round speculates a whole window of W = 64*NP consecutive positions against the table as it was BEFORE the
 * round (T0): every position gets hash, T0 candidate, 4-byte compare and a capped match length.  A scalar walk then
 * follows the state machine through the window (several tokens per round).  The speculation is exact iff the
 * positions the walk inserted (probed ones and the ip-1 inserts) have pairwise distinct buckets; otherwise the round
 * is cut at the first position whose bucket an earlier inserted position of the round already used, and re-walked
 * up to there.  A SPARSE round (long scans with stride >= 3, fragment tail) speculates the next 64 probe slots of
 * the current scan and accepts at most one token.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXW 256

typedef struct {
    uint64_t dense_rounds, sparse_rounds, cuts, tokens, dense_tokens, dense_advance, long_resolves, events;
} wm_stats;

static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

static uint32_t crc_step32(uint32_t x)
{
    for (int k = 0; k < 32; ++k) x = (x >> 1) ^ ((x & 1u) ? 0x82F63B78u : 0u);
    return x;
}
static uint16_t g_lut[4][256];
static int g_lut_ready = 0;
static void lut_init(void)
{
    /* crc_step32 is GF(2)-linear: step32(b) = XOR over the four bytes of LUT[k][byte k]  (HashTable.cs:109-112) */
    for (int k = 0; k < 4; ++k)
        for (int v = 0; v < 256; ++v) g_lut[k][v] = (uint16_t)(crc_step32((uint32_t)v << (8 * k)) & 0xffffu);
    g_lut_ready = 1;
}
static uint32_t bucket(uint32_t bytes, uint32_t mask, int variant)
{
    uint32_t hash;
    if (variant == 0) {
        uint32_t x = bytes ^ mask;   /* Sse42.Crc32(crc = bytes, data = mask) = step32(bytes ^ mask) */
        hash = g_lut[0][x & 0xff] ^ g_lut[1][(x >> 8) & 0xff] ^ g_lut[2][(x >> 16) & 0xff] ^ g_lut[3][x >> 24];
    } else {
        hash = (0x1e35a7bdu * bytes) >> 17;   /* HashTable.cs:121-122 */
    }
    return (hash & mask) >> 1;
}

/* probe offsets of the skip heuristic (:227,319-320): D[0] = 0, D[k+1] = D[k] + 1 + (D[k] >> 5), saturated at 65535 */
static uint16_t g_dtab[704];
static uint32_t dtab(uint32_t k) { return g_dtab[k]; }
static void dtab_init(void)
{
    uint32_t v = 0;
    for (int i = 0; i < 704; ++i) { g_dtab[i] = (uint16_t)(v > 0xffffu ? 0xffffu : v); v = v + 1 + (v >> 5); }
}

static uint32_t match_len(const uint8_t* in, uint32_t n, uint32_t cand, uint32_t p)   /* 4 + FindMatchLength  :562-688 */
{
    uint32_t m = 0;
    while (p + m < n && in[cand + m] == in[p + m]) ++m;
    return m;
}

static uint8_t* emit_literal(uint8_t* op, const uint8_t* lit, uint32_t len)   /* :418-464 */
{
    uint32_t k = len - 1;
    if (k < 60) *op++ = (uint8_t)(k << 2);
    else {
        int c = k < 256 ? 1 : k < 65536 ? 2 : 3;
        *op++ = (uint8_t)((59 + c) << 2);
        for (int i = 0; i < c; ++i) *op++ = (uint8_t)(k >> (8 * i));
    }
    memcpy(op, lit, len);
    return op + len;
}
static uint8_t* emit_copy(uint8_t* op, uint32_t off, uint32_t len)            /* :467-543, closed form */
{
    uint32_t q = len >= 68 ? (len - 4) >> 6 : 0, r = len - (q << 6);
    for (uint32_t t = 0; t < q; ++t) { *op++ = (uint8_t)(2 | (63 << 2)); *op++ = (uint8_t)off; *op++ = (uint8_t)(off >> 8); }
    if (r > 64) { *op++ = (uint8_t)(2 | (59 << 2)); *op++ = (uint8_t)off; *op++ = (uint8_t)(off >> 8); r -= 60; }
    if (r < 12 && off < 2048) { *op++ = (uint8_t)(1 | ((r - 4) << 2) | ((off >> 8) << 5)); *op++ = (uint8_t)off; }
    else { *op++ = (uint8_t)(2 | ((r - 1) << 2)); *op++ = (uint8_t)off; *op++ = (uint8_t)(off >> 8); }
    return op;
}

typedef struct { uint32_t S, pos, kb, pend, done; } pstate;   /* pos == S + D[kb] whenever pos >= S */
typedef struct { uint32_t t, cand, len; } token;

/* One walk of the state machine through window positions [0, cut) -- the kernel's scalar walk over ballot masks.
 * A dense round only runs while the scan is in its stride-1 zone (post-copy probe S-1, then S .. S+32), so the probed
 * positions are contiguous; every window position p has p + 17 <= n, hence p + bb <= limit for bb <= 2 (always legal).
 * hit/cand/mlen are the speculative per-position results; visited[] marks positions whose bucket gets written
 * (probes and ip-1 inserts).  Returns the token count. */
static int walk(const uint8_t* in, uint32_t n, uint32_t limit, uint32_t w, uint32_t cut, const uint8_t* hit,
                const uint32_t* cand, uint32_t* mlen, uint8_t* resolved, pstate* st, uint8_t* visited,
                token* tok, wm_stats* s)
{
    int nt = 0;
    memset(visited, 0, MAXW);
    uint32_t o = st->pos - w;
    if (st->pend) { visited[0] = 1; st->pend = 0; }
    for (;;) {
        if (o >= cut) { st->pos = w + o; st->kb = st->pos >= st->S ? st->pos - st->S : 0; return nt; }
        const uint32_t zone_end = st->S + 32 - w;
        const uint32_t lim = zone_end < cut - 1 ? zone_end : cut - 1;
        uint32_t t = o;
        while (t <= lim && !hit[t]) ++t;
        if (t > lim) {
            for (uint32_t q = o; q <= lim; ++q) visited[q] = 1;
            if (lim == zone_end) { st->pos = st->S + 34; st->kb = 33; }   /* probe 33 lies two bytes further (:319-320) */
            else { st->pos = w + cut; st->kb = st->pos >= st->S ? st->pos - st->S : 0; }
            return nt;
        }
        for (uint32_t q = o; q <= t; ++q) visited[q] = 1;
        if (!resolved[t]) { mlen[t] = match_len(in, n, cand[t], w + t); resolved[t] = 1; if (s) s->long_resolves++; }
        const uint32_t m = mlen[t];
        tok[nt].t = w + t; tok[nt].cand = cand[t]; tok[nt].len = m; ++nt;
        const uint32_t ip = w + t + m;
        if (ip >= limit) { st->done = 1; st->pos = ip; return nt; }       /* :381-384 */
        st->S = ip + 1;
        st->pos = ip;
        st->kb = 0;
        if (t + m - 1 < cut) visited[t + m - 1] = 1;
        else { st->pend = 1; return nt; }
        o = t + m;
    }
}

/* Compress one fragment; returns the compressed size.  np = window positions / 64 (1, 2 or 4); cap = bytes of match
 * length every hit position resolves speculatively (multiple of 16). */
size_t wm_compress_fragment(const uint8_t* in, uint32_t n, uint8_t* out, int variant, int np, int cap, wm_stats* s)
{
    if (!g_lut_ready) { dtab_init(); lut_init(); }
    static _Thread_local uint16_t table[16384];
    uint8_t* op = out;
    uint32_t emitted = 0;   /* input position up to which output has been produced */
    if (n >= 15) {
        const uint32_t tsize = n > 16384 ? 16384u : n < 256 ? 256u : (2u << (31 - __builtin_clz(n - 1)));
        const uint32_t mask = 2 * (tsize - 1);
        memset(table, 0, tsize * 2);
        const uint32_t limit = n - 15;
        const uint32_t W = 64u * (uint32_t)np;
        pstate st = {1, 1, 0, 0, 0};
        static _Thread_local uint8_t hit[MAXW], resolved[MAXW], visited[MAXW];
        static _Thread_local uint32_t cand[MAXW], mlen[MAXW], h[MAXW], pp[MAXW], kx[MAXW], nx[MAXW];
        token tok[MAXW];
        while (!st.done) {
            const uint32_t w = st.pend ? st.pos - 1 : st.pos;
            const int zone1 = (st.pos + 1 == st.S) || (st.pos - st.S <= 32);
            uint32_t cut0 = 0;                                       /* window positions p with p + cap + 1 <= n: cap-byte loads stay inside */
            if (w + (uint32_t)cap + 1 <= n) { cut0 = n - (uint32_t)cap - w; if (cut0 > W) cut0 = W; }
            int nt;
            pstate trial;
            if (zone1 && cut0 >= 16) {
                /* ---- dense round ---- */
                for (uint32_t q = 0; q < cut0; ++q) {
                    const uint32_t p = w + q, d = ld32(in + p);
                    h[q] = bucket(d, mask, variant);
                    cand[q] = table[h[q]];
                    hit[q] = ld32(in + cand[q]) == d;
                    mlen[q] = 0; resolved[q] = 1;
                    if (hit[q]) {
                        uint32_t m = 0;
                        resolved[q] = 0;
                        for (int k = 0; k < cap && !resolved[q]; k += 16) {
                            if (k && p + k + 16 > n) break;          /* the lane does not load past the fragment: stays unresolved */
                            int j = 0;
                            while (j < 16 && in[cand[q] + k + j] == in[p + k + j]) ++j;
                            m += j;
                            if (j < 16) resolved[q] = 1;
                        }
                        mlen[q] = m;
                    }
                }
                trial = st;
                nt = walk(in, n, limit, w, cut0, hit, cand, mlen, resolved, &trial, visited, tok, s);
                /* pairwise distinct buckets among the inserted positions?  else cut at the first repeat */
                uint32_t q2 = cut0;
                for (uint32_t a = 1; a < cut0 && q2 == cut0; ++a) {
                    if (!visited[a]) continue;
                    for (uint32_t b = 0; b < a; ++b)
                        if (visited[b] && h[b] == h[a]) { q2 = a; break; }
                }
                if (q2 < cut0) {
                    if (s) s->cuts++;
                    /* The kernel does not walk again: the events before q2 are those of the first walk, and the state at q2
                     * follows from the last token before it.  Both are computed here and must agree. */
                    pstate quick = st;
                    quick.pend = 0;
                    int last = -1;
                    for (int i = 0; i < nt; ++i)
                        if (tok[i].t - w < q2) last = i;
                    if (last < 0) { quick.pos = w + q2; }
                    else {
                        const uint32_t ip = tok[last].t + tok[last].len;          /* absolute */
                        quick.S = ip + 1;
                        if (w + q2 == ip - 1) { quick.pos = ip; quick.pend = 1; }   /* q2 is that copy's ip-1 insert: still pending */
                        else quick.pos = w + q2;
                    }
                    quick.kb = quick.pos >= quick.S ? quick.pos - quick.S : 0;
                    trial = st;
                    nt = walk(in, n, limit, w, q2, hit, cand, mlen, resolved, &trial, visited, tok, NULL);
                    if (quick.S != trial.S || quick.pos != trial.pos || quick.pend != trial.pend || quick.kb != trial.kb || trial.done || nt != last + 1) abort();
                }
                for (uint32_t q = 0; q < cut0; ++q)
                    if (visited[q]) { table[h[q]] = (uint16_t)(w + q); if (s) s->events++; }
                if (s) { s->dense_rounds++; s->dense_tokens += nt; s->dense_advance += (trial.pend ? trial.pos - 1 : trial.pos) - w; }
                st = trial;
            } else {
                /* ---- sparse round: lane j = probe slot j of the current scan (lane 0 = the pending insert), one token at most ---- */
                const uint32_t sh = st.pend ? 1 : 0;
                const int postcopy = st.pos + 1 == st.S;
                uint8_t legal[64], is_ins[64];
                for (uint32_t l = 0; l < 64; ++l) {
                    const uint32_t i = l - sh;
                    is_ins[l] = st.pend && l == 0;
                    if (is_ins[l]) { kx[l] = 0; pp[l] = st.pos - 1; nx[l] = st.pos; legal[l] = 1; }
                    else if (postcopy && i == 0) { kx[l] = 0; pp[l] = st.S - 1; nx[l] = st.S; legal[l] = 1; }
                    else {
                        uint32_t k = postcopy ? i - 1 : st.kb + i;
                        if (k > 702) k = 702;
                        kx[l] = k; pp[l] = st.S + dtab(k); nx[l] = st.S + dtab(k + 1); legal[l] = nx[l] <= limit;
                    }
                }
                uint32_t first0 = 64;
                for (uint32_t l = 0; l < 64; ++l) {
                    hit[l] = 0;
                    if (legal[l]) {
                        const uint32_t d = ld32(in + pp[l]);
                        h[l] = bucket(d, mask, variant);
                        cand[l] = table[h[l]];
                        hit[l] = !is_ins[l] && ld32(in + cand[l]) == d;
                    }
                    if (first0 == 64 && (hit[l] || !legal[l])) first0 = l;
                }
                const int is_hit = first0 < 64 && legal[first0];
                const int terminated = first0 < 64 && !is_hit;
                uint32_t last = is_hit ? first0 + 1 : first0;        /* slots [0, last) are this round's events */
                uint32_t q2 = last;
                for (uint32_t a = 1; a < last && q2 == last; ++a)
                    for (uint32_t b = 0; b < a; ++b)
                        if (h[b] == h[a]) { q2 = a; break; }
                const int cut = q2 < last;
                if (cut) { if (s) s->cuts++; last = q2; }
                for (uint32_t k = 0; k < last; ++k) { table[h[k]] = (uint16_t)pp[k]; if (s) s->events++; }
                nt = 0;
                if (cut) {
                    st.pend = 0;
                    if (!(postcopy && last == sh)) { st.pos = pp[last]; st.kb = kx[last]; }
                } else if (is_hit) {
                    const uint32_t t = pp[first0], m = match_len(in, n, cand[first0], t);
                    tok[0].t = t; tok[0].cand = cand[first0]; tok[0].len = m; nt = 1;
                    const uint32_t ip = t + m;
                    st.pend = 0;
                    if (ip >= limit) { st.done = 1; st.pos = ip; }
                    else { st.S = ip + 1; st.pos = ip; st.kb = 0; st.pend = 1; }
                } else if (terminated) {
                    st.done = 1;
                } else {
                    st.pend = 0;
                    st.pos = nx[63];
                    st.kb = kx[63] + 1;
                }
                if (s) s->sparse_rounds++;
            }
            /* emission of the accepted tokens (the kernel queues them and emits 64 at a time) */
            for (int i = 0; i < nt; ++i) {
                if (tok[i].t > emitted) op = emit_literal(op, in + emitted, tok[i].t - emitted);
                op = emit_copy(op, tok[i].t - tok[i].cand, tok[i].len);
                emitted = tok[i].t + tok[i].len;
            }
            if (s) s->tokens += nt;
        }
    }
    if (emitted < n) op = emit_literal(op, in + emitted, n - emitted);   /* emit_remainder  :406-411 */
    return (size_t)(op - out);
}

/* varint(n) || fragments  (SnappyCompressor.cs:24-83) */
size_t wm_compress(const uint8_t* in, size_t n, uint8_t* out, int variant, int np, int cap, wm_stats* s)
{
    uint8_t* op = out;
    uint32_t v = (uint32_t)n;
    while (v >= 128) { *op++ = (uint8_t)(v | 0x80); v >>= 7; }
    *op++ = (uint8_t)v;
    for (size_t o = 0; o < n; o += 65536) {
        const uint32_t f = n - o < 65536 ? (uint32_t)(n - o) : 65536u;
        op += wm_compress_fragment(in + o, f, op, variant, np, cap, s);
    }
    return (size_t)(op - out);
}
