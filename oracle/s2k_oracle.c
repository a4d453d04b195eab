/*
 * s2k_oracle.c -- CPU restatement of rchikhi/rust-seq2kminmers' seq -> k-min-mer path.
 *
 * TEST INFRASTRUCTURE ONLY (see s2k_oracle.h).  Parity status: PINNED against the
 * reference's own known-answer vectors G1..G4 (tests/test_oracle_golden.py).
 *
 * Plain C11, no dependencies beyond libc + pthreads.  Written from the reference's
 * *behaviour*; every function cites the file:line (under /root/reference) it restates.
 */
#define _GNU_SOURCE /* pthread_barrier_t, clock_gettime under -std=c11 */
#include "s2k_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Seeds.  64-bit ntHash1 seeds; the shipped configuration (H = u32, src/lib.rs:31) keeps the
 * low 32 bits (`as H`, src/nthash_hpc.rs:32-35).  N -> 0, every other byte -> 1
 * (src/nthash_hpc.rs:31,36 / :42,47).
 * ---------------------------------------------------------------------------------------- */
#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL

static inline uint64_t seed64_h(uint8_t c) {
    switch (c) {
    case 'A': return SEED_A;
    case 'C': return SEED_C;
    case 'G': return SEED_G;
    case 'T': return SEED_T;
    case 'N': return 0;
    default:  return 1;
    }
}
static inline uint64_t seed64_rc(uint8_t c) {
    switch (c) {
    case 'A': return SEED_T;
    case 'C': return SEED_G;
    case 'G': return SEED_C;
    case 'T': return SEED_A;
    case 'N': return 0;
    default:  return 1;
    }
}

static uint32_t H32[256], RC32[256];
static uint32_t HS32[256], RCS32[256]; /* SIMD-path seeds (low-nibble mapping) */
static pthread_once_t tables_once = PTHREAD_ONCE_INIT;

/* SIMD base -> code: low nibble through the pshufb table of src/nthash_avx512_32.rs:178-193
 * (nibble 1->0(A) 3->1(C) 7->2(G) 4->3(T) everything else ->4), then code -> seed through
 * permutexvar with seeds {A,C,G,T,0...} (fwd, :225-244) / {T,G,C,A,0...} (rev, :247-262). */
static inline int simd_code(uint8_t c) {
    switch (c & 0x0f) {
    case 1: return 0;
    case 3: return 1;
    case 7: return 2;
    case 4: return 3;
    default: return 4;
    }
}

static void init_tables(void) {
    static const uint32_t fw[5] = {(uint32_t)SEED_A, (uint32_t)SEED_C, (uint32_t)SEED_G, (uint32_t)SEED_T, 0};
    static const uint32_t rv[5] = {(uint32_t)SEED_T, (uint32_t)SEED_G, (uint32_t)SEED_C, (uint32_t)SEED_A, 0};
    for (int c = 0; c < 256; c++) {
        H32[c] = (uint32_t)seed64_h((uint8_t)c);
        RC32[c] = (uint32_t)seed64_rc((uint8_t)c);
        HS32[c] = fw[simd_code((uint8_t)c)];
        RCS32[c] = rv[simd_code((uint8_t)c)];
    }
}
static inline void tables(void) { pthread_once(&tables_once, init_tables); }

uint32_t s2k_oracle_seed_h(uint8_t c) { tables(); return H32[c]; }
uint32_t s2k_oracle_seed_rc(uint8_t c) { tables(); return RC32[c]; }

static inline uint32_t rotl32(uint32_t x, unsigned r) { r &= 31; return r ? (x << r) | (x >> (32 - r)) : x; }
static inline uint32_t rotr32(uint32_t x, unsigned r) { r &= 31; return r ? (x >> r) | (x << (32 - r)) : x; }
static inline uint64_t rotl64(uint64_t x, unsigned r) { r &= 63; return r ? (x << r) | (x >> (64 - r)) : x; }
static inline uint64_t rotr64(uint64_t x, unsigned r) { r &= 63; return r ? (x >> r) | (x << (64 - r)) : x; }

/* src/lib.rs:91 -- `((density as FH) * (H::MAX as FH)) as H` ; Rust float->int casts saturate,
 * NaN -> 0. */
uint32_t s2k_oracle_hash_bound(double density) {
    double v = density * 4294967295.0;
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

/* src/nthash_avx512_32.rs:46-48 -- density = bound / u32::MAX (f64), then
 * ((density as f32) * (u32::MAX as f32)) as u32, saturating. */
uint32_t s2k_oracle_hash_bound_simd(uint32_t hash_bound) {
    double density = (double)hash_bound / 4294967295.0;
    float f = (float)density * (float)4294967295u; /* u32::MAX as f32 == 4294967296.0f */
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 4294967296.0f) return 4294967295u;
    return (uint32_t)f;
}

/* src/lib.rs:157-169 */
uint64_t s2k_oracle_mix32(uint32_t h) {
    uint64_t x = h;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
}

/* ------------------------------------------------------------------------------------------
 * ntHash1, 32-bit, canonical: definitional closed form.
 *   fh(p) = XOR_i rotl(h[s[p+i]], l-1-i)   (src/nthash_hpc.rs:144)
 *   rh(p) = XOR_i rotl(rc[s[p+i]], i)      (src/nthash_hpc.rs:168)
 *   hash  = min(fh, rh)                    (src/nthash_hpc.rs:231,276)
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t nthash32_at(const uint8_t *s, unsigned l, const uint32_t *H, const uint32_t *RC) {
    uint32_t fh = 0, rh = 0;
    for (unsigned i = 0; i < l; i++) {
        fh ^= rotl32(H[s[i]], l - 1 - i);
        rh ^= rotl32(RC[s[i]], i);
    }
    return fh < rh ? fh : rh;
}

size_t s2k_oracle_nthash32_all(const uint8_t *s, size_t n, unsigned l, uint32_t *out) {
    tables();
    if (l == 0 || n < l) return 0;
    for (size_t p = 0; p + l <= n; p++) out[p] = nthash32_at(s + p, l, H32, RC32);
    return n - l + 1;
}

/* ------------------------------------------------------------------------------------------
 * Standalone homopolymer compression (src/hpc.rs).
 * ---------------------------------------------------------------------------------------- */
static inline int is_rle_char(uint8_t c) { /* "ACTGactgNn".contains(c), src/hpc.rs:14 */
    switch (c) {
    case 'A': case 'C': case 'T': case 'G': case 'a': case 'c': case 't': case 'g': case 'N': case 'n': return 1;
    default: return 0;
    }
}

size_t s2k_oracle_hpc(const uint8_t *s, size_t n, int which, uint8_t *out, uint64_t *pos) {
    if (which == 2) {
        /* encode_rle_simd, src/hpc.rs:44-147: keep s[i] iff i==0 or s[i]!=s[i-1] (mask built at
         * :86-95), emit byte and u32 position. An empty input yields an empty result. */
        size_t r = 0;
        for (size_t i = 0; i < n; i++)
            if (i == 0 || s[i] != s[i - 1]) {
                if (out) out[r] = s[i];
                if (pos) pos[r] = i;
                r++;
            }
        return r;
    }
    /* hpc (src/hpc.rs:28-41) / encode_rle (src/hpc.rs:7-25): prev_char starts as '#'; a char
     * equal to prev is skipped (encode_rle: only if it is one of ACTGactgNn); when a new char
     * arrives the previous one is flushed; the last one is flushed at the end -- even for an
     * empty string, which therefore yields "#" (and position 0). */
    uint8_t prev = '#';
    size_t prev_i = 0, r = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = s[i];
        if (c == prev && (which == 0 || is_rle_char(c))) continue;
        if (prev != '#') {
            if (out) out[r] = prev;
            if (pos) pos[r] = prev_i;
            r++;
            prev_i = i;
        }
        prev = c;
    }
    if (out) out[r] = prev;
    if (pos) pos[r] = prev_i;
    r++;
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Minimizer triples.
 * ---------------------------------------------------------------------------------------- */
#define EMIT(J, JE, HV)                                                                             \
    do {                                                                                            \
        if (cnt < cap) {                                                                            \
            if (j) j[cnt] = (J);                                                                    \
            if (jend) jend[cnt] = (JE);                                                             \
            if (hash) hash[cnt] = (HV);                                                             \
        }                                                                                           \
        cnt++;                                                                                      \
    } while (0)

/* Regular: src/lib.rs:215-230.  One canonical u32 ntHash1 per raw l-mer p in [0, n-l]
 * (the external nthash32::NtHashIterator, Cargo.toml:16, pinned by tests/main.rs:41-57),
 * kept iff hash <= bound (lib.rs:228); j = p, jend = p + l - 1 (lib.rs:225-226).
 * Rolling update is the one the in-tree scalar iterator uses (src/nthash_hpc.rs:245-249). */
static size_t minimizers_regular(const uint8_t *s, size_t n, unsigned l, uint32_t bound, uint64_t *j,
                                 uint64_t *jend, uint32_t *hash, size_t cap) {
    size_t cnt = 0;
    if (n <= l || l == 0) return 0; /* src/lib.rs:97 */
    uint32_t fh = 0, rh = 0;
    for (unsigned i = 0; i < l; i++) {
        fh ^= rotl32(H32[s[i]], l - 1 - i);
        rh ^= rotl32(RC32[s[i]], i);
    }
    for (size_t p = 0;; p++) {
        uint32_t hv = fh < rh ? fh : rh;
        if (hv <= bound) EMIT(p, p + l - 1, hv);
        if (p + l >= n) break;
        fh = rotl32(fh, 1) ^ rotl32(H32[s[p]], l) ^ H32[s[p + l]];
        rh = rotr32(rh, 1) ^ rotr32(RC32[s[p]], 1) ^ rotl32(RC32[s[p + l]], l - 1);
    }
    return cnt;
}

/* Hpc, closed form of NtHashHPCIterator (src/nthash_hpc.rs:115-283):
 *   runs of byte-equal characters (:138-150,:253-263) -> hs[0..R), st[r] = raw start of run r;
 *   l-mers p in [0, R-l-1] only: the roll to p = R-l happens but the end-of-sequence check
 *   (:265-267) returns None before the bound test, so the last HPC l-mer is never emitted;
 *   item = (st[p], st[p+l]-1, hash) (:233-234, :280-281); kept iff hash <= bound (:232,:277). */
static size_t minimizers_hpc(const uint8_t *s, size_t n, unsigned l, uint32_t bound, uint64_t *j,
                             uint64_t *jend, uint32_t *hash, size_t cap) {
    size_t cnt = 0;
    if (n <= l || l == 0) return 0; /* src/lib.rs:97 */
    uint8_t *hs = (uint8_t *)malloc(n);
    uint64_t *st = (uint64_t *)malloc(n * sizeof(uint64_t));
    size_t R = 0;
    for (size_t i = 0; i < n; i++)
        if (i == 0 || s[i] != s[i - 1]) {
            hs[R] = s[i];
            st[R] = i;
            R++;
        }
    if (R > l) {
        uint32_t fh = 0, rh = 0;
        for (unsigned i = 0; i < l; i++) {
            fh ^= rotl32(H32[hs[i]], l - 1 - i);
            rh ^= rotl32(RC32[hs[i]], i);
        }
        for (size_t p = 0; p + l < R; p++) { /* p <= R-l-1 */
            uint32_t hv = fh < rh ? fh : rh;
            if (hv <= bound) EMIT(st[p], st[p + l] - 1, hv);
            fh = rotl32(fh, 1) ^ rotl32(H32[hs[p]], l) ^ H32[hs[p + l]];
            rh = rotr32(rh, 1) ^ rotr32(RC32[hs[p]], 1) ^ rotl32(RC32[hs[p + l]], l - 1);
        }
    }
    free(hs);
    free(st);
    return cnt;
}

/* SIMD iterator semantics over a string t[0..m) (src/nthash_avx512_32.rs:32-164):
 *   assert l <= 31 (:33); nothing if m < l (:87); sentinel = m - l + 1 (:91);
 *   strict `<` against the f32-recomputed bound (:46-48,:55,:130);
 *   seeds through the low-nibble table (:178-193,:225-262);
 *   the tail mask (1 << (sentinel % 16)) - 1 is applied to the last 16-block when it is not the
 *   first one (:134-138) -- when sentinel % 16 == 0 that mask is 0 and the whole final block is
 *   dropped; the first block (built in new(), :51-58) is only filtered by `pos >= sentinel` (:96).
 * Emits positions in t-space through cb. */
typedef void (*simd_cb)(void *ctx, size_t p, uint32_t hv);
static void simd_scan(const uint8_t *t, size_t m, unsigned l, uint32_t bound, simd_cb cb, void *ctx) {
    if (l == 0 || l > 31 || m < l) return;
    uint32_t b2 = s2k_oracle_hash_bound_simd(bound);
    size_t sentinel = m - l + 1;
    size_t limit = sentinel;
    if (sentinel % 16 == 0 && sentinel >= 32) limit = sentinel - 16;
    for (size_t p = 0; p < limit; p++) {
        uint32_t hv = nthash32_at(t + p, l, HS32, RCS32);
        if (hv < b2) cb(ctx, p, hv);
    }
}

struct emit_ctx {
    uint64_t *j, *jend;
    uint32_t *hash;
    size_t cap, cnt;
    unsigned l;
    const uint64_t *st; /* HpcSimd: run starts */
};
static void emit_simd(void *c_, size_t p, uint32_t hv) { /* src/lib.rs:202: jend = j + l - 1 */
    struct emit_ctx *c = (struct emit_ctx *)c_;
    if (c->cnt < c->cap) {
        if (c->j) c->j[c->cnt] = p;
        if (c->jend) c->jend[c->cnt] = p + c->l - 1;
        if (c->hash) c->hash[c->cnt] = hv;
    }
    c->cnt++;
}
static void emit_hpcsimd(void *c_, size_t p, uint32_t hv) { /* src/nthash_hpc_simd.rs:64 */
    struct emit_ctx *c = (struct emit_ctx *)c_;
    if (c->cnt < c->cap) {
        if (c->j) c->j[c->cnt] = c->st[p];
        if (c->jend) c->jend[c->cnt] = c->st[p + c->l - 1]; /* START of the last run */
        if (c->hash) c->hash[c->cnt] = hv;
    }
    c->cnt++;
}

size_t s2k_oracle_minimizers(const uint8_t *s, size_t n, unsigned l, uint32_t bound, int mode, uint64_t *j,
                             uint64_t *jend, uint32_t *hash, size_t cap) {
    tables();
    switch (mode) {
    case S2K_O_REGULAR: return minimizers_regular(s, n, l, bound, j, jend, hash, cap);
    case S2K_O_HPC: return minimizers_hpc(s, n, l, bound, j, jend, hash, cap);
    case S2K_O_SIMD: {
        if (n <= l) return 0; /* src/lib.rs:97 */
        struct emit_ctx c = {j, jend, hash, cap, 0, l, NULL};
        simd_scan(s, n, l, bound, emit_simd, &c);
        return c.cnt;
    }
    case S2K_O_HPCSIMD: {
        if (n <= l) return 0; /* src/lib.rs:97 */
        uint8_t *hs = (uint8_t *)malloc(n);
        uint64_t *st = (uint64_t *)malloc(n * sizeof(uint64_t));
        size_t R = s2k_oracle_hpc(s, n, 2, hs, st); /* src/nthash_hpc_simd.rs:36 */
        struct emit_ctx c = {j, jend, hash, cap, 0, l, st};
        simd_scan(hs, R, l, bound, emit_hpcsimd, &c);
        free(hs);
        free(st);
        return c.cnt;
    }
    default: return 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * Literal transliteration of NtHashHPCIterator (src/nthash_hpc.rs:115-283), H = u32.
 * Kept structurally close to the state machine on purpose: it exists to prove that the closed
 * form above (runs + "drop the last l-mer" + end = st[p+l]-1) is what the iterator computes.
 * The reference's 1-past-the-end reads (:213-215) are guarded here; they never influence results.
 * ---------------------------------------------------------------------------------------- */
#define BUFLEN 256
size_t s2k_oracle_hpc_literal(const uint8_t *seq, size_t seq_len, unsigned k, uint32_t hash_bound, uint64_t *j,
                              uint64_t *jend, uint32_t *hash, size_t cap) {
    tables();
    size_t cnt = 0;
    if (k == 0 || k > seq_len || k >= BUFLEN) return 0; /* :117-125,:133 */
    uint32_t hbuf[BUFLEN], rcbuf[BUFLEN];
    size_t idxbuf[BUFLEN];
    memset(hbuf, 0, sizeof hbuf);
    memset(rcbuf, 0, sizeof rcbuf);
    memset(idxbuf, 0, sizeof idxbuf);
    /* new(): :126-189 */
    uint32_t fh = 0, rh = 0;
    size_t jj = 0, i = 0, prev_j = 0;
    uint8_t v, prev;
    while (i < k && jj < seq_len) {
        v = seq[jj];
        hbuf[i] = H32[v];
        idxbuf[i] = jj;
        fh ^= rotl32(H32[v], k - i - 1);
        i++;
        prev = v;
        prev_j = jj;
        while (jj < seq_len && seq[jj] == prev) jj++;
    }
    i -= 1;
    jj = prev_j;
    size_t cur_idx_plus_k = jj;
    for (;;) {
        v = seq[jj];
        rcbuf[i] = RC32[v];
        rh ^= rotl32(RC32[v], (unsigned)i);
        if (i == 0) break;
        i--;
        prev = v;
        while (jj > 0 && seq[jj] == prev) jj--;
    }
    /* next(): :196-283, called until None */
    size_t buffer_pos = 0;
    int first = 1;
    for (;;) {
        uint32_t hv, h_seqk, rc_seqk;
        uint8_t pv = seq[cur_idx_plus_k], cur = pv;
        int have = 0;
        if (first) { /* :208-236 */
            first = 0;
            for (;;) {
                cur_idx_plus_k++;
                if (cur_idx_plus_k >= seq_len) break;
                cur = seq[cur_idx_plus_k];
                if (cur != pv) break;
            }
            if (cur_idx_plus_k >= seq_len) return cnt;
            h_seqk = H32[cur];
            rc_seqk = RC32[cur];
            size_t pos = (buffer_pos + k) % BUFLEN;
            hbuf[pos] = h_seqk;
            rcbuf[pos] = rc_seqk;
            idxbuf[pos] = cur_idx_plus_k;
            buffer_pos++;
            hv = rh < fh ? rh : fh;
            if (hv <= hash_bound) have = 1;
        } else { /* :238 */
            h_seqk = hbuf[(buffer_pos + k - 1) % BUFLEN];
            rc_seqk = rcbuf[(buffer_pos + k - 1) % BUFLEN];
            hv = 0;
        }
        while (!have) { /* :241-278 */
            uint32_t h_seqi = hbuf[(buffer_pos - 1) % BUFLEN], rc_seqi = rcbuf[(buffer_pos - 1) % BUFLEN];
            fh = rotl32(fh, 1) ^ rotl32(h_seqi, k) ^ h_seqk;
            rh = rotr32(rh, 1) ^ rotr32(rc_seqi, 1) ^ rotl32(rc_seqk, k - 1);
            pv = cur;
            for (;;) {
                cur_idx_plus_k++;
                if (cur_idx_plus_k >= seq_len) break;
                cur = seq[cur_idx_plus_k];
                if (cur != pv) break;
            }
            if (cur_idx_plus_k >= seq_len) return cnt; /* :265-267: before the bound test */
            h_seqk = H32[cur];
            rc_seqk = RC32[cur];
            size_t pos = (buffer_pos + k) % BUFLEN;
            hbuf[pos] = h_seqk;
            rcbuf[pos] = rc_seqk;
            idxbuf[pos] = cur_idx_plus_k;
            buffer_pos++;
            hv = rh < fh ? rh : fh;
            if (hv <= hash_bound) have = 1;
        }
        EMIT(idxbuf[(buffer_pos + BUFLEN - 1) % BUFLEN], cur_idx_plus_k - 1, hv); /* :280-281 */
    }
}

/* Same machine, H = u64: the archived configuration whose KAT is src/old/nthash_hpc.rs.orig:68-77. */
size_t s2k_oracle_hpc_literal_u64(const uint8_t *seq, size_t seq_len, unsigned k, uint64_t hash_bound,
                                  uint64_t *j, uint64_t *hash, size_t cap) {
    size_t cnt = 0;
    uint64_t *jend = NULL;
    if (k == 0 || k > seq_len || k >= BUFLEN) return 0;
    uint64_t hbuf[BUFLEN], rcbuf[BUFLEN];
    size_t idxbuf[BUFLEN];
    memset(hbuf, 0, sizeof hbuf);
    memset(rcbuf, 0, sizeof rcbuf);
    memset(idxbuf, 0, sizeof idxbuf);
    uint64_t fh = 0, rh = 0;
    size_t jj = 0, i = 0, prev_j = 0;
    uint8_t v, prev;
    while (i < k && jj < seq_len) {
        v = seq[jj];
        hbuf[i] = seed64_h(v);
        idxbuf[i] = jj;
        fh ^= rotl64(seed64_h(v), k - i - 1);
        i++;
        prev = v;
        prev_j = jj;
        while (jj < seq_len && seq[jj] == prev) jj++;
    }
    i -= 1;
    jj = prev_j;
    size_t cur_idx_plus_k = jj;
    for (;;) {
        v = seq[jj];
        rcbuf[i] = seed64_rc(v);
        rh ^= rotl64(seed64_rc(v), (unsigned)i);
        if (i == 0) break;
        i--;
        prev = v;
        while (jj > 0 && seq[jj] == prev) jj--;
    }
    size_t buffer_pos = 0;
    int first = 1;
    for (;;) {
        uint64_t hv, h_seqk, rc_seqk;
        uint8_t pv = seq[cur_idx_plus_k], cur = pv;
        int have = 0;
        if (first) {
            first = 0;
            for (;;) {
                cur_idx_plus_k++;
                if (cur_idx_plus_k >= seq_len) break;
                cur = seq[cur_idx_plus_k];
                if (cur != pv) break;
            }
            if (cur_idx_plus_k >= seq_len) return cnt;
            h_seqk = seed64_h(cur);
            rc_seqk = seed64_rc(cur);
            size_t pos = (buffer_pos + k) % BUFLEN;
            hbuf[pos] = h_seqk;
            rcbuf[pos] = rc_seqk;
            idxbuf[pos] = cur_idx_plus_k;
            buffer_pos++;
            hv = rh < fh ? rh : fh;
            if (hv <= hash_bound) have = 1;
        } else {
            h_seqk = hbuf[(buffer_pos + k - 1) % BUFLEN];
            rc_seqk = rcbuf[(buffer_pos + k - 1) % BUFLEN];
            hv = 0;
        }
        while (!have) {
            uint64_t h_seqi = hbuf[(buffer_pos - 1) % BUFLEN], rc_seqi = rcbuf[(buffer_pos - 1) % BUFLEN];
            fh = rotl64(fh, 1) ^ rotl64(h_seqi, k) ^ h_seqk;
            rh = rotr64(rh, 1) ^ rotr64(rc_seqi, 1) ^ rotl64(rc_seqk, k - 1);
            pv = cur;
            for (;;) {
                cur_idx_plus_k++;
                if (cur_idx_plus_k >= seq_len) break;
                cur = seq[cur_idx_plus_k];
                if (cur != pv) break;
            }
            if (cur_idx_plus_k >= seq_len) return cnt;
            h_seqk = seed64_h(cur);
            rc_seqk = seed64_rc(cur);
            size_t pos = (buffer_pos + k) % BUFLEN;
            hbuf[pos] = h_seqk;
            rcbuf[pos] = rc_seqk;
            idxbuf[pos] = cur_idx_plus_k;
            buffer_pos++;
            hv = rh < fh ? rh : fh;
            if (hv <= hash_bound) have = 1;
        }
        EMIT(idxbuf[(buffer_pos + BUFLEN - 1) % BUFLEN], cur_idx_plus_k - 1, hv);
    }
}

/* ------------------------------------------------------------------------------------------
 * k-min-mer layer (src/lib.rs:231-266; closed form src/lib.rs:275-288).
 * ---------------------------------------------------------------------------------------- */

/* Literal rolling form: warm-up XORs (:262-266), first window (:238-242), rolling update
 * (:243-249), canonical + rev (:250-251). */
size_t s2k_oracle_kminmer_hashes_rolling(const uint32_t *mh, size_t m, unsigned k, uint64_t *hash, uint8_t *rev) {
    if (k == 0) return 0;
    uint64_t f = 0, r = 0;
    size_t count = 0;
    uint64_t *sk = (uint64_t *)malloc((m ? m : 1) * sizeof(uint64_t));
    for (size_t len = 1; len <= m; len++) {
        uint64_t x = s2k_oracle_mix32(mh[len - 1]); /* :232 */
        sk[len - 1] = x;
        if (len >= k) {
            if (len == k) {
                f ^= rotl64(x, (unsigned)(k - 1 - (len - 1)));
                r ^= rotl64(x, (unsigned)(len - 1));
            } else {
                f = rotl64(f, 1) ^ x ^ rotl64(sk[count - 1], k);
                r = rotr64(r, 1) ^ rotl64(x, k - 1) ^ rotr64(sk[count - 1], 1);
            }
            if (hash) hash[count] = f < r ? f : r;
            if (rev) rev[count] = r < f;
            count++;
        } else {
            f ^= rotl64(x, (unsigned)(k - 1 - (len - 1)));
            r ^= rotl64(x, (unsigned)(len - 1));
        }
    }
    free(sk);
    return count;
}

/* One read.  For every window c of k consecutive minimizers:
 *   F  = XOR_i rotl64(mix(m[c+i]), k-1-i),  Rv = XOR_i rotl64(mix(m[c+i]), i)
 *   KminmerHash{hash=min(F,Rv), start=j[c], end=jend[c+k-1], offset=c, rev=Rv<F}  (:250-258) */
static size_t kminmers_one(const uint8_t *s, size_t n, unsigned l, unsigned k, uint32_t bound, int mode,
                           uint64_t *hash, uint64_t *start, uint64_t *end, uint8_t *rev, size_t cap,
                           uint64_t **scratch, size_t *scratch_cap) {
    if (k == 0) return 0;
    size_t M = s2k_oracle_minimizers(s, n, l, bound, mode, NULL, NULL, NULL, 0);
    if (M < k) return 0;
    size_t cnt = M - k + 1;
    if (!hash && !start && !end && !rev) return cnt;
    size_t need = M * 3; /* j, jend as u64; hash32 + mixed packed in third */
    if (*scratch_cap < need + M) {
        free(*scratch);
        *scratch_cap = (need + M) * 2;
        *scratch = (uint64_t *)malloc(*scratch_cap * sizeof(uint64_t));
    }
    uint64_t *mj = *scratch, *mje = mj + M, *mx = mje + M;
    uint32_t *mh = (uint32_t *)(mx + M);
    s2k_oracle_minimizers(s, n, l, bound, mode, mj, mje, mh, M);
    for (size_t i = 0; i < M; i++) mx[i] = s2k_oracle_mix32(mh[i]);
    for (size_t c = 0; c < cnt && c < cap; c++) {
        uint64_t F = 0, Rv = 0;
        for (unsigned i = 0; i < k; i++) {
            F ^= rotl64(mx[c + i], k - 1 - i);
            Rv ^= rotl64(mx[c + i], i);
        }
        if (hash) hash[c] = F < Rv ? F : Rv;
        if (rev) rev[c] = Rv < F;
        if (start) start[c] = mj[c];
        if (end) end[c] = mje[c + k - 1];
    }
    return cnt;
}

size_t s2k_oracle_kminmers(const uint8_t *s, size_t n, unsigned l, unsigned k, double density, int mode,
                           uint64_t *hash, uint64_t *start, uint64_t *end, uint8_t *rev, size_t cap) {
    tables();
    uint64_t *scratch = NULL;
    size_t scap = 0;
    size_t r = kminmers_one(s, n, l, k, s2k_oracle_hash_bound(density), mode, hash, start, end, rev, cap, &scratch,
                            &scap);
    free(scratch);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Batch driver: threads over reads, the shape of src/main.rs:65-79 (parallel_fastx workers,
 * one KminmersIterator per read).  Count-only mode consumes the iterator like main.rs:69-73.
 * For the count-only/timing path the Hpc mode runs the literal state machine and Regular the
 * rolling scan, i.e. the same per-base work as the reference's scalar iterators.
 * ---------------------------------------------------------------------------------------- */
struct batch_job {
    const uint8_t *bases;
    const uint64_t *off;
    uint64_t r0, r1;
    unsigned l, k;
    uint32_t bound;
    int mode;
    uint64_t *cnt; /* per-read counts or NULL */
    uint64_t total;
    /* write pass */
    const uint64_t *km_off;
    uint64_t *hash;
    uint32_t *start, *end;
    uint8_t *rev;
    uint64_t cap;
};

static uint64_t count_one(const uint8_t *s, size_t n, unsigned l, unsigned k, uint32_t bound, int mode) {
    size_t M;
    if (n <= l) return 0;
    if (mode == S2K_O_HPC)
        M = s2k_oracle_hpc_literal(s, n, l, bound, NULL, NULL, NULL, 0);
    else
        M = s2k_oracle_minimizers(s, n, l, bound, mode, NULL, NULL, NULL, 0);
    return M >= k ? M - k + 1 : 0;
}

static void *batch_count_worker(void *p) {
    struct batch_job *jb = (struct batch_job *)p;
    uint64_t tot = 0;
    for (uint64_t r = jb->r0; r < jb->r1; r++) {
        uint64_t c = count_one(jb->bases + jb->off[r], (size_t)(jb->off[r + 1] - jb->off[r]), jb->l, jb->k,
                               jb->bound, jb->mode);
        if (jb->cnt) jb->cnt[r] = c;
        tot += c;
    }
    jb->total = tot;
    return NULL;
}

static void *batch_write_worker(void *p) {
    struct batch_job *jb = (struct batch_job *)p;
    uint64_t *scratch = NULL, *h64 = NULL, *st = NULL, *en = NULL;
    size_t scap = 0, ocap = 0;
    for (uint64_t r = jb->r0; r < jb->r1; r++) {
        uint64_t c = jb->km_off[r + 1] - jb->km_off[r];
        if (!c) continue;
        if (ocap < c) {
            free(h64);
            free(st);
            free(en);
            ocap = c * 2;
            h64 = (uint64_t *)malloc(ocap * 8);
            st = (uint64_t *)malloc(ocap * 8);
            en = (uint64_t *)malloc(ocap * 8);
        }
        uint64_t base = jb->km_off[r];
        uint8_t *rv = (jb->rev && base < jb->cap) ? jb->rev + base : NULL;
        uint64_t room = base < jb->cap ? jb->cap - base : 0;
        kminmers_one(jb->bases + jb->off[r], (size_t)(jb->off[r + 1] - jb->off[r]), jb->l, jb->k, jb->bound,
                     jb->mode, h64, st, en, rv, (size_t)(c < room ? c : room), &scratch, &scap);
        for (uint64_t i = 0; i < c && i < room; i++) {
            if (jb->hash) jb->hash[base + i] = h64[i];
            if (jb->start) jb->start[base + i] = (uint32_t)st[i];
            if (jb->end) jb->end[base + i] = (uint32_t)en[i];
        }
    }
    free(scratch);
    free(h64);
    free(st);
    free(en);
    return NULL;
}

static void run_jobs(struct batch_job *proto, uint64_t n_reads, int threads, void *(*fn)(void *), uint64_t *total) {
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    struct batch_job *jobs = (struct batch_job *)malloc(sizeof(*jobs) * threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(*th) * threads);
    uint64_t total_bases = n_reads ? proto->off[n_reads] - proto->off[0] : 0;
    uint64_t r = 0;
    for (int t = 0; t < threads; t++) { /* contiguous shards balanced by cumulative bases */
        jobs[t] = *proto;
        jobs[t].r0 = r;
        uint64_t target = proto->off[0] + (total_bases * (uint64_t)(t + 1)) / (uint64_t)threads;
        if (t == threads - 1) r = n_reads;
        else
            while (r < n_reads && proto->off[r + 1] <= target) r++;
        jobs[t].r1 = r;
        jobs[t].total = 0;
    }
    if (threads == 1) fn(&jobs[0]);
    else {
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, fn, &jobs[t]);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    uint64_t tot = 0;
    for (int t = 0; t < threads; t++) tot += jobs[t].total;
    if (total) *total = tot;
    free(jobs);
    free(th);
}

uint64_t s2k_oracle_batch(const uint8_t *bases, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                          double density, int mode, int threads, uint64_t *km_off, uint64_t *hash,
                          uint32_t *start, uint32_t *end, uint8_t *rev, uint64_t cap) {
    tables();
    struct batch_job proto;
    memset(&proto, 0, sizeof proto);
    proto.bases = bases;
    proto.off = off;
    proto.l = l;
    proto.k = k;
    proto.bound = s2k_oracle_hash_bound(density);
    proto.mode = mode;
    int want_out = hash || start || end || rev;
    uint64_t *cnt = NULL;
    if (km_off || want_out) cnt = (uint64_t *)malloc(sizeof(uint64_t) * (n_reads + 1));
    proto.cnt = cnt;
    uint64_t total = 0;
    if (k == 0) {
        if (km_off) memset(km_off, 0, sizeof(uint64_t) * (n_reads + 1));
        free(cnt);
        return 0;
    }
    run_jobs(&proto, n_reads, threads, batch_count_worker, &total);
    if (cnt) {
        uint64_t *ko = km_off ? km_off : cnt;
        uint64_t acc = 0;
        for (uint64_t r = 0; r < n_reads; r++) {
            uint64_t c = cnt[r];
            ko[r] = acc;
            acc += c;
        }
        ko[n_reads] = acc;
        if (want_out) {
            proto.km_off = ko;
            proto.hash = hash;
            proto.start = start;
            proto.end = end;
            proto.rev = rev;
            proto.cap = cap;
            run_jobs(&proto, n_reads, threads, batch_write_worker, NULL);
        }
    }
    free(cnt);
    return total;
}

/* ------------------------------------------------------------------------------------------
 * Timed count-only pass for bench.py's cpu_baseline: the worker threads are created first, each touches its shard once
 * (warm-up, untimed), all meet at a barrier, then every thread iterates `repeats` times over its shard; the clock runs
 * from the barrier to the last join.  (Timing thread creation and a 10 ms pass, as round 1 did, under-reported the
 * all-core rate several times.)  Returns the k-min-mer total of ONE pass; *seconds = wall time of the timed passes.
 * ---------------------------------------------------------------------------------------- */
#include <time.h>
struct timed_job {
    struct batch_job jb;
    int repeats;
    pthread_barrier_t *bar;
};
static void *timed_worker(void *p) {
    struct timed_job *tj = (struct timed_job *)p;
    struct batch_job warm = tj->jb;
    if (warm.r1 > warm.r0 + 8) warm.r1 = warm.r0 + 8; /* a few reads: page in the code, the tables and the thread */
    batch_count_worker(&warm);
    pthread_barrier_wait(tj->bar);
    uint64_t tot = 0;
    for (int i = 0; i < tj->repeats; i++) {
        batch_count_worker(&tj->jb);
        tot = tj->jb.total;
    }
    tj->jb.total = tot;
    return NULL;
}
uint64_t s2k_oracle_batch_count_timed(const uint8_t *bases, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                                      double density, int mode, int threads, int repeats, double *seconds) {
    tables();
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    if (repeats < 1) repeats = 1;
    struct timed_job *jobs = (struct timed_job *)calloc((size_t)threads, sizeof(*jobs));
    pthread_t *th = (pthread_t *)malloc(sizeof(*th) * (size_t)threads);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads + 1);
    uint64_t total_bases = n_reads ? off[n_reads] - off[0] : 0, r = 0;
    for (int t = 0; t < threads; t++) { /* contiguous shards balanced by cumulative bases */
        struct batch_job *jb = &jobs[t].jb;
        jb->bases = bases; jb->off = off; jb->l = l; jb->k = k; jb->bound = s2k_oracle_hash_bound(density); jb->mode = mode;
        jb->r0 = r;
        uint64_t target = off[0] + (total_bases * (uint64_t)(t + 1)) / (uint64_t)threads;
        if (t == threads - 1) r = n_reads;
        else
            while (r < n_reads && off[r + 1] <= target) r++;
        jb->r1 = r;
        jobs[t].repeats = repeats;
        jobs[t].bar = &bar;
        pthread_create(&th[t], NULL, timed_worker, &jobs[t]);
    }
    struct timespec a, b;
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &a);
    uint64_t tot = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        tot += jobs[t].jb.total;
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (seconds) *seconds = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    pthread_barrier_destroy(&bar);
    free(jobs);
    free(th);
    return k ? tot : 0;
}

/* ------------------------------------------------------------------------------------------
 * HiFi-like synthetic reads (SURVEY.md 8d C4 / BASELINE configs[3]): read r is a sequence of homopolymer runs drawn from
 * splitmix64 keyed by (seed, r) -- run lengths geometric with mean 2, about 0.1 % of the runs stretched to 20..2999 bases (so
 * that the l+1 run heads an l-mer needs overflow any fixed look-ahead), consecutive runs of different letters -- and its length
 * is ~N(15 000, 2 000) (eight 16-bit uniforms, integer arithmetic only), clipped to [2 000, 30 000].  The device generator of
 * the product library (csrc/s2k_util.hip: synth_hifi_kernel) implements the identical functions.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t z);
static inline uint64_t hifi_key(uint64_t seed, uint64_t r) { return seed ^ (r * 0xA24BAED4963EE407ULL) ^ 0x9FB21C651E98DF25ULL; }

uint64_t s2k_oracle_hifi_len(uint64_t seed, uint64_t r) {
    const uint64_t z0 = splitmix64(hifi_key(seed, r) ^ 0x5851F42D4C957F2DULL), z1 = splitmix64(z0);
    int64_t s = 0;
    for (int i = 0; i < 4; i++) s += (int64_t)((z0 >> (16 * i)) & 0xFFFF) + (int64_t)((z1 >> (16 * i)) & 0xFFFF);
    int64_t len = 15000 + ((s - 262140) * 2000) / 53510; /* eight uniforms: mean 262 140, sigma 53 510 */
    return (uint64_t)(len < 2000 ? 2000 : len > 30000 ? 30000 : len);
}

void s2k_oracle_hifi_lens(uint64_t seed, uint64_t r0, uint64_t n, uint64_t *out) {
    for (uint64_t i = 0; i < n; i++) out[i] = s2k_oracle_hifi_len(seed, r0 + i);
}

void s2k_oracle_hifi_read(uint64_t seed, uint64_t r, uint64_t n, uint8_t *out) {
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    uint64_t x = splitmix64(hifi_key(seed, r));
    uint32_t letter = 0;
    for (uint64_t pos = 0, run = 0; pos < n; run++) {
        x = splitmix64(x);
        uint64_t len = 1 + (uint64_t)__builtin_ctz((uint32_t)x | 0x80000000u); /* geometric, mean 2 */
        if (((x >> 32) & 1023) == 0) len = 20 + ((x >> 42) & 0x3FFF) % 2980;   /* ~0.1 % of the runs */
        letter = run == 0 ? (uint32_t)(x >> 60) & 3u : (letter + 1u + (uint32_t)((x >> 56) & 15u) % 3u) & 3u;
        for (; len && pos < n; len--) out[pos++] = (uint8_t)ACGT[letter];
    }
}

/* ------------------------------------------------------------------------------------------
 * Whole-run checksums of a synthetic batch (n_reads reads of read_len bases cut from the synth_bases stream
 * of `seed`), generated read by read so that BASELINE-size runs (10 Gbp) need no 10 GB host buffer.
 * out = { n_minimizers, n_kminmers, XOR hash, SUM start, SUM end, #rev,
 *         FOLD hash, FOLD start, FOLD end, FOLD rev, FOLD km_off, FOLD per-read count } where
 *   FOLD x      = SUM_g x[g] * (2 g + 1) mod 2^64, g = index of the item in the whole output (reads in order),
 *   FOLD km_off = SUM_r km_off[r] * (2 r + 1) mod 2^64 over r = 0 .. n_reads (the end entry included).
 * XOR / SUM do not see a permutation of the items; the folds do (two tiles' outputs swapped, a wrong km_off,
 * one read's items in another's place all change them).
 * ---------------------------------------------------------------------------------------- */
#define S2K_NSUM 12
struct sum_job {
    const uint64_t *off; /* NULL: reads of read_len bases each; else read r = stream [off[r], off[r+1]) */
    uint64_t seed, r0, r1, read_len;
    int hifi; /* reads generated per read (s2k_oracle_hifi_read) instead of cut from the uniform stream */
    unsigned l, k;
    uint32_t bound;
    int mode;
    uint64_t out[S2K_NSUM];
    /* thread-local pieces of the folds: items are numbered from 0 inside the thread's read range, the combiner adds the base */
    uint64_t sum_hash; /* SUM hash (mod 2^64) */
};

static void *sum_worker(void *p) {
    struct sum_job *jb = (struct sum_job *)p;
    uint64_t maxlen = jb->read_len;
    if (jb->off)
        for (uint64_t r = jb->r0; r < jb->r1; r++)
            if (jb->off[r + 1] - jb->off[r] > maxlen) maxlen = jb->off[r + 1] - jb->off[r];
    uint8_t *buf = (uint8_t *)malloc(maxlen + 1);
    size_t ocap = maxlen + 1, scap = 0;
    uint64_t *h = (uint64_t *)malloc(ocap * 8), *st = (uint64_t *)malloc(ocap * 8), *en = (uint64_t *)malloc(ocap * 8);
    uint8_t *rv = (uint8_t *)malloc(ocap);
    uint64_t *scratch = NULL;
    for (uint64_t r = jb->r0; r < jb->r1; r++) {
        const uint64_t a = jb->off ? jb->off[r] : r * jb->read_len, n = jb->off ? jb->off[r + 1] - a : jb->read_len;
        if (jb->hifi) s2k_oracle_hifi_read(jb->seed, r, n, buf);
        else s2k_oracle_synth_bases(jb->seed, a, n, buf);
        size_t M = s2k_oracle_minimizers(buf, n, jb->l, jb->bound, jb->mode, NULL, NULL, NULL, 0);
        size_t c = kminmers_one(buf, n, jb->l, jb->k, jb->bound, jb->mode, h, st, en, rv, ocap, &scratch, &scap);
        /* km_off of this read, counted from the thread's first read */
        jb->out[10] += jb->out[1] * (2 * r + 1);
        jb->out[11] += (uint64_t)c * (2 * r + 1);
        jb->out[0] += M;
        for (size_t i = 0; i < c; i++) {
            const uint64_t w = 2 * (jb->out[1] + i) + 1; /* local item index */
            jb->out[2] ^= h[i];
            jb->out[3] += st[i];
            jb->out[4] += en[i];
            jb->out[5] += rv[i];
            jb->sum_hash += h[i];
            jb->out[6] += h[i] * w;
            jb->out[7] += st[i] * w;
            jb->out[8] += en[i] * w;
            jb->out[9] += (uint64_t)rv[i] * w;
        }
        jb->out[1] += c;
    }
    free(buf); free(h); free(st); free(en); free(rv); free(scratch);
    return NULL;
}

static void synth_checksums_impl(uint64_t seed, const uint64_t *off, uint64_t n_reads, uint64_t read_len, unsigned l,
                                 unsigned k, double density, int mode, int threads, uint64_t out[S2K_NSUM], int hifi);

void s2k_oracle_synth_checksums(uint64_t seed, uint64_t n_reads, uint64_t read_len, unsigned l, unsigned k,
                                double density, int mode, int threads, uint64_t out[S2K_NSUM]) {
    synth_checksums_impl(seed, NULL, n_reads, read_len, l, k, density, mode, threads, out, 0);
}

void s2k_oracle_synth_checksums_off(uint64_t seed, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                                    double density, int mode, int threads, uint64_t out[S2K_NSUM]) {
    synth_checksums_impl(seed, off, n_reads, 0, l, k, density, mode, threads, out, 0);
}

/* HiFi-like reads: read r = s2k_oracle_hifi_read(seed, r, off[r+1] - off[r]) (off = prefix of s2k_oracle_hifi_len) */
void s2k_oracle_hifi_checksums(uint64_t seed, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                               double density, int mode, int threads, uint64_t out[S2K_NSUM]) {
    synth_checksums_impl(seed, off, n_reads, 0, l, k, density, mode, threads, out, 1);
}

static void synth_checksums_impl(uint64_t seed, const uint64_t *off, uint64_t n_reads, uint64_t read_len, unsigned l,
                                 unsigned k, double density, int mode, int threads, uint64_t out[S2K_NSUM], int hifi) {
    tables();
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    struct sum_job *jobs = (struct sum_job *)calloc(threads, sizeof(*jobs));
    pthread_t *th = (pthread_t *)malloc(sizeof(*th) * threads);
    for (int t = 0; t < threads; t++) {
        jobs[t].seed = seed;
        jobs[t].off = off;
        jobs[t].r0 = n_reads * (uint64_t)t / threads;
        jobs[t].r1 = n_reads * (uint64_t)(t + 1) / threads;
        jobs[t].read_len = read_len;
        jobs[t].hifi = hifi;
        jobs[t].l = l;
        jobs[t].k = k;
        jobs[t].bound = s2k_oracle_hash_bound(density);
        jobs[t].mode = mode;
        pthread_create(&th[t], NULL, sum_worker, &jobs[t]);
    }
    memset(out, 0, S2K_NSUM * sizeof(uint64_t));
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < threads; t++) { /* in read order: out[1] is the number of items before this thread's first one */
        const struct sum_job *jb = &jobs[t];
        const uint64_t g0 = out[1];
        /* SUM x (2 (g0 + i) + 1) = SUM x (2 i + 1) + 2 g0 SUM x */
        out[6] += jb->out[6] + 2 * g0 * jb->sum_hash;
        out[7] += jb->out[7] + 2 * g0 * jb->out[3];
        out[8] += jb->out[8] + 2 * g0 * jb->out[4];
        out[9] += jb->out[9] + 2 * g0 * jb->out[5];
        /* SUM_r (g0 + local_off[r]) (2 r + 1), SUM_{r0 <= r < r1} (2 r + 1) = r1^2 - r0^2 */
        out[10] += jb->out[10] + g0 * (jb->r1 * jb->r1 - jb->r0 * jb->r0);
        out[11] += jb->out[11];
        out[0] += jb->out[0];
        out[1] += jb->out[1];
        out[2] ^= jb->out[2];
        for (int i = 3; i < 6; i++) out[i] += jb->out[i];
    }
    out[10] += out[1] * (2 * n_reads + 1); /* the end entry km_off[n_reads] */
    free(jobs);
    free(th);
}

uint64_t s2k_oracle_batch_minimizers(const uint8_t *bases, const uint64_t *off, uint64_t n_reads, unsigned l,
                                     double density, int mode, uint64_t *mn_off, uint32_t *j, uint32_t *jend,
                                     uint32_t *hash, uint64_t cap) {
    tables();
    uint32_t bound = s2k_oracle_hash_bound(density);
    uint64_t acc = 0;
    uint64_t *tj = NULL, *tje = NULL;
    uint32_t *th = NULL;
    size_t tcap = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint8_t *s = bases + off[r];
        size_t n = (size_t)(off[r + 1] - off[r]);
        size_t M = s2k_oracle_minimizers(s, n, l, bound, mode, NULL, NULL, NULL, 0);
        if (mn_off) mn_off[r] = acc;
        if (M && (j || jend || hash)) {
            if (tcap < M) {
                free(tj);
                free(tje);
                free(th);
                tcap = M * 2;
                tj = (uint64_t *)malloc(tcap * 8);
                tje = (uint64_t *)malloc(tcap * 8);
                th = (uint32_t *)malloc(tcap * 4);
            }
            s2k_oracle_minimizers(s, n, l, bound, mode, tj, tje, th, M);
            for (size_t i = 0; i < M && acc + i < cap; i++) {
                if (j) j[acc + i] = (uint32_t)tj[i];
                if (jend) jend[acc + i] = (uint32_t)tje[i];
                if (hash) hash[acc + i] = th[i];
            }
        }
        acc += M;
    }
    if (mn_off) mn_off[n_reads] = acc;
    free(tj);
    free(tje);
    free(th);
    return acc;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic bases: 32 bases per splitmix64 output, keyed by (seed, absolute 32-base block).
 * Mirrors benches/bench.rs:19-31 (uniform random ACGT); the device generator in the product
 * library implements the identical function.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void s2k_oracle_synth_bases(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *out) {
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t i = 0; i < n; i++) {
        uint64_t q = first_base + i;
        uint64_t z = splitmix64(seed ^ ((q >> 5) * 0xD6E8FEB86659FD93ULL));
        out[i] = (uint8_t)ACGT[(z >> (2 * (q & 31))) & 3];
    }
}
