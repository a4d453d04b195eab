/*
 * s2k_oracle_avx512.c -- AVX-512 CPU baseline for the reference's fast path (HashMode::Simd / HpcSimd).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see s2k_oracle.h).  It exists because the reference's own quick
 * path is AVX-512 (src/nthash_avx512_32.rs, src/hpc.rs:44-147; README.md:23 "around 1 GB/sec") and BASELINE.md
 * asks for that throughput to be timed beside the GPU numbers.  It is a restatement of the *semantics*
 * (strict '<', f32-recomputed bound, low-nibble seed map, dropped final 16-block when #l-mers % 16 == 0 --
 * SURVEY.md 8a traps i, ii, v, vi), NOT of the reference's code: the reference rolls 16 lanes with a
 * Hillis-Steele rotate-xor scan over (in ^ rot(out)) terms plus a lane-15 carry
 * (src/nthash_avx512_32.rs:367-420); here nothing rolls -- block hashes of 1, 2, 4, 8, 16 bases are built by
 * doubling and combined per the binary digits of l ("ntHash needs no rolling").
 * Results are checked against the scalar oracle's Simd-mode output in tests/test_oracle_avx512.py.
 *
 * Variant 1, "reference shape" (round 5): the reference's OWN algorithm restated from its behaviour, for the timing north_star and SURVEY.md 8d
 * name -- 16 l-mers per step: the first 16 by l rotate-xor steps over 16 staggered windows (src/nthash_avx512_32.rs:281-341), every later
 * block by per-lane in ^ rot(out) terms, a four-step rotate-xor prefix scan over the 16 lanes and the previous block's lane 15 carried in,
 * rotated by 1..16 (:348-428, :432-509); bases -> 3-bit codes by the low nibble, codes -> seeds by a 16-entry permute (:178-277); strict
 * '<' mask and compress-stores of hashes and positions into a 16-entry scratch that the iterator drains (:117-151); HpcSimd first compresses the
 * whole read 16 bases per step, widening to 32-bit lanes to compress (src/hpc.rs:74-115), run starts included as the reference always
 * stores them.  Same result semantics as variant 0, checked against the scalar oracle in the same test.
 *
 * Build: gcc -O3 -mavx512f -mavx512bw -mavx512vl -mavx512vbmi2 (the loader adds the flags; the entry points
 * report "unsupported" at run time when the CPU lacks them).
 */
#define _GNU_SOURCE /* pthread_barrier_t, clock_gettime under -std=c11 */
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

uint32_t s2k_oracle_hash_bound(double density);
uint32_t s2k_oracle_hash_bound_simd(uint32_t hash_bound);

int s2k_avx512_supported(void) {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
           __builtin_cpu_supports("avx512vbmi2");
}

#define SEED_A 0x95c60474u
#define SEED_C 0x62a02b4cu
#define SEED_G 0x82572324u
#define SEED_T 0x4be24456u

/* 16 bases -> 16 seeds: index = low nibble (1->A 3->C 7->G 4->T, else 0), src/nthash_avx512_32.rs:178-193,225-262 */
static inline __m512i seeds16(const uint8_t *p, __m512i table) {
    __m512i idx = _mm512_cvtepu8_epi32(_mm_loadu_si128((const __m128i *)p));
    return _mm512_permutexvar_epi32(_mm512_and_si512(idx, _mm512_set1_epi32(15)), table);
}

/* lanes [sh, sh+16) of the concatenation (lo, hi) */
#define SHIFTED(lo, hi, sh) _mm512_alignr_epi32((hi), (lo), (sh))

/* Canonical hashes of the 16 l-mers starting at p[0..15]; reads p[0 .. 16+l-2] (caller guarantees padding). */
static inline __m512i hash16(const uint8_t *p, unsigned l, __m512i tf, __m512i tr) {
    /* level-1 vectors: seeds of bases [0,16), [16,32), [32,48) */
    __m512i f1[3], r1[3];
    for (int v = 0; v < 3; v++) {
        f1[v] = seeds16(p + 16 * v, tf);
        r1[v] = seeds16(p + 16 * v, tr);
    }
    /* doubling: A_2m[q] = rotl(A_m[q], m) ^ A_m[q+m] ;  B_2m[q] = B_m[q] ^ rotl(B_m[q+m], m)
     * (A_m[q] = XOR_{i<m} rotl(h[q+i], m-1-i),  B_m[q] = XOR_{i<m} rotl(rc[q+i], i)).  Level m is needed for
     * start offsets up to 16 + (31 - m): two vectors (offsets 0..31) suffice for every level once the third
     * level-1 vector exists. */
    __m512i f2[3], r2[3], f4[3], r4[3], f8[2], r8[2], f16[1], r16[1];
    const __m512i z = _mm512_setzero_si512();
    for (int v = 0; v < 3; v++) {
        __m512i nf = SHIFTED(f1[v], v < 2 ? f1[v + 1] : z, 1), nr = SHIFTED(r1[v], v < 2 ? r1[v + 1] : z, 1);
        f2[v] = _mm512_xor_si512(_mm512_rol_epi32(f1[v], 1), nf);
        r2[v] = _mm512_xor_si512(r1[v], _mm512_rol_epi32(nr, 1));
    }
    for (int v = 0; v < 3; v++) {
        __m512i nf = SHIFTED(f2[v], v < 2 ? f2[v + 1] : z, 2), nr = SHIFTED(r2[v], v < 2 ? r2[v + 1] : z, 2);
        f4[v] = _mm512_xor_si512(_mm512_rol_epi32(f2[v], 2), nf);
        r4[v] = _mm512_xor_si512(r2[v], _mm512_rol_epi32(nr, 2));
    }
    for (int v = 0; v < 2; v++) {
        __m512i nf = SHIFTED(f4[v], f4[v + 1], 4), nr = SHIFTED(r4[v], r4[v + 1], 4);
        f8[v] = _mm512_xor_si512(_mm512_rol_epi32(f4[v], 4), nf);
        r8[v] = _mm512_xor_si512(r4[v], _mm512_rol_epi32(nr, 4));
    }
    {
        __m512i nf = SHIFTED(f8[0], f8[1], 8), nr = SHIFTED(r8[0], r8[1], 8);
        f16[0] = _mm512_xor_si512(_mm512_rol_epi32(f8[0], 8), nf);
        r16[0] = _mm512_xor_si512(r8[0], _mm512_rol_epi32(nr, 8));
    }
    /* combine blocks by the binary digits of l, most significant first; `done` bases are already covered */
    __m512i fh = z, rh = z;
    unsigned done = 0;
#define BLOCK(M, FA, RA, NV)                                                                          \
    if (l & (M)) {                                                                                    \
        const unsigned after = l - done - (M); /* bases after this block */                           \
        const int v = (int)(done >> 4), sh = (int)(done & 15);                                        \
        __m512i bf, br;                                                                               \
        switch (sh) { /* alignr needs an immediate */                                                 \
        default: bf = FA[v]; br = RA[v]; break;                                                       \
        case 8:  bf = SHIFTED(FA[v], v + 1 < (NV) ? FA[v + 1] : z, 8);  br = SHIFTED(RA[v], v + 1 < (NV) ? RA[v + 1] : z, 8);  break; \
        case 12: bf = SHIFTED(FA[v], v + 1 < (NV) ? FA[v + 1] : z, 12); br = SHIFTED(RA[v], v + 1 < (NV) ? RA[v + 1] : z, 12); break; \
        case 14: bf = SHIFTED(FA[v], v + 1 < (NV) ? FA[v + 1] : z, 14); br = SHIFTED(RA[v], v + 1 < (NV) ? RA[v + 1] : z, 14); break; \
        case 4:  bf = SHIFTED(FA[v], v + 1 < (NV) ? FA[v + 1] : z, 4);  br = SHIFTED(RA[v], v + 1 < (NV) ? RA[v + 1] : z, 4);  break; \
        case 6:  bf = SHIFTED(FA[v], v + 1 < (NV) ? FA[v + 1] : z, 6);  br = SHIFTED(RA[v], v + 1 < (NV) ? RA[v + 1] : z, 6);  break; \
        case 2:  bf = SHIFTED(FA[v], v + 1 < (NV) ? FA[v + 1] : z, 2);  br = SHIFTED(RA[v], v + 1 < (NV) ? RA[v + 1] : z, 2);  break; \
        case 10: bf = SHIFTED(FA[v], v + 1 < (NV) ? FA[v + 1] : z, 10); br = SHIFTED(RA[v], v + 1 < (NV) ? RA[v + 1] : z, 10); break; \
        }                                                                                             \
        fh = _mm512_xor_si512(fh, _mm512_rolv_epi32(bf, _mm512_set1_epi32((int)after)));              \
        rh = _mm512_xor_si512(rh, _mm512_rolv_epi32(br, _mm512_set1_epi32((int)done)));               \
        done += (M);                                                                                  \
    }
    BLOCK(16, f16, r16, 1)
    BLOCK(8, f8, r8, 2)
    BLOCK(4, f4, r4, 3)
    BLOCK(2, f2, r2, 3)
    BLOCK(1, f1, r1, 3)
#undef BLOCK
    return _mm512_min_epu32(fh, rh);
}

/* Minimizers of t[0..m) with the Simd iterator's result semantics.  pos/hash may be NULL (count only).
 * t must be readable up to t[m + 48) (callers pad). Returns the count. */
static size_t simd_scan_avx512(const uint8_t *t, size_t m, unsigned l, uint32_t bound, uint32_t *pos, uint32_t *hash, size_t cap) {
    if (l == 0 || l > 31 || m < l) return 0;
    const uint32_t b2 = s2k_oracle_hash_bound_simd(bound); /* f32 re-derivation, src/nthash_avx512_32.rs:46-48 */
    const size_t sentinel = m - l + 1;
    size_t limit = sentinel;
    if (sentinel % 16 == 0 && sentinel >= 32) limit = sentinel - 16; /* tail-mask quirk, :134-138 */
    const __m512i tf = _mm512_set_epi32(0, 0, 0, 0, 0, 0, 0, 0, (int)SEED_G, 0, 0, (int)SEED_T, (int)SEED_C, 0, (int)SEED_A, 0);
    const __m512i tr = _mm512_set_epi32(0, 0, 0, 0, 0, 0, 0, 0, (int)SEED_C, 0, 0, (int)SEED_A, (int)SEED_G, 0, (int)SEED_T, 0);
    const __m512i vb = _mm512_set1_epi32((int)b2);
    const __m512i lane = _mm512_set_epi32(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    size_t cnt = 0;
    for (size_t p = 0; p < limit; p += 16) {
        __m512i hv = hash16(t + p, l, tf, tr);
        __mmask16 k = _mm512_cmplt_epu32_mask(hv, vb); /* strict '<', :55,:130 */
        size_t left = limit - p;
        if (left < 16) k &= (__mmask16)((1u << left) - 1u);
        if (!k) continue;
        unsigned n = (unsigned)__builtin_popcount(k);
        if (pos && cnt + n <= cap) {
            _mm512_mask_compressstoreu_epi32(pos + cnt, k, _mm512_add_epi32(lane, _mm512_set1_epi32((int)p)));
            _mm512_mask_compressstoreu_epi32(hash + cnt, k, hv);
        }
        cnt += n;
    }
    return cnt;
}

/* Homopolymer compression, 64 bases per step (semantics of encode_rle_simd, src/hpc.rs:44-147: keep s[i] iff
 * i == 0 or s[i] != s[i-1]).  out must hold n + 64 bytes; st (run starts) may be NULL. Returns the run count. */
static size_t hpc_avx512(const uint8_t *s, size_t n, uint8_t *out, uint32_t *st) {
    size_t r = 0, i = 0;
    if (n == 0) return 0;
    out[r] = s[0];
    if (st) st[r] = 0;
    r++;
    i = 1;
    const __m512i iota = _mm512_set_epi32(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    for (; i + 64 <= n; i += 64) {
        __m512i cur = _mm512_loadu_si512((const void *)(s + i)), prv = _mm512_loadu_si512((const void *)(s + i - 1));
        __mmask64 k = _mm512_cmpneq_epu8_mask(cur, prv);
        _mm512_mask_compressstoreu_epi8(out + r, k, cur);
        if (st) {
            size_t rr = r;
            for (int q = 0; q < 4; q++) {
                __mmask16 kq = (__mmask16)(k >> (16 * q));
                _mm512_mask_compressstoreu_epi32(st + rr, kq, _mm512_add_epi32(iota, _mm512_set1_epi32((int)(i + 16 * q))));
                rr += (size_t)__builtin_popcount(kq);
            }
        }
        r += (size_t)__builtin_popcountll(k);
    }
    for (; i < n; i++)
        if (s[i] != s[i - 1]) {
            out[r] = s[i];
            if (st) st[r] = (uint32_t)i;
            r++;
        }
    return r;
}

/* ---- variant 1: the reference's algorithm shape ------------------------------------------------------------------------------- */
static inline __m512i codes16(const uint8_t *p) { /* 16 bases -> 3-bit codes in 32-bit lanes: low nibble 1 -> 0 (A), 3 -> 1 (C), 7 -> 2 (G), 4 -> 3 (T), else 4 */
    const __m128i table = _mm_set_epi8(4, 4, 4, 4, 4, 4, 4, 4, 2, 4, 4, 3, 1, 4, 0, 4);
    const __m128i v = _mm_loadu_si128((const __m128i *)p);
    return _mm512_cvtepu8_epi32(_mm_shuffle_epi8(table, _mm_and_si128(v, _mm_set1_epi8(0x0f))));
}
#define SHUP(M, V) _mm512_maskz_expand_epi32((M), (V)) /* lanes move up by the number of zero bits at the low end of M */
/* t must be readable up to t[m + 32 + l) (callers pad).  Emits through the 16-entry scratch like the iterator: returns the count. */
static size_t simd_scan_refshape(const uint8_t *t, size_t m, unsigned l, uint32_t bound, uint32_t *pos, uint32_t *hash, size_t cap) {
    if (l == 0 || l > 31 || m < l) return 0;
    const uint32_t b2 = s2k_oracle_hash_bound_simd(bound);
    const size_t sentinel = m - l + 1;
    const __m512i seedf = _mm512_set_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (int)SEED_T, (int)SEED_G, (int)SEED_C, (int)SEED_A);
    const __m512i seedr = _mm512_set_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, (int)SEED_A, (int)SEED_C, (int)SEED_G, (int)SEED_T);
    const __m512i vb = _mm512_set1_epi32((int)b2), v16 = _mm512_set1_epi32(16), lane15 = _mm512_set1_epi32(15);
    const __m512i shift = _mm512_set_epi32(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m512i vl = _mm512_set1_epi32((int)l), vlm = _mm512_set1_epi32((int)l - 1), vck = _mm512_set1_epi32((int)(32 - (l % 32)));
    __m512i positions = _mm512_set_epi32(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    uint32_t sh[16], sp[16]; /* the iterator's scratch */
    size_t cnt = 0;
    /* first 16 l-mers: l steps over 16 staggered windows */
    __m512i fh = _mm512_setzero_si512(), rh = _mm512_setzero_si512();
    for (unsigned i = 0; i < l; i++) {
        const __m512i c = codes16(t + i);
        fh = _mm512_xor_si512(_mm512_rol_epi32(fh, 1), _mm512_permutexvar_epi32(c, seedf));
        rh = _mm512_ror_epi32(_mm512_xor_si512(rh, _mm512_rorv_epi32(_mm512_permutexvar_epi32(c, seedr), vck)), 1);
    }
    __m512i hv = _mm512_mask_blend_epi32(_mm512_cmpgt_epu32_mask(fh, rh), fh, rh);
    {
        __mmask16 k = _mm512_cmplt_epu32_mask(hv, vb);
        _mm512_mask_compressstoreu_epi32(sh, k, hv);
        _mm512_mask_compressstoreu_epi32(sp, k, positions);
        positions = _mm512_add_epi32(positions, v16);
        const unsigned n = (unsigned)__builtin_popcount(k);
        for (unsigned q = 0; q < n; q++) { /* the drain: an entry at or behind the sentinel ends the iteration */
            if (sp[q] >= sentinel) return cnt;
            if (pos && cnt < cap) {
                pos[cnt] = sp[q];
                hash[cnt] = sh[q];
            }
            cnt++;
        }
    }
    for (size_t i = 16; i < sentinel;) {
        const __m512i cin = codes16(t + i - 1 + l), cout = codes16(t + i - 1);
        /* forward strand */
        __m512i x = _mm512_xor_si512(_mm512_permutexvar_epi32(cin, seedf), _mm512_rolv_epi32(_mm512_permutexvar_epi32(cout, seedf), vl));
        x = _mm512_xor_si512(x, SHUP(0xfffe, _mm512_rol_epi32(x, 1)));
        x = _mm512_xor_si512(x, SHUP(0xfffc, _mm512_rol_epi32(x, 2)));
        x = _mm512_xor_si512(x, SHUP(0xfff0, _mm512_rol_epi32(x, 4)));
        x = _mm512_xor_si512(x, SHUP(0xff00, _mm512_rol_epi32(x, 8)));
        fh = _mm512_xor_si512(_mm512_rolv_epi32(_mm512_permutexvar_epi32(lane15, fh), shift), x);
        /* reverse strand */
        __m512i y = _mm512_xor_si512(_mm512_rolv_epi32(_mm512_permutexvar_epi32(cin, seedr), vlm), _mm512_ror_epi32(_mm512_permutexvar_epi32(cout, seedr), 1));
        y = _mm512_xor_si512(y, SHUP(0xfffe, _mm512_ror_epi32(y, 1)));
        y = _mm512_xor_si512(y, SHUP(0xfffc, _mm512_ror_epi32(y, 2)));
        y = _mm512_xor_si512(y, SHUP(0xfff0, _mm512_ror_epi32(y, 4)));
        y = _mm512_xor_si512(y, SHUP(0xff00, _mm512_ror_epi32(y, 8)));
        rh = _mm512_xor_si512(_mm512_rorv_epi32(_mm512_permutexvar_epi32(lane15, rh), shift), y);
        hv = _mm512_mask_blend_epi32(_mm512_cmpgt_epu32_mask(fh, rh), fh, rh);
        i += 16;
        __mmask16 k = _mm512_cmplt_epu32_mask(hv, vb);
        if (k) {
            if (i >= sentinel) k &= (__mmask16)((1u << (sentinel % 16)) - 1u); /* 0 when the l-mer count is a multiple of 16: the block is lost (:134-138) */
            _mm512_mask_compressstoreu_epi32(sh, k, hv);
            _mm512_mask_compressstoreu_epi32(sp, k, positions);
        }
        positions = _mm512_add_epi32(positions, v16);
        const unsigned n = (unsigned)__builtin_popcount(k);
        for (unsigned q = 0; q < n; q++) {
            if (sp[q] >= sentinel) return cnt;
            if (pos && cnt < cap) {
                pos[cnt] = sp[q];
                hash[cnt] = sh[q];
            }
            cnt++;
        }
    }
    return cnt;
}
#undef SHUP

/* Homopolymer compression 16 bases per step, run starts always stored (out: n + 64 bytes, st: n + 64 entries). */
static size_t hpc_refshape(const uint8_t *s, size_t n, uint8_t *out, uint32_t *st) {
    size_t r = 0;
    __m512i positions = _mm512_set_epi32(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m512i v16 = _mm512_set1_epi32(16);
    const size_t blocks = (n + 15) / 16;
    for (size_t i = 0; i < blocks; i++) {
        const __m128i v = _mm_loadu_si128((const __m128i *)(s + 16 * i)); /* (the caller pads the read: the reference reads past its end here) */
        __mmask16 k = (__mmask16)(~_mm_cmpeq_epi8_mask(v, _mm_slli_si128(v, 1)) & 0xFFFE);
        k |= (__mmask16)(i == 0 || s[16 * i] != s[16 * i - 1]);
        if (16 * i + 16 > n) k &= (__mmask16)((1u << (n & 15)) - 1u);
        const __m512i packed = _mm512_maskz_compress_epi32(k, _mm512_cvtepi8_epi32(v));
        const unsigned c = (unsigned)__builtin_popcount(k);
        _mm_mask_storeu_epi8(out + r, (__mmask16)((1u << c) - 1u), _mm512_cvtepi32_epi8(packed));
        _mm512_mask_compressstoreu_epi32(st + r, k, positions);
        positions = _mm512_add_epi32(positions, v16);
        r += c;
    }
    return r;
}

static int g_dummy_variant_guard; /* (keeps the two variants in one translation unit: same flags, same harness) */

/* One read, Simd (hpc = 0) or HpcSimd (hpc = 1) semantics: minimizer triples like s2k_oracle_minimizers.
 * scratch: caller-provided buffer of at least n + 128 bytes (+ 4 n bytes for run starts when hpc && j). */
size_t s2k_avx512_minimizers_v(const uint8_t *s, size_t n, unsigned l, uint32_t bound, int hpc, uint32_t *j, uint32_t *jend,
                               uint32_t *hash, size_t cap, int variant) {
    (void)g_dummy_variant_guard;
    if (!s2k_avx512_supported() || n <= l || l == 0 || l > 31) return 0; /* src/lib.rs:97 */
    uint8_t *buf = (uint8_t *)malloc(n + 128);
    uint32_t *st = NULL;
    size_t m = n;
    if (hpc) {
        if (j || variant) st = (uint32_t *)malloc((n + 64) * sizeof(uint32_t));
        if (variant) {
            uint8_t *padded = (uint8_t *)malloc(n + 32); /* (16-byte loads up to the end of the last block) */
            memcpy(padded, s, n);
            memset(padded + n, 0, 32);
            m = hpc_refshape(padded, n, buf, st);
            free(padded);
        } else {
            m = hpc_avx512(s, n, buf, st);
        }
    } else {
        memcpy(buf, s, n);
    }
    memset(buf + m, 0, 112); /* hash16 reads up to 47 bytes past the last l-mer start */
    size_t cnt = variant ? simd_scan_refshape(buf, m, l, bound, j, hash, cap) : simd_scan_avx512(buf, m, l, bound, j, hash, cap);
    if (j) {
        size_t lim = cnt < cap ? cnt : cap;
        for (size_t i = 0; i < lim; i++) {
            uint32_t p = j[i];
            if (hpc) { /* (hpc_pos[p], hpc_pos[p+l-1]) -- start of the last run, src/nthash_hpc_simd.rs:64 */
                j[i] = st[p];
                if (jend) jend[i] = st[p + l - 1];
            } else if (jend) {
                jend[i] = p + l - 1; /* src/lib.rs:202 */
            }
        }
    }
    free(buf);
    free(st);
    return cnt;
}
size_t s2k_avx512_minimizers(const uint8_t *s, size_t n, unsigned l, uint32_t bound, int hpc, uint32_t *j, uint32_t *jend,
                             uint32_t *hash, size_t cap) {
    return s2k_avx512_minimizers_v(s, n, l, bound, hpc, j, jend, hash, cap, 0);
}

/* Batch count-only pass for timing (what src/main.rs:65-76 does per read, with HashMode::Simd / HpcSimd):
 * returns the total number of k-min-mers.  Single thread; the caller runs one per shard. */
static uint64_t batch_count_v(const uint8_t *bases, const uint64_t *off, uint64_t r0, uint64_t r1, unsigned l, unsigned k,
                              double density, int hpc, int variant, uint64_t n_bases_total) {
    if (!s2k_avx512_supported()) return 0;
    const uint32_t bound = s2k_oracle_hash_bound(density);
    uint64_t total = 0;
    size_t cap = 0;
    uint8_t *buf = NULL;
    uint32_t *st = NULL;
    for (uint64_t r = r0; r < r1; r++) {
        const uint8_t *s = bases + off[r];
        size_t n = (size_t)(off[r + 1] - off[r]);
        if (n <= l) continue;
        if (cap < n + 128) {
            free(buf);
            free(st);
            cap = (n + 128) * 2;
            buf = (uint8_t *)malloc(cap);
            st = variant ? (uint32_t *)malloc(cap * sizeof(uint32_t)) : NULL;
        }
        size_t m = n, M;
        if (variant) {
            /* the reference works in place on the read (HpcSimd: on the compressed copy it makes, positions and all); a read that ends
             * within 16 + l bytes of the end of the whole stream is copied so that the 16-byte loads stay inside memory we own */
            const int near_end = off[r + 1] + 64 > n_bases_total;
            if (hpc) {
                if (near_end) {
                    uint8_t *padded = (uint8_t *)malloc(n + 32);
                    memcpy(padded, s, n);
                    memset(padded + n, 0, 32);
                    m = hpc_refshape(padded, n, buf, st);
                    free(padded);
                } else {
                    m = hpc_refshape(s, n, buf, st);
                }
                memset(buf + m, 0, 64);
                M = simd_scan_refshape(buf, m, l, bound, NULL, NULL, 0);
            } else if (near_end) {
                memcpy(buf, s, n);
                memset(buf + n, 0, 64);
                M = simd_scan_refshape(buf, n, l, bound, NULL, NULL, 0);
            } else {
                M = simd_scan_refshape(s, n, l, bound, NULL, NULL, 0); /* reads into the next read's bases, as the reference reads past its slice */
            }
        } else {
            if (hpc) m = hpc_avx512(s, n, buf, NULL);
            else memcpy(buf, s, n);
            memset(buf + m, 0, 112);
            M = simd_scan_avx512(buf, m, l, bound, NULL, NULL, 0);
        }
        if (M >= k) total += M - k + 1;
    }
    free(buf);
    free(st);
    return total;
}
uint64_t s2k_avx512_batch_count(const uint8_t *bases, const uint64_t *off, uint64_t r0, uint64_t r1, unsigned l, unsigned k,
                                double density, int hpc) {
    return batch_count_v(bases, off, r0, r1, l, k, density, hpc, 0, 0);
}

/* threads over contiguous read shards (the shape of parallel_fastx workers, src/main.rs:79) */
#include <pthread.h>
struct avx_job {
    const uint8_t *bases;
    const uint64_t *off;
    uint64_t r0, r1;
    unsigned l, k;
    double density;
    int hpc;
    uint64_t total;
    int variant;
    uint64_t n_bases_total;
};
static void *avx_worker(void *p) {
    struct avx_job *j = (struct avx_job *)p;
    j->total = batch_count_v(j->bases, j->off, j->r0, j->r1, j->l, j->k, j->density, j->hpc, j->variant, j->n_bases_total);
    return NULL;
}
uint64_t s2k_avx512_batch_count_mt_v(const uint8_t *bases, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                                     double density, int hpc, int threads, int variant) {
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    struct avx_job *jobs = (struct avx_job *)malloc(sizeof(*jobs) * (size_t)threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(*th) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (struct avx_job){bases, off, n_reads * (uint64_t)t / (uint64_t)threads, n_reads * (uint64_t)(t + 1) / (uint64_t)threads,
                                   l, k, density, hpc, 0, variant, n_reads ? off[n_reads] : 0};
        pthread_create(&th[t], NULL, avx_worker, &jobs[t]);
    }
    uint64_t tot = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        tot += jobs[t].total;
    }
    free(jobs);
    free(th);
    return tot;
}

uint64_t s2k_avx512_batch_count_mt(const uint8_t *bases, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                                   double density, int hpc, int threads) {
    return s2k_avx512_batch_count_mt_v(bases, off, n_reads, l, k, density, hpc, threads, 0);
}

/* Timed variant for bench.py (see s2k_oracle_batch_count_timed): threads first, warm-up, barrier, `repeats` passes. */
#include <time.h>
struct avx_tjob {
    struct avx_job j;
    int repeats;
    pthread_barrier_t *bar;
};
static void *avx_timed_worker(void *p) {
    struct avx_tjob *t = (struct avx_tjob *)p;
    uint64_t w1 = t->j.r0 + 8 < t->j.r1 ? t->j.r0 + 8 : t->j.r1;
    (void)batch_count_v(t->j.bases, t->j.off, t->j.r0, w1, t->j.l, t->j.k, t->j.density, t->j.hpc, t->j.variant, t->j.n_bases_total);
    pthread_barrier_wait(t->bar);
    for (int i = 0; i < t->repeats; i++)
        t->j.total = batch_count_v(t->j.bases, t->j.off, t->j.r0, t->j.r1, t->j.l, t->j.k, t->j.density, t->j.hpc, t->j.variant, t->j.n_bases_total);
    return NULL;
}
uint64_t s2k_avx512_batch_count_timed_v(const uint8_t *bases, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                                        double density, int hpc, int threads, int repeats, double *seconds, int variant) {
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_reads) threads = n_reads ? (int)n_reads : 1;
    if (repeats < 1) repeats = 1;
    struct avx_tjob *jobs = (struct avx_tjob *)calloc((size_t)threads, sizeof(*jobs));
    pthread_t *th = (pthread_t *)malloc(sizeof(*th) * (size_t)threads);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads + 1);
    for (int t = 0; t < threads; t++) {
        jobs[t].j = (struct avx_job){bases, off, n_reads * (uint64_t)t / (uint64_t)threads, n_reads * (uint64_t)(t + 1) / (uint64_t)threads,
                                     l, k, density, hpc, 0, variant, n_reads ? off[n_reads] : 0};
        jobs[t].repeats = repeats;
        jobs[t].bar = &bar;
        pthread_create(&th[t], NULL, avx_timed_worker, &jobs[t]);
    }
    struct timespec a, b;
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &a);
    uint64_t tot = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        tot += jobs[t].j.total;
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (seconds) *seconds = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
    pthread_barrier_destroy(&bar);
    free(jobs);
    free(th);
    return tot;
}
uint64_t s2k_avx512_batch_count_timed(const uint8_t *bases, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                                      double density, int hpc, int threads, int repeats, double *seconds) {
    return s2k_avx512_batch_count_timed_v(bases, off, n_reads, l, k, density, hpc, threads, repeats, seconds, 0);
}
