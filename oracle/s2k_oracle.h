/*
 * s2k_oracle.h -- CPU restatement of the reference's seq -> k-min-mer path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  The restatement reproduces every known-answer vector
 * the reference holds for this path (see oracle/README.md and
 * tests/test_oracle_golden.py):
 *   G1  tests/main.rs:41-57,60-73   Regular, l=10 k=5 d=1e-4 -> 15 k-min-mer hashes
 *   G2  src/old/nthash_hpc.rs.orig:68-77  HPC iterator (H=u64), l=4 d=0.1 -> 6 (pos,hash)
 *   G3  tests/main.rs:76-78         encode_rle == hpc == encode_rle_simd (+ run starts)
 *   G4  tests/main.rs:82-89         18-cell grid Regular==Simd, Hpc==HpcSimd (hash-only)
 * The reference itself (nightly Rust + un-vendored git crates) cannot be built
 * in this image, so there is no oracle/_ref.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef S2K_ORACLE_H
#define S2K_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* HashMode, src/lib.rs:21-27 (same numbering as include/s2k.h) */
enum { S2K_O_REGULAR = 0, S2K_O_HPC = 1, S2K_O_SIMD = 2, S2K_O_HPCSIMD = 3 };

/* hash_bound = (density * u32::MAX as f64) as u32 -- src/lib.rs:91 (saturating cast). */
uint32_t s2k_oracle_hash_bound(double density);
/* bound as recomputed through f32 inside NtHashSIMDIterator::new -- src/nthash_avx512_32.rs:46-48 */
uint32_t s2k_oracle_hash_bound_simd(uint32_t hash_bound);

/* seed tables, src/nthash_hpc.rs:30-49 */
uint32_t s2k_oracle_seed_h(uint8_t c);
uint32_t s2k_oracle_seed_rc(uint8_t c);

/* xorshift mix of a u32 minimizer hash -- src/lib.rs:157-169 */
uint64_t s2k_oracle_mix32(uint32_t h);

/* Canonical 32-bit ntHash1 of every l-mer of s[0..n) (closed form of
 * src/nthash_hpc.rs:144,168,245-249).  out must hold n-l+1 values. Returns count. */
size_t s2k_oracle_nthash32_all(const uint8_t *s, size_t n, unsigned l, uint32_t *out);

/* Standalone HPC: src/hpc.rs:28-41 (hpc: any equal bytes), src/hpc.rs:7-25 (encode_rle:
 * collapses only "ACTGactgNn"), src/hpc.rs:44-147 (encode_rle_simd: any equal bytes).
 * out/pos must hold n entries.  Return the number of runs. `which`: 0=hpc 1=encode_rle 2=encode_rle_simd */
size_t s2k_oracle_hpc(const uint8_t *s, size_t n, int which, uint8_t *out, uint64_t *pos);

/* Minimizer triples (j, jend, hash) in iterator order for one read.
 *   mode REGULAR : src/lib.rs:215-230 (+ ntHash1 definition)
 *   mode HPC     : closed form of NtHashHPCIterator, src/nthash_hpc.rs:115-283
 *   mode SIMD    : NtHashSIMDIterator semantics, src/nthash_avx512_32.rs:32-164 (+ lib.rs:198-204)
 *   mode HPCSIMD : NtHashHPCSIMDIterator semantics, src/nthash_hpc_simd.rs:35-68
 * `bound` is the u32 hash_bound of src/lib.rs:91 (SIMD modes re-derive their f32 bound from it).
 * Arrays may be NULL (count only); at most cap entries are written. Returns the count. */
size_t s2k_oracle_minimizers(const uint8_t *s, size_t n, unsigned l, uint32_t bound, int mode,
                             uint64_t *j, uint64_t *jend, uint32_t *hash, size_t cap);

/* Literal transliteration of the NtHashHPCIterator state machine (rings, first_element
 * special case, end-of-sequence return) -- src/nthash_hpc.rs:115-283.  Used only to prove
 * the closed form in s2k_oracle_minimizers(mode HPC). Requires l < 256, n >= l. */
size_t s2k_oracle_hpc_literal(const uint8_t *s, size_t n, unsigned l, uint32_t bound,
                              uint64_t *j, uint64_t *jend, uint32_t *hash, size_t cap);

/* Same state machine with H = u64 (full 64-bit ntHash seeds, rotations mod 64), to check the
 * archived KAT src/old/nthash_hpc.rs.orig:68-77.  Emits (start, hash64). */
size_t s2k_oracle_hpc_literal_u64(const uint8_t *s, size_t n, unsigned l, uint64_t bound,
                                  uint64_t *j, uint64_t *hash, size_t cap);

/* k-min-mers of one read: KminmersIterator, src/lib.rs:89-131,179-270; record layout
 * src/kminmer.rs:128-135.  offset of item i is i. Arrays may be NULL. Returns the count. */
size_t s2k_oracle_kminmers(const uint8_t *s, size_t n, unsigned l, unsigned k, double density, int mode,
                           uint64_t *hash, uint64_t *start, uint64_t *end, uint8_t *rev, size_t cap);

/* Literal rolling form of src/lib.rs:231-266 over given minimizer hashes (to prove the closed
 * form used by s2k_oracle_kminmers). */
size_t s2k_oracle_kminmer_hashes_rolling(const uint32_t *mh, size_t m, unsigned k, uint64_t *hash, uint8_t *rev);

/* Batch: n_reads reads stored back to back in `bases`, read r = [off[r], off[r+1]).
 * km_off (n_reads+1) receives the exclusive prefix of per-read k-min-mer counts.
 * Output arrays may be NULL (count only). `threads` <= 1 runs inline. Returns total count. */
uint64_t s2k_oracle_batch(const uint8_t *bases, const uint64_t *off, uint64_t n_reads,
                          unsigned l, unsigned k, double density, int mode, int threads,
                          uint64_t *km_off, uint64_t *hash, uint32_t *start, uint32_t *end, uint8_t *rev,
                          uint64_t cap);

/* Batch minimizer triples, same conventions (mn_off = per-read prefix of minimizer counts). */
uint64_t s2k_oracle_batch_minimizers(const uint8_t *bases, const uint64_t *off, uint64_t n_reads,
                                     unsigned l, double density, int mode,
                                     uint64_t *mn_off, uint32_t *j, uint32_t *jend, uint32_t *hash, uint64_t cap);

/* Deterministic synthetic reads (splitmix64 keyed by (seed, 32-base block index)); identical to
 * the device generator in the product library so inputs can be created in HBM. */
void s2k_oracle_synth_bases(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *out);

/* Whole-run checksums of n_reads synthetic reads of read_len bases (read r = synth stream [r*read_len, (r+1)*read_len)),
 * generated on the fly: out = { n_minimizers, n_kminmers, XOR hash, SUM start, SUM end, #rev, FOLD hash, FOLD start,
 * FOLD end, FOLD rev, FOLD km_off, FOLD per-read count }, FOLD x = SUM_g x[g] (2 g + 1) mod 2^64 over the global item
 * index g (order-sensitive), FOLD km_off = SUM_r km_off[r] (2 r + 1) over r = 0 .. n_reads. */
void s2k_oracle_synth_checksums(uint64_t seed, uint64_t n_reads, uint64_t read_len, unsigned l, unsigned k,
                                double density, int mode, int threads, uint64_t out[12]);

/* Same for ragged reads: read r = synth stream [off[r], off[r+1]). */
void s2k_oracle_synth_checksums_off(uint64_t seed, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                                    double density, int mode, int threads, uint64_t out[12]);

/* HiFi-like synthetic reads (BASELINE configs[3]): homopolymer runs drawn from splitmix64 keyed by
 * (seed, read index) -- geometric run lengths of mean 2, ~0.1 % of the runs stretched to 20..2999 bases -- lengths
 * ~N(15 000, 2 000) clipped to [2 000, 30 000]; identical to the device generator s2k_synth_hifi_device. */
uint64_t s2k_oracle_hifi_len(uint64_t seed, uint64_t r);
void s2k_oracle_hifi_lens(uint64_t seed, uint64_t r0, uint64_t n, uint64_t *out);
void s2k_oracle_hifi_read(uint64_t seed, uint64_t r, uint64_t n, uint8_t *out);
/* Whole-run checksums (as s2k_oracle_synth_checksums_off) of the reads r = 0 .. n_reads-1, read r of off[r+1] - off[r] bases. */
void s2k_oracle_hifi_checksums(uint64_t seed, const uint64_t *off, uint64_t n_reads, unsigned l, unsigned k,
                               double density, int mode, int threads, uint64_t out[12]);

#ifdef __cplusplus
}
#endif
#endif
