/*
 * s2k.h -- C ABI of the MI355X-native k-min-mer extraction engine.
 *
 * This is the drop-in boundary for ONE path of rchikhi/rust-seq2kminmers: "reads in -> ordered
 * stream of k-min-mers out".  In the reference that path is
 *     KminmersIterator::new(seq, l, k, density, mode)      src/lib.rs:89-131
 *     impl Iterator for KminmersIterator { Item = KminmerHash }   src/lib.rs:179-270
 *     struct KminmerHash { hash, start, end, offset, rev }  src/kminmer.rs:128-135
 * called once per read from a thread pool (src/main.rs:65-79).  A GPU needs >= 10^4 reads per
 * launch, so the ABI is batch-oriented: n_reads reads stored back to back, one call, SoA results;
 * the per-read Iterator shape is rebuilt on top of it by include/s2k.hpp (C++) and by the Rust shim
 * shown in INTEGRATION.md.  The crate's own precedent for an opaque-handle C ABI is
 * src/nthash_c.rs:14-29 (create / roll / get / destroy).
 *
 * Conventions: plain C types only; the caller owns inputs; the library owns s2k_result buffers until
 * s2k_result_free(); nothing panics or throws across the ABI -- parameter violations that make the
 * reference panic (src/lib.rs:99 unwrap of KSizeTooBig, src/nthash_hpc.rs:133 assert!(k<256),
 * src/lib.rs:246 underflow at k==0) come back as status codes.  One s2k_ctx per (thread, device);
 * contexts are independent.  All results are bit-exact to the reference's scalar path
 * (HashMode::Regular / HashMode::Hpc); the Simd / HpcSimd *result semantics* (strict '<', f32 bound,
 * start-of-run end, kept last l-mer, low-nibble seed map -- SURVEY.md 8a traps i-vi) are selectable.
 */
#ifndef S2K_H
#define S2K_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: round-1 surface.  2: + s2k_set_host_batch, s2k_hpc_device_ex, s2k_count_device, s2k_partition_device,
 * S2K_FLAG_NO_PACK2; s2k_extract sends 2-bit packed bases and pipelines sub-batches by default; the arrays of an s2k_result
 * are NULL when their count is 0; device-resident read tables are validated on the device (S2K_ERR_INVALID_ARG /
 * S2K_ERR_READ_TOO_LONG from s2k_extract_device, s2k_sync and s2k_hpc_device*); S2K_ERR_NON_ASCII is no longer returned;
 * + s2k_trim, s2k_density_for_bound, S2K_FLAG_LEGACY_PATH; s2k_counts.path tells the descriptor path (0) from the legacy one (2).
 * (Round 4 only ADDED entry points -- the benchmark helpers s2k_synth_hifi_lengths / s2k_synth_hifi_device, and s2k_chain_after: no existing entry point or
 * struct changed, the version stays 2.)
 * A binding must refuse a library whose s2k_abi_version() differs from the header it was built against. */
#define S2K_ABI_VERSION 2

typedef struct s2k_ctx s2k_ctx; /* opaque; cf. nthashc_create/destroy, src/nthash_c.rs:14-29 */

typedef enum s2k_status {
    S2K_OK = 0,
    S2K_ERR_INVALID_ARG = 1,   /* NULL pointer, unknown mode, a read table that does not start at 0 / is not non-decreasing / does not end at n_bases */
    S2K_ERR_L_RANGE = 2,       /* l == 0 or l >= 256 (src/nthash_hpc.rs:123-125,133); Simd modes: l > 31 (src/nthash_avx512_32.rs:33) */
    S2K_ERR_K_RANGE = 3,       /* k == 0 (src/lib.rs:246 would underflow) or k > 4096 */
    S2K_ERR_READ_TOO_LONG = 4, /* a read longer than 2^32-2 bases (positions are u32, cf. src/nthash_hpc_simd.rs:26) */
    S2K_ERR_DEVICE = 5,        /* HIP runtime error; s2k_last_error() has the text */
    S2K_ERR_NOMEM = 6,
    S2K_ERR_CAPACITY = 7,      /* caller-provided device output too small; counts say what is needed */
    S2K_ERR_NO_DEVICE = 8,     /* no gfx950 device visible: there is NO CPU fallback */
    S2K_ERR_NON_ASCII = 9      /* reserved: returned by ABI 1 for bytes >= 0x80 in Hpc mode; every byte value is accepted now */
} s2k_status;

/* HashMode, src/lib.rs:21-27 */
typedef enum s2k_mode {
    S2K_MODE_REGULAR = 0, /* src/lib.rs:215-230 */
    S2K_MODE_HPC = 1,     /* NtHashHPCIterator, src/nthash_hpc.rs:115-283 */
    S2K_MODE_SIMD = 2,    /* result semantics of NtHashSIMDIterator, src/nthash_avx512_32.rs:32-164 */
    S2K_MODE_HPCSIMD = 3  /* result semantics of NtHashHPCSIMDIterator, src/nthash_hpc_simd.rs:35-68 */
} s2k_mode;

enum {
    S2K_FLAG_WANT_MINIMIZERS = 1u << 0, /* also return the (j, jend, hash32) triples (NtHashHPCIterator::Item, src/nthash_hpc.rs:193) */
    S2K_FLAG_FORCE_SERIAL = 1u << 1,    /* use the read-serial kernels (exact for every input; slow) instead of the tiled ones */
    S2K_FLAG_NO_PACK2 = 1u << 2,        /* s2k_extract, s2k_run_file: send the text as it is instead of 2-bit packed (+ exception list) over PCIe */
    S2K_FLAG_LEGACY_PATH = 1u << 3      /* tiled kernels with the round-2 records (16 B with the read index, per-read scans; counts.path 2) instead of the descriptor path */
};

typedef struct s2k_params {
    uint32_t l;       /* minimizer length            (src/lib.rs:89 `l`) */
    uint32_t k;       /* k-min-mer order             (src/lib.rs:89 `k`) */
    double density;   /* FH = f64                    (src/lib.rs:34,89) */
    int32_t mode;     /* s2k_mode */
    uint32_t flags;
} s2k_params;

typedef struct s2k_counts {
    uint64_t n_reads;
    uint64_t n_bases;
    uint64_t n_minimizers;
    uint64_t n_kminmers;
    uint64_t xor_hash;   /* XOR of all k-min-mer hashes (cheap whole-run checksum) */
    uint32_t hash_bound; /* the u32 bound of src/lib.rs:91 that was used */
    uint32_t path;       /* 0 = tiled kernels, descriptor path (8-byte tile-relative records); 1 = read-serial kernels;
                          * 2 = tiled kernels, legacy records (k > 32, a tile with more than 30 read starts, S2K_FLAG_LEGACY_PATH) */
} s2k_counts;

/* Host-side result, SoA.  Item i of read r (km_off[r] <= i < km_off[r+1]) is
 * KminmerHash{ hash[i], start[i], end[i], offset = i - km_off[r], rev[i] }  (src/kminmer.rs:128-135).
 * hash / start / end / rev are NULL when n_kminmers == 0, the mn_* arrays when n_minimizers == 0. */
typedef struct s2k_result {
    uint64_t n_reads;
    uint64_t n_kminmers;
    uint64_t *km_off; /* n_reads + 1 */
    uint64_t *hash;   /* KH = u64, src/lib.rs:37 */
    uint32_t *start;
    uint32_t *end;
    uint8_t *rev;
    /* only with S2K_FLAG_WANT_MINIMIZERS, else NULL / 0 */
    uint64_t n_minimizers;
    uint64_t *mn_off; /* n_reads + 1 */
    uint32_t *mn_j;
    uint32_t *mn_jend;
    uint32_t *mn_hash; /* H = u32, src/lib.rs:31 */
    s2k_counts counts;
    void *_owner;
} s2k_result;

/* Device-resident output buffers provided by the caller (device pointers, capacities in items). */
typedef struct s2k_device_out {
    uint64_t km_capacity;
    uint64_t *km_off; /* n_reads + 1, required */
    uint64_t *hash;   /* may be NULL to skip */
    uint32_t *start;
    uint32_t *end;
    uint8_t *rev;
    uint64_t mn_capacity; /* 0 => minimizer triples not wanted */
    uint64_t *mn_off;
    uint32_t *mn_j;
    uint32_t *mn_jend;
    uint32_t *mn_hash;
} s2k_device_out;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int s2k_abi_version(void);
int s2k_device_count(void);
/* Creates a context on HIP device `device`.  Fails with S2K_ERR_NO_DEVICE when no GPU is visible. */
s2k_ctx *s2k_create(int device, s2k_status *status);
void s2k_destroy(s2k_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own. */
s2k_status s2k_set_stream(s2k_ctx *ctx, void *hip_stream);
/* Double buffering with two contexts on one device (a loop over many device-resident batches): after s2k_chain_after(b, a) the minimizer
 * kernels of b's calls wait for the minimizer kernels of a's most recent call, nothing else does -- so the tail of a's call (its last
 * k-min-mer kernel, the totals, the host's look at the counts) runs beside the first chunk of b's, and two persistent kernels never compete
 * for the device.  Chain both ways (a after b, b after a) and alternate the calls from ONE host thread.  prev = NULL removes the link;
 * s2k_destroy(prev) removes every link to prev by itself (the survivor then runs unchained), from any thread: links are read and cut under one
 * lock.  A prev that is not a live context is refused (S2K_ERR_INVALID_ARG).  Results do not depend on it. */
s2k_status s2k_chain_after(s2k_ctx *ctx, s2k_ctx *prev);
/* s2k_extract cuts a call into sub-batches of whole reads of about `bases` bases each (default 2^29; 0 restores it) and
 * pipelines them: H2D of one, kernels of the previous, D2H of the one before run side by side.  Results do not depend
 * on it. */
s2k_status s2k_set_host_batch(s2k_ctx *ctx, uint64_t bases);
const char *s2k_strerror(s2k_status st);
const char *s2k_last_error(const s2k_ctx *ctx);

/* (density * u32::MAX as f64) as u32 -- src/lib.rs:91 */
uint32_t s2k_hash_bound(double density);
/* A density d with s2k_hash_bound(d) == bound, for callers that hold the u32 bound itself -- the crate's minimizer iterators
 * take it instead of a density: NtHashHPCIterator::new(seq, k, hash_bound) src/nthash_hpc.rs:115, NtHashSIMDIterator::new
 * src/nthash_avx512_32.rs:32, NtHashHPCSIMDIterator::new src/nthash_hpc_simd.rs:35. */
double s2k_density_for_bound(uint32_t bound);

/* ---- the hot path ---------------------------------------------------------------------------- */
/* Replaces: for each read r { KminmersIterator::new(&bases[read_off[r]..read_off[r+1]], l, k, density,
 * mode)?.collect() } -- src/lib.rs:89,179 driven by src/main.rs:65-79.  Host buffers in, host SoA out
 * (library-owned; release with s2k_result_free).  Reads with len <= l yield nothing (src/lib.rs:97).
 * s2k_result_free hands the arrays back to a pool kept by the context that made them (fresh gigabyte allocations cost
 * more in page faults than the copy): a long-lived context retains the arrays of up to two freed results (18 blocks, sized
 * by the largest recent call) of host memory until s2k_trim() or s2k_destroy(). */
s2k_status s2k_extract(s2k_ctx *ctx, const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads,
                       const s2k_params *params, s2k_result *out);
void s2k_result_free(s2k_result *res);

/* Releases what a context keeps between calls: idle host result blocks and the device workspace (both grow again on
 * demand).  Call it when a long-lived context has seen an unusually large batch. */
s2k_status s2k_trim(s2k_ctx *ctx);

/* Same computation with inputs and outputs resident in HBM: d_read_off[0] must be 0, the table non-decreasing and
 * d_read_off[n_reads] == n_bases (read r = d_bases[d_read_off[r] .. d_read_off[r+1])) -- checked on the device: a table
 * that breaks this yields S2K_ERR_INVALID_ARG (a read longer than 2^32-2 bases S2K_ERR_READ_TOO_LONG) from this call
 * or, with counts == NULL, from s2k_sync(), and nothing is computed.  A d_bases pointer that is not 16-byte aligned is
 * accepted: the stream is first copied to an aligned buffer (one device-to-device pass) and takes the same tiled
 * kernels.  Enqueued on the context's stream; if
 * `counts` is non-NULL the call waits for completion and fills it (and returns S2K_ERR_CAPACITY if
 * an output capacity was too small -- counts then hold the required sizes).  With counts == NULL the
 * call returns after enqueueing; s2k_sync() later waits and reports the same status/counts. */
s2k_status s2k_extract_device(s2k_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_read_off,
                              uint64_t n_reads, uint64_t n_bases, const s2k_params *params,
                              const s2k_device_out *out, s2k_counts *counts);
s2k_status s2k_sync(s2k_ctx *ctx, s2k_counts *counts);

/* ---- standalone homopolymer compression (src/hpc.rs:28-41 hpc, :44-147 encode_rle_simd) ---------- */
/* Per read: compressed string + run-start positions (read-relative); same layout rules as s2k_extract_device.
 * d_hpc_off: n_reads+1 prefix of
 * run counts; d_hpc / d_pos (either may be NULL) receive up to `capacity` entries. */
s2k_status s2k_hpc_device(s2k_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads,
                          uint64_t n_bases, uint64_t *d_hpc_off, uint8_t *d_hpc, uint32_t *d_pos,
                          uint64_t capacity, uint64_t *n_runs);
/* Same with flags.  S2K_HPC_RLE_ALPHABET selects the run rule of `encode_rle` (src/hpc.rs:7-25): a repeated character
 * collapses only if it is one of "ACTGactgNn" (src/hpc.rs:14); without it any repeated byte collapses (`hpc`
 * src/hpc.rs:28-41, `encode_rle_simd` src/hpc.rs:44-147).  Not reproduced on the device: the two scalar functions start
 * from prev_char = '#', so they drop '#' characters that are followed by another character, report a stale position for
 * the run after a '#', and return "#" for an empty string; here '#' is an ordinary byte and an empty read has no runs
 * (the host facades add the empty-string case and refuse input that contains '#'). */
#define S2K_HPC_RLE_ALPHABET 1u
s2k_status s2k_hpc_device_ex(s2k_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads,
                             uint64_t n_bases, uint32_t flags, uint64_t *d_hpc_off, uint8_t *d_hpc, uint32_t *d_pos,
                             uint64_t capacity, uint64_t *n_runs);

/* ---- downstream of the path: k-min-mer counting ------------------------------------ */
/* What the consumer of the iterator does in rust-mdbg (a concurrent map keyed by the k-min-mer hash; the reference only
 * hints at it, src/lib.rs:256-257; KminmerHash is Eq/Hash by `hash` alone, src/kminmer.rs:181-204): the distinct values
 * of d_hash[0..n) and how often each occurs.  d_keys / d_counts (either may be NULL) receive up to `capacity` pairs in
 * unspecified order; *n_distinct is always set (S2K_ERR_CAPACITY when it exceeds capacity). */
s2k_status s2k_count_device(s2k_ctx *ctx, const uint64_t *d_hash, uint64_t n, uint64_t *d_keys, uint32_t *d_counts,
                            uint64_t capacity, uint64_t *n_distinct);
/* Send buffers of the multi-GPU version: d_out = the keys grouped into n_parts ranges of the hash space (part p holds the
 * hashes h with floor(h * n_parts / 2^64) == p, i.e. split by hash prefix), d_part_off[n_parts + 1] = group starts.  After
 * an all-to-all of the groups, rank p owns every occurrence of its range and counts locally. */
#define S2K_MAX_PARTS 64
s2k_status s2k_partition_device(s2k_ctx *ctx, const uint64_t *d_hash, uint64_t n, uint32_t n_parts, uint64_t *d_out,
                                uint64_t *d_part_off);

/* ---- helpers for device-resident benchmarking ------------------------------------------------- */
/* Fills d_bases[0..n) with the deterministic synthetic ACGT stream (splitmix64 keyed by seed and
 * absolute base index; same function as oracle/s2k_oracle.c:s2k_oracle_synth_bases). */
s2k_status s2k_synth_bases_device(s2k_ctx *ctx, uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *d_bases);
/* HiFi-like synthetic reads (BASELINE configs[3]: lengths ~N(15 000, 2 000), homopolymer runs of geometric length with mean 2,
 * ~0.1 % of them stretched to 20..2999 bases; splitmix64 keyed by (seed, read index); same functions as the oracle's
 * s2k_oracle_hifi_len / s2k_oracle_hifi_read).  s2k_synth_hifi_lengths fills a HOST array with the lengths of reads
 * r0 .. r0+n_reads-1; s2k_synth_hifi_device writes read r0+i at d_bases + d_read_off[i] (d_read_off: device, n_reads+1 entries,
 * the prefix sums of those lengths). */
void s2k_synth_hifi_lengths(uint64_t seed, uint64_t r0, uint64_t n_reads, uint64_t *lengths);
s2k_status s2k_synth_hifi_device(s2k_ctx *ctx, uint64_t seed, uint64_t r0, uint64_t n_reads, const uint64_t *d_read_off, uint8_t *d_bases);
/* Duration in ms of the kernels of the last s2k_extract_device call, measured with HIP events on the
 * context's stream: which: 0 = whole pipeline (first kernel's start to the last one's end), 1 = the minimizer kernel (dominant; the
 * descriptor path launches it once per chunk of tiles: first start to last end), 2 = scan + k-min-mer kernels (first start to last
 * end; on the descriptor path they run on a second stream BESIDE the minimizer kernel of the next chunk, so 1 and 2 overlap and
 * do not add up to 0). */
s2k_status s2k_last_kernel_ms(s2k_ctx *ctx, int which, float *ms);
s2k_status s2k_enable_timing(s2k_ctx *ctx, int on);
/* Sum over every s2k_extract_device call since s2k_enable_timing(ctx, 1) of the same HIP-event
 * durations (`which` as above), and the number of calls: average kernel duration over a timed region. */
s2k_status s2k_timing_total(s2k_ctx *ctx, int which, double *ms_sum, uint32_t *n_calls);

/* ---- FASTA / FASTQ ingest and batching (next row of SURVEY.md 8f) --------------------------------------
 * Counterpart of the reference's file mode: rust_parallelfastx::parallel_fastx(file, threads, task) feeding one
 * KminmersIterator per record (src/main.rs:51-83).  Records are parsed on the host into batches
 * (bases back to back + read_off) held in pinned memory, so a batch is what s2k_extract consumes. */
typedef struct s2k_fastx s2k_fastx; /* opaque reader */
/* Opens a plain (uncompressed) FASTA (multi-line allowed) or FASTQ (4-line) file; the format is detected from the
 * first non-blank byte. */
s2k_fastx *s2k_fastx_open(const char *path, s2k_status *status);
/* Parses the next records until `max_bases` bases or `max_reads` reads are collected (at least one record).
 * *bases / *read_off point into reader-owned buffers, valid until the next call; *n_reads == 0 at end of file. */
s2k_status s2k_fastx_next(s2k_fastx *rd, uint64_t max_bases, uint64_t max_reads, const uint8_t **bases,
                          const uint64_t **read_off, uint64_t *n_reads);
void s2k_fastx_close(s2k_fastx *rd);
/* Record splitting on the GPU: FASTA/FASTQ text that is already in HBM (d_text, 16-byte aligned, < 4 GiB, beginning
 * at a record start and ending at a record end) becomes the (bases, read_off) pair s2k_extract_device consumes.
 * format: S2K_FASTA (multi-line allowed) or S2K_FASTQ (strict 4-line records).  *n_reads / *n_bases always receive
 * the sizes; S2K_ERR_CAPACITY when bases_capacity < *n_bases or off_capacity < *n_reads + 1 (call again),
 * S2K_ERR_INVALID_ARG for malformed text. */
enum { S2K_FASTA = 0, S2K_FASTQ = 1 };
s2k_status s2k_fastx_parse_device(s2k_ctx *ctx, const uint8_t *d_text, uint64_t n_bytes, int format, uint8_t *d_bases,
                                  uint64_t bases_capacity, uint64_t *d_read_off, uint64_t off_capacity,
                                  uint64_t *n_reads, uint64_t *n_bases);
/* Streams a whole file through the GPU in batches of ~batch_bases and returns the totals: the file's bytes are
 * staged to HBM by several threads, records are split on the GPU (as s2k_fastx_parse_device), and staging +
 * splitting of batch i+1 overlap the k-min-mer kernels of batch i; k-min-mers are produced in HBM and counted, like the reference's demo task
 * (src/main.rs:65-76).  `seconds` receives the wall time of the whole call (file read included). */
s2k_status s2k_run_file(s2k_ctx *ctx, const char *path, const s2k_params *params, uint64_t batch_bases,
                        s2k_counts *totals, double *seconds);

#ifdef __cplusplus
}
#endif
#endif /* S2K_H */
