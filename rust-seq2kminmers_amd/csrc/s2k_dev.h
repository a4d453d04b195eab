// s2k_dev.h -- shared device/host definitions for the gfx950 kernels.
// Semantics follow the reference's scalar path; citations are to /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s2k {

// low 32 bits of the ntHash1 seeds (`as H`, H = u32): src/nthash_hpc.rs:30-49, src/lib.rs:31
constexpr uint32_t SEED_A = 0x95c60474u, SEED_C = 0x62a02b4cu, SEED_G = 0x82572324u, SEED_T = 0x4be24456u;

// scalar-path tables: N -> 0, every other byte -> 1 (src/nthash_hpc.rs:31,36,42,47)
__host__ __device__ inline uint32_t seed_h_scalar(uint32_t c) {
    return c == 'A' ? SEED_A : c == 'C' ? SEED_C : c == 'G' ? SEED_G : c == 'T' ? SEED_T : c == 'N' ? 0u : 1u;
}
__host__ __device__ inline uint32_t seed_rc_scalar(uint32_t c) {
    return c == 'A' ? SEED_T : c == 'C' ? SEED_G : c == 'G' ? SEED_C : c == 'T' ? SEED_A : c == 'N' ? 0u : 1u;
}
// AVX-512-path mapping: low nibble 1->A 3->C 7->G 4->T, everything else -> 0
// (src/nthash_avx512_32.rs:178-193 pshufb table, :225-262 permutexvar seeds)
__host__ __device__ inline uint32_t seed_h_simd(uint32_t c) {
    uint32_t n = c & 15u;
    return n == 1 ? SEED_A : n == 3 ? SEED_C : n == 7 ? SEED_G : n == 4 ? SEED_T : 0u;
}
__host__ __device__ inline uint32_t seed_rc_simd(uint32_t c) {
    uint32_t n = c & 15u;
    return n == 1 ? SEED_T : n == 3 ? SEED_G : n == 7 ? SEED_C : n == 4 ? SEED_A : 0u;
}

#if defined(__HIP_DEVICE_COMPILE__)
// a ^ b ^ c in ONE instruction: gfx950's three-input bit operation (truth table 0x96).  The compiler keeps two v_xor_b32 (each
// cheaper than a VOP3 instruction; the pair is not: 4.1 vs 3.1 issue cycles at three waves per SIMD, tools/experiments/valu_rate.hip)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#else
__host__ __device__ inline uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; } // (the host pass only parses the kernels)
#endif

__host__ __device__ inline uint32_t rotl32(uint32_t x, uint32_t r) { r &= 31u; return (x << r) | (x >> ((32u - r) & 31u)); }
__host__ __device__ inline uint32_t rotr32(uint32_t x, uint32_t r) { r &= 31u; return (x >> r) | (x << ((32u - r) & 31u)); }
__host__ __device__ inline uint64_t rotl64(uint64_t x, uint32_t r) { r &= 63u; return (x << r) | (x >> ((64u - r) & 63u)); }

// MixHash for u32: xorshift 13/7/17 on the zero-extended value -- src/lib.rs:157-169
__host__ __device__ inline uint64_t mix32(uint32_t h) {
    uint64_t x = h;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
}

// What a HashMode means, resolved on the host (s2k_api.hip:resolve_sem).
struct Sem {
    uint32_t l;
    uint32_t k;
    uint32_t bound_le;   // keep iff hash <= bound_le (strict '<' modes use bound-1) ...
    uint32_t enabled;    // ... and only if enabled (strict '<' with bound 0 keeps nothing)
    uint32_t hpc;        // minimizers in homopolymer-compressed space
    uint32_t simd_seeds; // low-nibble seed mapping (Simd/HpcSimd)
    uint32_t keep_last;  // Hpc scalar drops the last HPC l-mer (src/nthash_hpc.rs:265-267); others keep theirs
    uint32_t end_kind;   // 0: j+l-1 (lib.rs:202,226)  1: st[p+l]-1 (nthash_hpc.rs:281)  2: st[p+l-1] (nthash_hpc_simd.rs:64)
    uint32_t tail_quirk; // drop the final 16-block when #l-mers % 16 == 0 (src/nthash_avx512_32.rs:134-138)
    uint32_t dbg_skip;   // timing ablations only (env S2K_DEBUG_SKIP; results are wrong when set): 1 hash loop, 2 dense phase, 4 hpc compaction
    uint32_t pad_;
    // HpcSimd on the tiled kernel: its tail rule needs the number of runs of the whole read.  tile_heads (default): every tile
    // publishes, right after its compaction, how many run heads of the read that continues past its end it holds; a tile in which
    // such a read ends looks back (HW_* below).  read_runs (the fall-back, launch_read_run_counts): the table of a pre-pass.
    const uint32_t *read_runs;
    uint32_t *tile_heads;
};

struct Counts { // mirrored by s2k_counts (include/s2k.h)
    uint64_t n_reads, n_bases, n_minimizers, n_kminmers, xor_hash;
    uint32_t hash_bound, path;
    // internal
    uint64_t pool_needed;
    uint32_t pool_overflow, bad_input, km_overflow, mn_overflow; // bad_input: BAD_* bits set by validate_read_off_kernel
    uint32_t need_legacy, need_runs; // need_runs: HpcSimd: a look-back for the run heads of earlier tiles gave up (bounded polls): the host re-runs the call with the
                                     // runs of every read counted first.  need_legacy: descriptor path: a tile it cannot handle was met (> 30 read starts, a span >= 2^18): the host re-runs the call through the legacy path
    uint64_t dbg_cycles[64][16]; // S2K_DEBUG_SKIP & 8: shader-clock cycles per phase, summed over waves
#ifdef S2K_DEBUG_KNOBS
    uint64_t dbg_wave[4096][2];  // S2K_DEBUG_SKIP & 32: per wave {finish time (100 MHz) , XCC_ID << 32 | HW_ID}
#endif
};

// word a tile publishes in Sem::tile_heads: run heads of the tile that belong to the read continuing past its end (bits 0-15; the
// whole tile when HW_PASS: no read starts in it, the read it lies in began before it)
constexpr uint32_t HW_VALID = 1u << 31, HW_PASS = 1u << 30, HW_COUNT = 0xFFFFu;

// what validate_read_off_kernel found wrong with a caller's device-resident read table (s2k_extract_device trusts nothing)
enum : uint32_t { BAD_FIRST = 1u, BAD_ORDER = 2u, BAD_END = 4u, BAD_LONG = 8u };

constexpr int XOR_SHARDS = 4096;

// tile geometry of the tiled minimizer kernel (s2k_tile.hip)
constexpr int TILE_T = 144;              // bases per lane
constexpr int TILE_BASES = 64 * TILE_T;  // 9216 bases per wave-tile
// cursors of the tiled kernel's dynamic tile deal (one per lane of a wave that looks for an open one); they follow the
// overflow-pool cursor in one zeroed array of 64-bit words: word 0 = pool cursor, cursor g = the low half of word 16 + 16 g (32-bit, 128 B apart)
constexpr int TILE_CURSORS = 64;
constexpr int CURSOR_WORDS = 16 + 16 * TILE_CURSORS;

// Orders LDS traffic between the lanes of ONE wave (no workgroup barrier: waves of a block run independent
// tiles with different trip counts).  A wave's DS operations execute in program order, so waiting for the
// wave's own outstanding LDS operations (lgkmcnt) and stopping the compiler from moving memory accesses
// across this point is enough.  Deliberately NOT a fence: a wavefront-scope fence also drains vmcnt, i.e.
// it would stall on every global store / atomic / prefetch load in flight at each of the ~10 syncs per tile.
__device__ inline void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

struct Records { // SoA pool of minimizer records written by the minimizer kernels
    uint32_t *j, *jend, *hash, *rid;
    uint64_t capacity;  // total entries
    // tiled kernel: tile t owns the fixed slab [t*slab_cap, (t+1)*slab_cap); a tile with more hits than
    // that (low-complexity sequence) takes space from the shared overflow region [ovf_base, capacity)
    // with one atomic.  A single shared cursor for every tile serialises the whole kernel (~88 atomics
    // per microsecond on one address = 12 ms for 1.1 M tiles).
    uint64_t slab_cap, ovf_base;
    // ... ONE cursor for the whole call: the chunks of the two-stream pipeline each have their own tile cursors, but the overflow
    // region is shared -- and the k-min-mer kernel of a chunk reads its records while the minimizer kernel of the next writes
    unsigned long long *ovf_cursor;
};

// ---- descriptor path (the default for k <= 32): tile-relative records + one descriptor word per tile -----------------------------
// The tiled kernel leaves, per minimizer, 8 bytes -- its 32-bit hash and {offset of the l-mer's first base inside the tile : 14,
// span to its last base : 18} -- and, per tile, one word that says what the tile does to the pair (G, p):
//   G = k-min-mers that end before the tile (and, beside it, minimizers before the tile),
//   p = min(k-1, minimizers the read that continues into the tile has so far),
// namely   k-min-mers ending in the tile = C + (dep ? max(0, m_f - (k-1) + p) : 0),   p after the tile = pass ? min(k-1, p + m_f) : q.
// These functions compose associatively (agg_then), so one small scan over the tile words (s2k_desc.hip) gives every tile its
// (G, p); the k-min-mer kernel then needs no per-read table at all: a window that ends at the i-th minimizer of a tile goes to
// G + (windows ending earlier in the tile), read positions come from the tile's own list of read starts.  (Round 2 kept 16-byte
// records with the read index, three scans over per-read / per-tile counters and three dependent global round trips per tile
// in the k-min-mer kernel; that path is still here for k > 32 and for tiles with more than 30 read starts.)
struct AggF {
    uint64_t m_f, C, N;
    uint32_t q;
    bool dep, pass;
};
// agg word: m_f bits [0,14) = minimizers of the tile's first read segment, C [14,28) = k-min-mers ending in the tile that do not
// depend on p, N [28,42) = minimizers of the tile, q [42,48), dep bit 48, pass bit 49
__host__ __device__ inline unsigned long long agg_pack(uint32_t m_f, uint32_t C, uint32_t N, uint32_t q, bool dep, bool pass) {
    return (unsigned long long)m_f | ((unsigned long long)C << 14) | ((unsigned long long)N << 28) | ((unsigned long long)q << 42) |
           ((unsigned long long)(dep ? 1 : 0) << 48) | ((unsigned long long)(pass ? 1 : 0) << 49);
}
__host__ __device__ inline AggF agg_unpack(unsigned long long w) {
    AggF a;
    a.m_f = w & 0x3FFFu;
    a.C = (w >> 14) & 0x3FFFu;
    a.N = (w >> 28) & 0x3FFFu;
    a.q = (uint32_t)(w >> 42) & 63u;
    a.dep = ((w >> 48) & 1u) != 0;
    a.pass = ((w >> 49) & 1u) != 0;
    return a;
}
__host__ __device__ inline AggF agg_identity() { return AggF{0, 0, 0, 0, true, true}; }
__host__ __device__ inline uint64_t agg_windows(const AggF &a, uint32_t p, uint32_t K1) {
    uint64_t v = a.C;
    if (a.dep && a.m_f + p > K1) v += a.m_f + p - K1;
    return v;
}
__host__ __device__ inline uint32_t agg_p(const AggF &a, uint32_t p, uint32_t K1) {
    if (!a.pass) return a.q;
    const uint64_t v = a.m_f + p;
    return v > K1 ? K1 : (uint32_t)v;
}
// the run A followed by the run B
__host__ __device__ inline AggF agg_then(const AggF &A, const AggF &B, uint32_t K1) {
    AggF R;
    R.N = A.N + B.N;
    if (A.pass) { // A is one stretch of a read that began earlier and goes on: no k-min-mer count of its own yet (C == 0)
        R.dep = true;
        R.pass = B.pass;
        R.m_f = B.dep ? A.m_f + B.m_f : A.m_f; // max(0, a - K1 + p) + max(0, b - K1 + min(K1, p + a)) == max(0, a + b - K1 + p)
        R.C = B.C;
        R.q = B.q;
    } else { // after A, p is the constant A.q
        R.dep = A.dep;
        R.pass = false;
        R.m_f = A.m_f;
        R.C = A.C + agg_windows(B, A.q, K1);
        R.q = agg_p(B, A.q, K1);
    }
    return R;
}
constexpr int META_SEGS = 32; // read segments of a tile whose start is kept (NBL of s2k_tile_impl.h); tiles with more take the legacy path
struct TileMeta {            // written by the tiled kernel, read by the k-min-mer kernel
    unsigned long long rec_base; // the tile's first record in the pool (its slab, or a piece of the overflow region)
    unsigned long long rs0;      // stream position where read r0 (the one that holds the tile's first base) starts
    uint32_t r0;                 // index of that read
    uint16_t nb, nrd;            // read starts strictly inside the tile; reads that start in (t0, end of the tile]
    uint16_t segstart[META_SEGS]; // minimizers of the tile before read segment s (s = 0 .. nb)
    uint16_t rs16[META_SEGS];     // read_off[r0 + s] - t0 for s = 1 .. nb
};
struct TileState { // written by the scan
    unsigned long long g;   // p << 48 | k-min-mers that end before the tile
    unsigned long long gmn; // minimizers before the tile
};
constexpr uint32_t REC_SPAN_MAX = (1u << 18) - 1u; // a span that does not fit sends the call to the legacy path
struct Desc { // arguments of the descriptor path (device pointers)
    unsigned long long *agg; // n_tiles
    TileMeta *meta;          // n_tiles
    TileState *state;        // n_tiles + 1 (the last entry holds the totals)
    uint32_t k;
    uint32_t spec_n; // records of a tile the k-min-mer kernel fetches with its first round trip (three per lane): the context's guess from its last call (0: none)
    unsigned long long km_capacity, mn_capacity;
    unsigned long long *o_km_off, *o_hash;
    uint32_t *o_start, *o_end;
    uint8_t *o_rev;
    unsigned long long *o_mn_off; // with mn_capacity != 0
    uint32_t *o_mn_j, *o_mn_jend, *o_mn_hash;
    unsigned long long *xor_shards;
};

#define S2K_HIP_CHECK(expr)                                                      \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) return _e;                                         \
    } while (0)

// ---- host launchers implemented in the kernel translation units --------------------------------
hipError_t launch_scan_u32(const uint32_t *in, uint64_t n, uint64_t *out /*n+1*/, uint64_t *block_tmp,
                           uint32_t sub_k /*0: identity, else max(0,x-sub_k+1)*/, hipStream_t st);
size_t scan_tmp_bytes(uint64_t n);

// encode_rle (src/hpc.rs:7-25) collapses a repeated character only if it is one of "ACTGactgNn" (src/hpc.rs:14); hpc and
// encode_rle_simd collapse any repeated byte.  `rle` selects the former in the standalone HPC kernels.
__host__ __device__ inline bool is_rle_char(uint32_t c) {
    const uint32_t d = (c & 0xDFu) - 'A'; // fold case; A C G N T are letters 0, 2, 6, 13, 19: one bit set each in a 20-bit mask
    return d < 20u && ((0x82045u >> d) & 1u) != 0u;
}
__host__ __device__ inline bool run_head(uint32_t cur, uint32_t prev, bool rle) { return cur != prev || (rle && !is_rle_char(cur)); }

// runs[r] = number of homopolymer runs of read r (equal adjacent bytes collapse; a read start always begins a run).
// blk_cnt: n_bases/256+1 u32, blk_off: n_bases/256+2 u64 (prefix of neq-counts per 256-byte block), scan_tmp:
// scan_tmp_bytes(n_bases/256+1).  read_c0 (optional, n_reads u64): that prefix evaluated at the start of every read.
hipError_t launch_read_run_counts(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases,
                                  uint32_t *blk_cnt, uint64_t *blk_off, uint64_t *scan_tmp, uint32_t *runs, uint64_t *read_c0,
                                  hipStream_t st, bool rle = false);
// Standalone homopolymer compression, segment-parallel (s2k_hpc_seg.hip): the compressed bytes and read-relative run
// starts of the whole batch, given hpc_off (= prefix of runs[]), blk_off and read_c0 from launch_read_run_counts.
hipError_t launch_hpc_segments(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases,
                               const uint64_t *hpc_off, const uint64_t *blk_off, const uint64_t *read_c0, uint32_t *seg_index, uint8_t *o_hpc,
                               uint32_t *o_pos, uint64_t capacity, hipStream_t st, bool rle = false);
// The same outputs (hpc_off of every read included) in ONE pass over the bases: the segments' first output slots by a decoupled look-back over the
// blocks' run-head counts (s2k_hpc_seg.hip).  ws: hpc_single_pass_words(n_bases) 32-bit words, 8-byte aligned; *fail_word (inside ws) is non-zero
// afterwards when a look-back gave up (bounded polls): then the outputs are incomplete and the two-pass path above must be run.
size_t hpc_single_pass_words(uint64_t n_bases);
hipError_t launch_hpc_single_pass(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, uint32_t *ws, uint64_t *o_hpc_off,
                                  uint8_t *o_hpc, uint32_t *o_pos, uint64_t capacity, uint32_t **fail_word, hipStream_t st, bool rle = false);
size_t hpc_segment_index_words(uint64_t n_bases); // uint32 words of seg_index (8-byte aligned workspace: per segment its first output slot, the read that holds its first byte and where that read starts)

// read_off[0] == 0, non-decreasing, read_off[n_reads] == n_bases, no read longer than 2^32 - 2: anything else sets BAD_*
// bits in *bad (a device word); the kernels that follow in the stream look at it and do nothing when it is set
hipError_t launch_validate_read_off(const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, uint32_t *bad, hipStream_t st);
// the same checks + tile_read0[0 .. n_tiles] (the read that holds the first base of every tile; entry n_tiles: n_reads - 1) + (tile_words != nullptr)
// tile_words[0 .. n_tiles) = word0, in one pass over the read table (s2k_util.hip: read_table_kernel)
hipError_t launch_read_table(const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, uint64_t n_tiles, uint32_t *bad, uint32_t *tile_read0,
                             unsigned long long *tile_words, unsigned long long word0, hipStream_t st);

hipError_t launch_synth(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *d, hipStream_t st);
hipError_t launch_fill_u64(unsigned long long *d, uint64_t n, unsigned long long v, hipStream_t st);
// HiFi-like reads r0 .. r0+n_reads-1 at d + read_off[i] (read_off = prefix of synth_hifi_len), see s2k_util.hip
uint64_t synth_hifi_len(uint64_t seed, uint64_t r);
hipError_t launch_synth_hifi(uint64_t seed, uint64_t r0, uint64_t n_reads, const uint64_t *read_off, uint8_t *d, hipStream_t st);

hipError_t launch_serial_count(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, Sem sem,
                               uint32_t *mn_cnt, hipStream_t st);
hipError_t launch_serial_write(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, Sem sem,
                               const uint64_t *mn_off, Records rec, Counts *counts, hipStream_t st);

hipError_t launch_kminmers(uint64_t n_tiles, const uint64_t *tile_rec_off, const uint32_t *tile_cnt,
                           const uint64_t *tile_goff, Records rec, const uint64_t *mn_off, const uint64_t *km_off,
                           uint32_t k, uint64_t km_capacity, uint64_t *o_hash, uint32_t *o_start, uint32_t *o_end,
                           uint8_t *o_rev, uint64_t mn_capacity, uint32_t *o_mn_j, uint32_t *o_mn_jend,
                           uint32_t *o_mn_hash, uint64_t *xor_shards, const Counts *counts, hipStream_t st);
hipError_t launch_finalize(Counts *counts, const uint64_t *xor_shards, const uint64_t *mn_total, const uint64_t *km_total,
                           uint64_t km_capacity, uint64_t mn_capacity, hipStream_t st);

// desc != nullptr: descriptor path (8-byte records in rec.hash / rec.j, agg word + meta per tile; tile_rec_off / tile_cnt / mn_cnt unused).
// The launch works on the tiles [tile_begin, n_tiles); pool_cursor must be zeroed (CURSOR_WORDS words) for every launch.
hipError_t launch_tile_minimizers(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases,
                                  uint64_t n_tiles, const uint32_t *tile_read0, Sem sem, Records rec,
                                  uint64_t *pool_cursor, uint64_t *tile_rec_off, uint32_t *tile_cnt, uint32_t *mn_cnt,
                                  Counts *counts, const Desc *desc, uint64_t tile_begin, hipStream_t st);
// descriptor path, for the tiles [tile_begin, tile_end): (G, p) of every tile from the tile words -- starting from state[tile_begin]
// (the totals a previous call for the tiles before left there; zeros for tile_begin == 0) and leaving the totals in state[tile_end] --
// then the k-min-mers of those tiles
hipError_t launch_desc_scan(uint64_t tile_begin, uint64_t tile_end, Desc dz, unsigned long long *scan_tmp /* desc_scan_tmp_words(n_tiles) */,
                            const Counts *counts, hipStream_t st);
size_t desc_scan_tmp_words(uint64_t n_tiles);
// alone: no other kernel of the call runs beside this launch (it may use the CU's LDS for staged, line-filling stores)
hipError_t launch_desc_kminmers(uint64_t tile_begin, uint64_t tile_end, uint64_t n_tiles, uint64_t n_reads, Desc dz, Records rec, Counts *counts,
                                hipStream_t st, bool alone);

hipError_t launch_hpc_count(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint32_t *run_cnt, hipStream_t st,
                            bool rle = false);
hipError_t launch_hpc_write(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, const uint64_t *hpc_off,
                            uint8_t *o_hpc, uint32_t *o_pos, uint64_t capacity, hipStream_t st, bool rle = false);

// FASTA/FASTQ record splitting in HBM (s2k_fastx_dev.hip).  count: totals = {records, sequence bytes, syntax errors};
// write: bases + read_off (n_records + 1 entries).  `ws` holds fx_ws_bytes(n) bytes and must survive both phases.
size_t fx_ws_bytes(uint64_t n);
hipError_t fx_parse_count(const uint8_t *d_raw, uint64_t n, int fastq, void *ws, uint64_t *d_totals, hipStream_t st);
hipError_t fx_parse_write(const uint8_t *d_raw, uint64_t n, int fastq, void *ws, uint64_t *d_totals, uint8_t *d_bases,
                          uint64_t bases_cap, uint64_t *d_read_off, uint64_t off_cap, hipStream_t st);

} // namespace s2k
