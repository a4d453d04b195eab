// s2k_util.hip -- small supporting kernels: exclusive scans over per-read / per-tile counts,
// the synthetic base generator, and the final counts reduction.  None of these is hot: they touch
// O(n_reads + n_tiles) words while the minimizer kernel touches O(n_bases) bytes.
#include "s2k_dev.h"

namespace s2k {

namespace {
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

__device__ inline uint64_t scan_f(uint32_t x, uint32_t sub_k) {
    // sub_k != 0: number of k-min-mers of a read with x minimizers = max(0, x-k+1) (src/lib.rs:235-261)
    return sub_k ? (x >= sub_k ? (uint64_t)(x - sub_k + 1) : 0ull) : (uint64_t)x;
}

__device__ inline uint64_t block_reduce_sum(uint64_t v, uint64_t *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    uint64_t t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += sh[i];
    return t; // valid in thread 0
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums(const uint32_t *__restrict__ in, uint64_t n,
                                                                uint64_t *__restrict__ block_tmp, uint32_t sub_k) {
    __shared__ uint64_t sh[SCAN_THREADS / 64];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK;
    uint64_t s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t idx = base + (uint64_t)i * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += scan_f(in[idx], sub_k);
    }
    uint64_t t = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) block_tmp[blockIdx.x] = t;
}

// single block: in-place exclusive scan of block_tmp[0..nb), total -> block_tmp[nb]
__global__ __launch_bounds__(1024) void scan_block_offsets(uint64_t *block_tmp, uint64_t nb) {
    __shared__ uint64_t sh[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < nb; base += 1024) {
        uint64_t idx = base + threadIdx.x;
        uint64_t v = idx < nb ? block_tmp[idx] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint64_t a = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += a;
            __syncthreads();
        }
        uint64_t incl = sh[threadIdx.x];
        uint64_t c = carry;
        if (idx < nb) block_tmp[idx] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_tmp[nb] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_write(const uint32_t *__restrict__ in, uint64_t n,
                                                           uint64_t *__restrict__ out,
                                                           const uint64_t *__restrict__ block_tmp, uint64_t nb,
                                                           uint32_t sub_k) {
    __shared__ uint64_t sh[SCAN_THREADS];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t idx = base + i;
        v[i] = idx < n ? scan_f(in[idx], sub_k) : 0;
        s += v[i];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {
        uint64_t a = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += a;
        __syncthreads();
    }
    uint64_t run = block_tmp[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t idx = base + i;
        if (idx < n) out[idx] = run;
        run += v[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_tmp[nb];
}

__global__ void scan_empty(uint64_t *out) { out[0] = 0; }

// ---- synthetic bases -------------------------------------------------------------------------
__device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// One thread = one aligned 16-base group (16 B store).  Uniform random ACGT like benches/bench.rs:19-31,
// keyed by absolute base index so any sub-range reproduces (oracle: s2k_oracle_synth_bases).
__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *__restrict__ d) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // 16-base group relative to aligned start
    uint64_t q0 = (first_base & ~15ull) + g * 16;
    if (q0 >= first_base + n) return;
    uint64_t z = splitmix64(seed ^ ((q0 >> 5) * 0xD6E8FEB86659FD93ULL));
    uint32_t bits = (uint32_t)(z >> (2 * (q0 & 31)));
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t x = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t code = (bits >> (2 * (4 * i + b))) & 3u;
            uint32_t ch = code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T';
            x |= ch << (8 * b);
        }
        w[i] = x;
    }
    if (q0 >= first_base && q0 + 16 <= first_base + n && (((uintptr_t)(d + (q0 - first_base))) & 15) == 0) {
        *reinterpret_cast<uint4 *>(d + (q0 - first_base)) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (int b = 0; b < 16; b++) {
            uint64_t q = q0 + b;
            if (q >= first_base && q < first_base + n) d[q - first_base] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
}

__global__ __launch_bounds__(256) void finalize_kernel(Counts *counts, const uint64_t *__restrict__ xor_shards,
                                                       const uint64_t *mn_total, const uint64_t *km_total,
                                                       uint64_t km_capacity, uint64_t mn_capacity) {
    __shared__ uint64_t sh[4];
    uint64_t x = 0;
    for (int i = threadIdx.x; i < XOR_SHARDS; i += blockDim.x) x ^= xor_shards[i];
    for (int o = 32; o > 0; o >>= 1) x ^= __shfl_down(x, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        counts->xor_hash = sh[0] ^ sh[1] ^ sh[2] ^ sh[3];
        const uint64_t M48 = (1ull << 48) - 1ull; // (the descriptor path keeps p in the bits above: zero behind the last tile anyway)
        const uint64_t mn = *mn_total & M48, km = *km_total & M48;
        counts->n_minimizers = mn;
        counts->n_kminmers = km;
        counts->km_overflow = (km > km_capacity) ? 1u : 0u;
        counts->mn_overflow = (mn_capacity != 0 && mn > mn_capacity) ? 1u : 0u;
    }
}

} // namespace

size_t scan_tmp_bytes(uint64_t n) { return ((n + SCAN_BLOCK - 1) / SCAN_BLOCK + 2) * sizeof(uint64_t); }

hipError_t launch_scan_u32(const uint32_t *in, uint64_t n, uint64_t *out, uint64_t *block_tmp, uint32_t sub_k,
                           hipStream_t st) {
    if (n == 0) {
        hipLaunchKernelGGL(scan_empty, dim3(1), dim3(1), 0, st, out);
        return hipGetLastError();
    }
    uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, block_tmp, sub_k);
    hipLaunchKernelGGL(scan_block_offsets, dim3(1), dim3(1024), 0, st, block_tmp, nb);
    hipLaunchKernelGGL(scan_write, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, out, block_tmp, nb, sub_k);
    return hipGetLastError();
}

// ---- HiFi-like synthetic reads (BASELINE configs[3]) ---------------------------------------------------------
// Read r is a sequence of homopolymer runs drawn from splitmix64 keyed by (seed, r): run lengths geometric with mean 2, ~0.1 % of
// the runs stretched to 20..2999 bases, consecutive runs of different letters; its length is ~N(15 000, 2 000) from eight 16-bit
// uniforms in integer arithmetic, clipped to [2 000, 30 000].  Same functions as oracle/s2k_oracle.c (s2k_oracle_hifi_len / _read).
__host__ __device__ inline uint64_t splitmix64_hd(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t hifi_key(uint64_t seed, uint64_t r) { return seed ^ (r * 0xA24BAED4963EE407ULL) ^ 0x9FB21C651E98DF25ULL; }
uint64_t synth_hifi_len(uint64_t seed, uint64_t r) {
    const uint64_t z0 = splitmix64_hd(hifi_key(seed, r) ^ 0x5851F42D4C957F2DULL), z1 = splitmix64_hd(z0);
    int64_t s = 0;
    for (int i = 0; i < 4; i++) s += (int64_t)((z0 >> (16 * i)) & 0xFFFF) + (int64_t)((z1 >> (16 * i)) & 0xFFFF);
    const int64_t len = 15000 + ((s - 262140) * 2000) / 53510;
    return (uint64_t)(len < 2000 ? 2000 : len > 30000 ? 30000 : len);
}
// One thread per read, runs written 16 bytes at a time where the destination is aligned (a read is a sequential process; a
// million of them run side by side).  Test / bench scaffolding, not a hot kernel.
__global__ __launch_bounds__(256) void synth_hifi_kernel(uint64_t seed, uint64_t r0, uint64_t n_reads, const uint64_t *__restrict__ read_off,
                                                         uint8_t *__restrict__ d) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    const uint64_t a = read_off[i], n = read_off[i + 1] - a;
    uint8_t *out = d + a;
    uint64_t x = splitmix64_hd(hifi_key(seed, r0 + i));
    uint32_t letter = 0;
    for (uint64_t pos = 0, run = 0; pos < n; run++) {
        x = splitmix64_hd(x);
        uint64_t len = 1 + (uint64_t)__builtin_ctz((uint32_t)x | 0x80000000u);
        if (((x >> 32) & 1023) == 0) len = 20 + ((x >> 42) & 0x3FFF) % 2980;
        letter = run == 0 ? (uint32_t)(x >> 60) & 3u : (letter + 1u + (uint32_t)((x >> 56) & 15u) % 3u) & 3u;
        const uint32_t ch = letter == 0 ? 'A' : letter == 1 ? 'C' : letter == 2 ? 'G' : 'T';
        if (len > n - pos) len = n - pos;
        if (len >= 48) { // long run: bytes up to the next 16-byte boundary, whole 16-byte pieces, the rest
            const uint32_t w = ch * 0x01010101u;
            while ((((uintptr_t)(out + pos)) & 15) && len) { out[pos++] = (uint8_t)ch; len--; }
            for (; len >= 16; len -= 16, pos += 16) *reinterpret_cast<uint4 *>(out + pos) = make_uint4(w, w, w, w);
        }
        for (; len; len--) out[pos++] = (uint8_t)ch;
    }
}
hipError_t launch_synth_hifi(uint64_t seed, uint64_t r0, uint64_t n_reads, const uint64_t *read_off, uint8_t *d, hipStream_t st) {
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(synth_hifi_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, seed, r0, n_reads, read_off, d);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long *__restrict__ d, uint64_t n, unsigned long long v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = v;
}
hipError_t launch_fill_u64(unsigned long long *d, uint64_t n, unsigned long long v, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, n, v);
    return hipGetLastError();
}

hipError_t launch_synth(uint64_t seed, uint64_t first_base, uint64_t n, uint8_t *d, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint64_t groups = ((first_base + n + 15) >> 4) - (first_base >> 4);
    uint64_t blocks = (groups + 255) / 256;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, first_base, n, d);
    return hipGetLastError();
}

hipError_t launch_finalize(Counts *counts, const uint64_t *xor_shards, const uint64_t *mn_total, const uint64_t *km_total,
                           uint64_t km_capacity, uint64_t mn_capacity, hipStream_t st) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, counts, xor_shards, mn_total, km_total, km_capacity,
                       mn_capacity);
    return hipGetLastError();
}


// ---- runs per read (HpcSimd tail rule needs the run count of the whole read) ---------------------------------
// neq(q) = q == 0 || s[q] != s[q-1].  Pass 1 counts neq over 256-byte blocks of the stream, a scan turns the counts
// into prefixes, pass 2 evaluates the prefix C at both ends of every read (at most 255 bytes each, 16 lanes per read) and
// runs = C(end) - C(start) + (neq(start) ? 0 : 1): a read start begins a run even inside a homopolymer.
namespace {
constexpr int RUN_BLK = 256;

// Four 4 KiB stretches per block (16 bytes per thread each, all four loads in flight); the byte before a thread's 16 comes from the lane before it
// (its last byte), only lane 0 of a wave fetches it from memory (round 6: one vector load per thread and stretch instead of two, 4 in flight instead of 1)
constexpr int RUN_UNROLL = 4;
__global__ __launch_bounds__(256) void run_block_counts(const uint8_t *__restrict__ s, uint64_t n, uint32_t *__restrict__ cnt, bool rle) {
    const int lane = threadIdx.x & 63;
    uint64_t q0[RUN_UNROLL];
    uint4 v[RUN_UNROLL];
    uint32_t pm[RUN_UNROLL];
#pragma unroll
    for (int k = 0; k < RUN_UNROLL; k++) {
        q0[k] = (((uint64_t)blockIdx.x * RUN_UNROLL + k) * 256 + threadIdx.x) * 16; // 16 bytes per thread, 16 threads per block of 256
        v[k] = make_uint4(0, 0, 0, 0);
        pm[k] = 0x100u;
        if (q0[k] + 16 <= n && !rle) v[k] = *reinterpret_cast<const uint4 *>(s + q0[k]);
        if (q0[k] < n && q0[k] && (lane == 0 || rle)) pm[k] = s[q0[k] - 1];
    }
#pragma unroll
    for (int k = 0; k < RUN_UNROLL; k++) {
        uint32_t c = 0;
        if (q0[k] < n) {
            // the lane before holds the 16 bytes that end where this lane's begin (it lies inside the stream: fully loaded)
            const uint32_t from_lane = (uint32_t)__shfl_up((int)(v[k].w >> 24), 1);
            uint32_t prev = (lane == 0 || rle) ? pm[k] : from_lane;
            if (q0[k] + 16 <= n && !rle) {
                const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t prv = (w[i] << 8) | (prev & 0xFFu);
                    uint32_t x = w[i] ^ prv; // byte j != 0 <=> s[q] != s[q-1]
                    uint32_t nz = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu | x) & 0x80808080u;
                    if (i == 0 && prev == 0x100u) nz |= 0x80u; // q == 0
                    c += __popc(nz);
                    prev = w[i] >> 24;
                }
            } else {
                if (!rle && q0[k]) prev = s[q0[k] - 1]; // (the stream's last, partial 16 bytes: the lane before may not have loaded)
                for (uint64_t q = q0[k]; q < n && q < q0[k] + 16; q++) { // tail of the stream, and the encode_rle flavour
                    const uint32_t b = s[q];
                    c += (prev == 0x100u || run_head(b, prev, rle)) ? 1u : 0u;
                    prev = b;
                }
            }
        }
        // 16 consecutive threads share a block
        c += __shfl_xor(c, 1);
        c += __shfl_xor(c, 2);
        c += __shfl_xor(c, 4);
        c += __shfl_xor(c, 8);
        const uint64_t blk = q0[k] / RUN_BLK;
        if ((threadIdx.x & 15) == 0 && blk * RUN_BLK < n + RUN_BLK) cnt[blk] = c;
    }
}

// Run heads in [start of b's 256-byte block, b) that fall into the 16 bytes lane `sub` of a 16-lane group looks at; the group's sum
// plus blk_off[block] is the prefix C(b).  (One thread per read walked up to 255 bytes, byte load after byte load, three times:
// 2.1 ms per million reads -- as long as the pass over the 10 GB that makes the block counts.)
__device__ inline uint32_t run_prefix_part(const uint8_t *__restrict__ s, uint64_t n, uint64_t b, int sub, bool rle) {
    const uint64_t q0 = (b / RUN_BLK) * RUN_BLK + 16u * (uint32_t)sub;
    if (q0 >= b) return 0;
    const uint32_t lim = b - q0 < 16 ? (uint32_t)(b - q0) : 16u; // bytes of this lane that lie before b
    uint32_t prev = q0 ? s[q0 - 1] : 0x100u;
    uint32_t c = 0;
    if (q0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s + q0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t cur = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            c += ((uint32_t)i < lim && (prev == 0x100u || run_head(cur, prev, rle))) ? 1u : 0u;
            prev = cur;
        }
    } else { // last bytes of the stream
        for (uint32_t i = 0; i < lim; i++) {
            const uint32_t cur = s[q0 + i];
            c += (prev == 0x100u || run_head(cur, prev, rle)) ? 1u : 0u;
            prev = cur;
        }
    }
    return c;
}

// 16 lanes per read
__global__ __launch_bounds__(256) void read_run_counts(const uint8_t *__restrict__ s, uint64_t n, const uint64_t *__restrict__ read_off,
                                                       uint64_t n_reads, const uint64_t *__restrict__ blk_off, uint32_t *__restrict__ runs,
                                                       uint64_t *__restrict__ read_c0, bool rle) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = gid >> 4;
    const int sub = (int)(gid & 15);
    const bool live = r < n_reads; // whole groups of 16 are live or not: the shuffles below stay inside a group
    uint64_t a = 0, b = 0;
    if (live) { // entries of a malformed table (see validate_read_off_kernel) must not become out-of-bounds reads
        a = read_off[r];
        b = read_off[r + 1];
        if (a > n) a = n;
        if (b > n) b = n;
        if (b < a) b = a;
    }
    uint32_t pa = 0, pb = 0;
    if (live && b > a) {
        pa = run_prefix_part(s, n, a, sub, rle);
        pb = run_prefix_part(s, n, b, sub, rle);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        pa += __shfl_xor(pa, o);
        pb += __shfl_xor(pb, o);
    }
    if (!live || sub != 0) return;
    uint32_t R = 0;
    uint64_t c0 = 0;
    if (b > a) {
        const uint64_t ca = blk_off[a / RUN_BLK] + pa, cb = blk_off[b / RUN_BLK] + pb;
        const bool neq_a = a == 0 || run_head(s[a], s[a - 1], rle);
        R = (uint32_t)(cb - ca) + (neq_a ? 0u : 1u);
        c0 = ca;
    }
    if (read_c0) read_c0[r] = c0;
    runs[r] = R;
}
} // namespace

namespace {
__global__ __launch_bounds__(256) void validate_read_off_kernel(const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases,
                                                                uint32_t *__restrict__ bad) {
    uint32_t f = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride) {
        const uint64_t a = read_off[r], b = read_off[r + 1];
        if (b < a) f |= BAD_ORDER;
        else if (b - a > 0xFFFFFFFEull) f |= BAD_LONG;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (read_off[0] != 0) f |= BAD_FIRST;
        if (read_off[n_reads] != n_bases) f |= BAD_END;
    }
    if (f) atomicOr(bad, f);
}
} // namespace

hipError_t launch_validate_read_off(const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, uint32_t *bad, hipStream_t st) {
    uint64_t blocks = (n_reads + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(validate_read_off_kernel, dim3((unsigned)blocks), dim3(256), 0, st, read_off, n_reads, n_bases, bad);
    return hipGetLastError();
}

namespace {
// One pass over the read table in front of the tiled kernels (one thread per read, O(1) loads each -- the kernel it replaces searched the table
// once per TILE, twenty dependent loads per thread):
//  * the checks of validate_read_off_kernel;
//  * tile_read0[t] = the read that holds base t * TILE_BASES, i.e. the LAST read r with read_off[r] <= t * TILE_BASES (of reads that share an
//    offset -- empty ones, then at most one that is not -- only the last holds a position), written by that read's own thread: a read
//    [a, b) owns the tiles ceil(a / TILE) .. (b - 1) / TILE.  A read of more than eight tiles (contigs) is spread over the lanes of its wave.
//    Entry n_tiles (the position n_bases) is n_reads - 1.  A malformed table leaves some entries unwritten; nothing reads them then (BAD_*).
//  * descriptor path: every tile's word starts as word0 (the identity of the scan, see s2k_api.hip).
__global__ __launch_bounds__(256) void read_table_kernel(const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases, uint64_t n_tiles,
                                                         uint32_t *__restrict__ bad, uint32_t *__restrict__ tile_read0,
                                                         unsigned long long *__restrict__ tile_words, unsigned long long word0) {
    uint32_t f = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint64_t n_round = (n_reads + 63) & ~63ull; // whole waves take every trip of the loop: the ballots below need them
    for (uint64_t r = gid; r < n_round; r += stride) {
        uint64_t t0 = 0, cnt = 0;
        if (r < n_reads) {
            const uint64_t a = read_off[r], b = read_off[r + 1];
            if (b < a) f |= BAD_ORDER;
            else if (b - a > 0xFFFFFFFEull) f |= BAD_LONG;
            if (b > a && b <= n_bases) { // (clamped: a malformed table must not become an out-of-bounds store)
                t0 = (a + TILE_BASES - 1) / TILE_BASES;
                const uint64_t t1 = (b - 1) / TILE_BASES; // t1 < n_tiles since b <= n_bases
                cnt = t1 >= t0 ? t1 - t0 + 1 : 0;
            }
        }
        if (cnt <= 8)
            for (uint64_t i = 0; i < cnt; i++) tile_read0[t0 + i] = (uint32_t)r;
        uint64_t big = __ballot(cnt > 8);
        while (big) { // wave-uniform
            const int z = __builtin_ctzll(big);
            big &= big - 1;
            const uint64_t zt0 = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(t0 >> 32), z) << 32) | (uint32_t)__shfl((int)(uint32_t)t0, z);
            const uint64_t zc = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(cnt >> 32), z) << 32) | (uint32_t)__shfl((int)(uint32_t)cnt, z);
            const uint32_t zr = (uint32_t)(r - (uint64_t)lane + (uint64_t)z); // lanes of a wave hold consecutive reads
            for (uint64_t i = lane; i < zc; i += 64) tile_read0[zt0 + i] = zr;
        }
    }
    if (gid == 0) {
        if (read_off[0] != 0) f |= BAD_FIRST;
        if (read_off[n_reads] != n_bases) f |= BAD_END;
        tile_read0[n_tiles] = (uint32_t)(n_reads - 1);
    }
    if (f) atomicOr(bad, f);
    if (tile_words)
        for (uint64_t t = gid; t < n_tiles; t += stride) tile_words[t] = word0;
}
} // namespace

hipError_t launch_read_table(const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, uint64_t n_tiles, uint32_t *bad, uint32_t *tile_read0,
                             unsigned long long *tile_words, unsigned long long word0, hipStream_t st) {
    if (n_reads == 0) return launch_validate_read_off(read_off, n_reads, n_bases, bad, st); // (no tiles: only the two end checks)
    const uint64_t work = n_reads > n_tiles ? n_reads : n_tiles;
    uint64_t blocks = (work + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(read_table_kernel, dim3((unsigned)blocks), dim3(256), 0, st, read_off, n_reads, n_bases, n_tiles, bad, tile_read0, tile_words, word0);
    return hipGetLastError();
}

hipError_t launch_read_run_counts(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases,
                                  uint32_t *blk_cnt, uint64_t *blk_off, uint64_t *scan_tmp, uint32_t *runs, uint64_t *read_c0,
                                  hipStream_t st, bool rle) {
    if (n_reads == 0) return hipSuccess;
    const uint64_t nblk = n_bases / RUN_BLK + 1; // the block that holds position n_bases exists too (count 0 past the end)
    const uint64_t threads = nblk * 16;
    hipLaunchKernelGGL(run_block_counts, dim3((unsigned)((threads + 256 * RUN_UNROLL - 1) / (256 * RUN_UNROLL))), dim3(256), 0, st, bases, n_bases, blk_cnt, rle);
    hipError_t e = launch_scan_u32(blk_cnt, nblk, blk_off, scan_tmp, 0, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(read_run_counts, dim3((unsigned)((n_reads * 16 + 255) / 256)), dim3(256), 0, st, bases, n_bases, read_off, n_reads,
                       blk_off, runs, read_c0, rle);
    return hipGetLastError();
}

} // namespace s2k
