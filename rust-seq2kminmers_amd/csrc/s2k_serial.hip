// s2k_serial.hip -- read-serial kernels: one lane walks one read, exactly like the reference's
// scalar iterators.  They are the engine's *exact fallback* (any byte values, any l < 256, all four
// HashMode result semantics) and a second, structurally independent GPU implementation used to
// cross-check the tiled kernels at full size.  They are NOT the fast path: byte loads are
// uncoalesced and lanes diverge.  The fast path is s2k_tile.hip.
//
// Two-pointer formulation instead of the reference's 256-entry ring buffers
// (src/nthash_hpc.rs:107-109): `po` walks the run heads of the outgoing base, `pi` the run heads l
// further, so the outgoing seed is re-read from the sequence rather than remembered.
#include "s2k_dev.h"

namespace s2k {
namespace {

struct SeedsScalar {
    static __device__ inline uint32_t h(uint32_t c) { return seed_h_scalar(c); }
    static __device__ inline uint32_t rc(uint32_t c) { return seed_rc_scalar(c); }
};
struct SeedsSimd {
    static __device__ inline uint32_t h(uint32_t c) { return seed_h_simd(c); }
    static __device__ inline uint32_t rc(uint32_t c) { return seed_rc_simd(c); }
};

// next run head after raw position q (HPC) or q+1 (raw space); n if none
template <bool HPC>
__device__ inline uint64_t next_head(const uint8_t *__restrict__ s, uint64_t n, uint64_t q) {
    uint64_t r = q + 1;
    if (HPC) {
        uint8_t c = s[q];
        while (r < n && s[r] == c) r++;
    }
    return r;
}

// Walks one read and calls emit(j, jend, hash) for every minimizer in iterator order.
//   Regular : src/lib.rs:215-230 (p in [0,n-l], <=, jend=j+l-1)
//   Hpc     : src/nthash_hpc.rs:115-283 (runs; p in [0,R-l-1]; jend = st[p+l]-1)
//   Simd    : src/nthash_avx512_32.rs:32-164 (<, f32 bound, nibble seeds, tail quirk)
//   HpcSimd : src/nthash_hpc_simd.rs:35-68 (jend = st[p+l-1], last l-mer kept)
template <bool HPC, class Seeds, class Emit>
__device__ inline void walk_read(const uint8_t *__restrict__ s, uint64_t n, const Sem &sem, Emit emit) {
    const uint32_t l = sem.l;
    if (n <= l || !sem.enabled) return; // src/lib.rs:97
    uint64_t limit = ~0ull;             // number of l-mers that may be emitted
    if (sem.tail_quirk) {
        uint64_t m = n;
        if (HPC) { // R = number of runs
            m = 1;
            for (uint64_t q = 1; q < n; q++) m += (s[q] != s[q - 1]);
        }
        if (m < l) return;
        uint64_t sentinel = m - l + 1;
        limit = (sentinel % 16 == 0 && sentinel >= 32) ? sentinel - 16 : sentinel; // nthash_avx512_32.rs:134-138
    }
    uint32_t fh = 0, rh = 0;
    uint64_t q = 0, pl = 0;
    for (uint32_t i = 0; i < l; i++) {
        if (q >= n) return; // fewer than l run heads
        uint32_t c = s[q];
        fh = rotl32(fh, 1) ^ Seeds::h(c);
        rh = rotr32(rh, 1) ^ rotl32(Seeds::rc(c), l - 1);
        pl = q;
        q = next_head<HPC>(s, n, q);
    }
    uint64_t po = 0, pi = q; // po = st[p], pl = st[p+l-1], pi = st[p+l] (or n)
    for (uint64_t p = 0; p < limit; p++) {
        if (!sem.keep_last && pi >= n) break; // src/nthash_hpc.rs:265-267: end check precedes the bound test
        uint32_t hv = fh < rh ? fh : rh;
        if (hv <= sem.bound_le) {
            uint64_t je = sem.end_kind == 0 ? po + l - 1 : sem.end_kind == 1 ? pi - 1 : pl;
            emit((uint32_t)po, (uint32_t)je, hv);
        }
        if (pi >= n) break;
        uint32_t co = s[po], ci = s[pi];
        fh = rotl32(fh, 1) ^ rotl32(Seeds::h(co), l) ^ Seeds::h(ci);
        rh = rotr32(rh, 1) ^ rotr32(Seeds::rc(co), 1) ^ rotl32(Seeds::rc(ci), l - 1);
        po = next_head<HPC>(s, n, po);
        pl = pi;
        pi = next_head<HPC>(s, n, pi);
    }
}

// A caller's device-resident read table is validated by a kernel earlier in the stream (validate_read_off_kernel) whose
// verdict the host sees only afterwards: until then a malformed entry must not turn into an out-of-bounds read.
__device__ inline void clamp_read(uint64_t &a, uint64_t &b, uint64_t n_bases) {
    if (a > n_bases) a = n_bases;
    if (b > n_bases) b = n_bases;
    if (b < a) b = a;
    if (b - a > 0xFFFFFFFEull) b = a + 0xFFFFFFFEull;
}

template <bool HPC, class Seeds>
__global__ __launch_bounds__(64) void serial_count_kernel(const uint8_t *__restrict__ bases,
                                                          const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases,
                                                          Sem sem, uint32_t *__restrict__ mn_cnt) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t a = read_off[r], b = read_off[r + 1];
    clamp_read(a, b, n_bases);
    uint32_t c = 0;
    walk_read<HPC, Seeds>(bases + a, b - a, sem, [&](uint32_t, uint32_t, uint32_t) { c++; });
    mn_cnt[r] = c;
}

template <bool HPC, class Seeds>
__global__ __launch_bounds__(64) void serial_write_kernel(const uint8_t *__restrict__ bases,
                                                          const uint64_t *__restrict__ read_off, uint64_t n_reads, uint64_t n_bases,
                                                          Sem sem, const uint64_t *__restrict__ mn_off, Records rec,
                                                          Counts *counts) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t a = read_off[r], b = read_off[r + 1];
    clamp_read(a, b, n_bases);
    uint64_t o = mn_off[r];
    if (r == 0 && mn_off[n_reads] > rec.capacity) {
        counts->pool_overflow = 1;
        counts->pool_needed = mn_off[n_reads];
    }
    walk_read<HPC, Seeds>(bases + a, b - a, sem, [&](uint32_t j, uint32_t je, uint32_t hv) {
        if (o < rec.capacity) {
            rec.j[o] = j;
            rec.jend[o] = je;
            rec.hash[o] = hv;
            rec.rid[o] = (uint32_t)r;
        }
        o++;
    });
}

// ---- standalone homopolymer compression: hpc() src/hpc.rs:28-41 / encode_rle_simd src/hpc.rs:44-147 ----
__global__ __launch_bounds__(64) void hpc_count_kernel(const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ read_off, uint64_t n_reads,
                                                       uint32_t *__restrict__ run_cnt, bool rle) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t a = read_off[r], b = read_off[r + 1];
    uint32_t c = 0;
    for (uint64_t q = a; q < b; q++) c += (q == a || run_head(bases[q], bases[q - 1], rle));
    run_cnt[r] = c;
}

__global__ __launch_bounds__(64) void hpc_write_kernel(const uint8_t *__restrict__ bases,
                                                       const uint64_t *__restrict__ read_off, uint64_t n_reads,
                                                       const uint64_t *__restrict__ hpc_off, uint8_t *__restrict__ o_hpc,
                                                       uint32_t *__restrict__ o_pos, uint64_t capacity, bool rle) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t a = read_off[r], b = read_off[r + 1];
    uint64_t o = hpc_off[r];
    for (uint64_t q = a; q < b; q++)
        if (q == a || run_head(bases[q], bases[q - 1], rle)) {
            if (o < capacity) {
                if (o_hpc) o_hpc[o] = bases[q];
                if (o_pos) o_pos[o] = (uint32_t)(q - a);
            }
            o++;
        }
}

} // namespace

hipError_t launch_serial_count(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, Sem sem,
                               uint32_t *mn_cnt, hipStream_t st) {
    if (n_reads == 0) return hipSuccess;
    dim3 g((unsigned)((n_reads + 63) / 64)), b(64);
    if (sem.hpc) {
        if (sem.simd_seeds) hipLaunchKernelGGL((serial_count_kernel<true, SeedsSimd>), g, b, 0, st, bases, read_off, n_reads, n_bases, sem, mn_cnt);
        else hipLaunchKernelGGL((serial_count_kernel<true, SeedsScalar>), g, b, 0, st, bases, read_off, n_reads, n_bases, sem, mn_cnt);
    } else {
        if (sem.simd_seeds) hipLaunchKernelGGL((serial_count_kernel<false, SeedsSimd>), g, b, 0, st, bases, read_off, n_reads, n_bases, sem, mn_cnt);
        else hipLaunchKernelGGL((serial_count_kernel<false, SeedsScalar>), g, b, 0, st, bases, read_off, n_reads, n_bases, sem, mn_cnt);
    }
    return hipGetLastError();
}

hipError_t launch_serial_write(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint64_t n_bases, Sem sem,
                               const uint64_t *mn_off, Records rec, Counts *counts, hipStream_t st) {
    if (n_reads == 0) return hipSuccess;
    dim3 g((unsigned)((n_reads + 63) / 64)), b(64);
    if (sem.hpc) {
        if (sem.simd_seeds) hipLaunchKernelGGL((serial_write_kernel<true, SeedsSimd>), g, b, 0, st, bases, read_off, n_reads, n_bases, sem, mn_off, rec, counts);
        else hipLaunchKernelGGL((serial_write_kernel<true, SeedsScalar>), g, b, 0, st, bases, read_off, n_reads, n_bases, sem, mn_off, rec, counts);
    } else {
        if (sem.simd_seeds) hipLaunchKernelGGL((serial_write_kernel<false, SeedsSimd>), g, b, 0, st, bases, read_off, n_reads, n_bases, sem, mn_off, rec, counts);
        else hipLaunchKernelGGL((serial_write_kernel<false, SeedsScalar>), g, b, 0, st, bases, read_off, n_reads, n_bases, sem, mn_off, rec, counts);
    }
    return hipGetLastError();
}

hipError_t launch_hpc_count(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, uint32_t *run_cnt, hipStream_t st, bool rle) {
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(hpc_count_kernel, dim3((unsigned)((n_reads + 63) / 64)), dim3(64), 0, st, bases, read_off, n_reads, run_cnt, rle);
    return hipGetLastError();
}
hipError_t launch_hpc_write(const uint8_t *bases, const uint64_t *read_off, uint64_t n_reads, const uint64_t *hpc_off,
                            uint8_t *o_hpc, uint32_t *o_pos, uint64_t capacity, hipStream_t st, bool rle) {
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(hpc_write_kernel, dim3((unsigned)((n_reads + 63) / 64)), dim3(64), 0, st, bases, read_off, n_reads, hpc_off, o_hpc, o_pos, capacity, rle);
    return hipGetLastError();
}

} // namespace s2k
