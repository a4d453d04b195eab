// s2k_count.hip -- downstream of the path: how often does every k-min-mer hash occur.
// What it stands in for: the consumer of KminmersIterator in rust-mdbg keeps the items in a concurrent map keyed by the
// k-min-mer hash (the reference only hints at it: "DashMap" in the comment at src/lib.rs:256-257; KminmerHash is Eq/Ord/Hash by
// `hash` alone, src/kminmer.rs:181-204).  Nothing of this is in the reference crate, so the oracle is a dictionary over
// the oracle's hashes (tests/test_count.py).
//
// Two device ops, both hand-written (no rocPRIM):
//  * s2k_count_device: open-addressing table in HBM (linear probing on a table of 1.5 n slots -- any size: the first slot
//    is mulhi(mixed key, slots) -- 64-bit atomicCAS for the key, 32-bit atomicAdd for the count), then an unordered
//    compaction of the occupied slots.  The table lives in a grow-only buffer of the context (no hipMalloc / hipFree,
//    i.e. no device-wide synchronisation, per call).
//  * s2k_partition_device: split the keys into n_parts ranges of the hash space (part = mulhi(hash, n_parts), i.e. by
//    hash PREFIX) -- the send buffers of the one real exchange step of the multi-GPU version: an all-to-all after which
//    rank p owns every occurrence of the hashes in its range and counts locally (rust-seq2kminmers_amd/sharding.py).
#include "../../include/s2k.h"
#include "s2k_dev.h"
#include "s2k_hostcopy.h"

namespace s2k {
namespace {

constexpr uint64_t EMPTY = ~0ull; // a key equal to EMPTY is counted in a side counter

__device__ inline uint64_t slot_hash(uint64_t k) { // the keys are hashes already, but canonical minima crowd the low range
    k ^= k >> 32;
    k *= 0x9E3779B97F4A7C15ull;
    return k ^ (k >> 29);
}

__global__ __launch_bounds__(256) void count_insert_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint64_t *tkeys,
                                                           uint32_t *tcnt, uint64_t slots, uint32_t *n_empty_key) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        if (k == EMPTY) {
            atomicAdd(n_empty_key, 1u);
            continue;
        }
        uint64_t s = __umul64hi(slot_hash(k), slots);
        for (uint64_t probe = 0; probe < slots; probe++) { // the table has >= 1.5 n slots: terminates long before
            const uint64_t prev = atomicCAS((unsigned long long *)&tkeys[s], (unsigned long long)EMPTY, (unsigned long long)k);
            if (prev == EMPTY || prev == k) {
                atomicAdd(&tcnt[s], 1u);
                break;
            }
            s = s + 1 == slots ? 0 : s + 1;
        }
    }
}

__global__ __launch_bounds__(256) void count_compact_kernel(const uint64_t *__restrict__ tkeys, const uint32_t *__restrict__ tcnt,
                                                            uint64_t slots, uint64_t *__restrict__ o_keys, uint32_t *__restrict__ o_cnt,
                                                            uint64_t capacity, unsigned long long *cursor, const uint32_t *n_empty_key) {
    __shared__ unsigned long long base;
    __shared__ uint32_t warp_cnt[4];
    for (uint64_t s0 = (uint64_t)blockIdx.x * 256; s0 < slots; s0 += (uint64_t)gridDim.x * 256) {
        const uint64_t s = s0 + threadIdx.x;
        const bool occ = s < slots && tkeys[s] != EMPTY;
        const uint64_t m = __ballot(occ);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) warp_cnt[w] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) base = atomicAdd(cursor, (unsigned long long)(warp_cnt[0] + warp_cnt[1] + warp_cnt[2] + warp_cnt[3]));
        __syncthreads();
        uint32_t before = 0;
        for (int j = 0; j < w; j++) before += warp_cnt[j];
        if (occ) {
            const uint64_t o = base + before + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
            if (o < capacity) {
                if (o_keys) o_keys[o] = tkeys[s];
                if (o_cnt) o_cnt[o] = tcnt[s];
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && *n_empty_key) { // the one key the table cannot hold
        const uint64_t o = atomicAdd(cursor, 1ull);
        if (o < capacity) {
            if (o_keys) o_keys[o] = EMPTY;
            if (o_cnt) o_cnt[o] = *n_empty_key;
        }
    }
}

__device__ inline uint32_t part_of(uint64_t k, uint32_t n_parts) { return (uint32_t)__umul64hi(k, (uint64_t)n_parts); }

__global__ __launch_bounds__(256) void part_hist_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint32_t n_parts,
                                                        unsigned long long *hist) {
    __shared__ uint32_t h[S2K_MAX_PARTS];
    for (int i = threadIdx.x; i < (int)n_parts; i += 256) h[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) atomicAdd(&h[part_of(keys[i], n_parts)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < (int)n_parts; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}
__global__ void part_scan_kernel(const unsigned long long *hist, uint32_t n_parts, uint64_t *part_off, unsigned long long *cursor) {
    uint64_t acc = 0;
    for (uint32_t p = 0; p < n_parts; p++) {
        part_off[p] = acc;
        cursor[p] = acc;
        acc += hist[p];
    }
    part_off[n_parts] = acc;
}
__global__ __launch_bounds__(256) void part_scatter_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint32_t n_parts,
                                                           unsigned long long *cursor, uint64_t *__restrict__ out) {
    // block-local counting sort of 256 keys, then one reservation per (block, part): keeps the global atomics rare
    __shared__ uint32_t h[S2K_MAX_PARTS];
    __shared__ unsigned long long basep[S2K_MAX_PARTS];
    for (uint64_t i0 = (uint64_t)blockIdx.x * 256; i0 < n; i0 += (uint64_t)gridDim.x * 256) {
        for (int i = threadIdx.x; i < (int)n_parts; i += 256) h[i] = 0;
        __syncthreads();
        const uint64_t i = i0 + threadIdx.x;
        uint64_t k = 0;
        uint32_t p = 0, r = 0;
        if (i < n) {
            k = keys[i];
            p = part_of(k, n_parts);
            r = atomicAdd(&h[p], 1u);
        }
        __syncthreads();
        for (int j = threadIdx.x; j < (int)n_parts; j += 256)
            if (h[j]) basep[j] = atomicAdd(&cursor[j], (unsigned long long)h[j]);
        __syncthreads();
        if (i < n) out[basep[p] + r] = k;
        __syncthreads();
    }
}

} // namespace
} // namespace s2k

using namespace s2k;

extern "C" {

s2k_status s2k_count_device(s2k_ctx *ctx, const uint64_t *d_hash, uint64_t n, uint64_t *d_keys, uint32_t *d_counts,
                            uint64_t capacity, uint64_t *n_distinct) {
    if (!ctx || (!d_hash && n) || !n_distinct) return S2K_ERR_INVALID_ARG;
    hipStream_t st = ctx_stream(ctx);
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return S2K_ERR_DEVICE;
    const uint64_t slots = (n + n / 2 + 1024 + 255) & ~(uint64_t)255; // load factor <= 2/3
    const size_t bytes = slots * 12 + 64;
    void *ws = ctx_count_table(ctx, bytes);
    if (!ws) {
        ctx_set_error(ctx, "s2k_count_device: table allocation");
        return S2K_ERR_NOMEM;
    }
    uint64_t *tkeys = (uint64_t *)ws;
    uint32_t *tcnt = (uint32_t *)((char *)ws + slots * 8);
    unsigned long long *cursor = (unsigned long long *)((char *)ws + slots * 12);
    uint32_t *n_empty = (uint32_t *)((char *)ws + slots * 12 + 8);
    hipError_t e = hipMemsetAsync(tkeys, 0xFF, slots * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(tcnt, 0, slots * 4 + 64, st);
    if (e == hipSuccess && n) {
        const uint64_t blocks = (n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536;
        hipLaunchKernelGGL(count_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_hash, n, tkeys, tcnt, slots, n_empty);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        const uint64_t blocks = slots / 256 < 65536 ? slots / 256 : 65536;
        hipLaunchKernelGGL(count_compact_kernel, dim3((unsigned)blocks), dim3(256), 0, st, tkeys, tcnt, slots, d_keys, d_counts, capacity,
                           cursor, n_empty);
        e = hipGetLastError();
    }
    unsigned long long total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total, cursor, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        ctx_set_error(ctx, "s2k_count_device: HIP error");
        return S2K_ERR_DEVICE;
    }
    *n_distinct = total;
    return (total > capacity && (d_keys || d_counts)) ? S2K_ERR_CAPACITY : S2K_OK;
}

s2k_status s2k_partition_device(s2k_ctx *ctx, const uint64_t *d_hash, uint64_t n, uint32_t n_parts, uint64_t *d_out,
                                uint64_t *d_part_off) {
    if (!ctx || (!d_hash && n) || (!d_out && n) || !d_part_off || n_parts == 0 || n_parts > S2K_MAX_PARTS) return S2K_ERR_INVALID_ARG;
    hipStream_t st = ctx_stream(ctx);
    if (hipSetDevice(ctx_device(ctx)) != hipSuccess) return S2K_ERR_DEVICE;
    unsigned long long *ws = (unsigned long long *)ctx_count_table(ctx, 2 * S2K_MAX_PARTS * 8);
    if (!ws) {
        ctx_set_error(ctx, "s2k_partition_device: workspace allocation");
        return S2K_ERR_NOMEM;
    }
    hipError_t e = hipMemsetAsync(ws, 0, 2 * S2K_MAX_PARTS * 8, st);
    const uint64_t blocks = n ? ((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384) : 1;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(part_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_hash, n, n_parts, ws);
        hipLaunchKernelGGL(part_scan_kernel, dim3(1), dim3(1), 0, st, ws, n_parts, d_part_off, ws + S2K_MAX_PARTS);
        hipLaunchKernelGGL(part_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_hash, n, n_parts, ws + S2K_MAX_PARTS, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        ctx_set_error(ctx, "s2k_partition_device: HIP error");
        return S2K_ERR_DEVICE;
    }
    return S2K_OK;
}

} // extern "C"
