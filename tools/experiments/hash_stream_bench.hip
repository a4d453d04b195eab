// hash_stream_bench.hip -- the Regular hash loop WITHOUT a tile buffer: every lane reads its 144 + l bytes of the tile straight from global
// memory, 16 bytes at a time, two pieces ahead of their use (the loop of s2k_tile_impl.h with a global source pointer), over a stream
// far larger than every cache; blocks of HS_TW waves, HS_BPC blocks per CU -> HS_TW * HS_BPC / 4 waves per SIMD.  Question: does the
// strided access (64 lanes x 16 B, 144 B apart, every 128-byte line touched by eight or nine different instructions of the wave) keep
// up when the occupancy is no longer bounded by 9 KB of LDS per wave?
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I rust-seq2kminmers_amd/csrc -DS2K_TW=<waves per block> -DHS_BPC=<blocks per CU> ...
#include "s2k_tile_impl.h"
#include <stdio.h>
#include <vector>

using namespace s2k;
constexpr int TW = S2K_TW; // (left s2k_tile_impl.h in round 5: tw<HPC>())
#ifndef HS_BPC
#define HS_BPC 1
#endif

template <int L, int LA>
__global__ __launch_bounds__(64 * TW, (TW * HS_BPC + 3) / 4) void hs_kernel(const uint8_t *__restrict__ bases, uint64_t n_tiles, uint32_t *out, uint64_t *cyc,
                                                                            uint32_t bound) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint2 *tab = reinterpret_cast<uint2 *>(smem);
    const int lane0 = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int c = threadIdx.x; c < 256; c += 64 * TW) {
        uint32_t h = seed_h_scalar(c), r = seed_rc_scalar(c);
        tab[c] = make_uint2(h, rotl32(r, L - 1));
        tab[256 + c] = make_uint2(rotl32(h, L), rotr32(r, 1));
    }
    __syncthreads();
    uint32_t acc = 0;
    const uint64_t n_waves = (uint64_t)gridDim.x * TW;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t done = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * TW + w; t < n_tiles; t += n_waves, done++) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        uint32_t caps[NPC], raw[5];
#pragma unroll
        for (int g = 0; g < NPC; g++) caps[g] = 0;
#pragma unroll
        for (int g = 0; g < 5; g++) raw[g] = 0;
        hash_loop_static<L, LA, false>(bases + t * (uint64_t)TILE_BASES, bound, lane, 9, caps, raw);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NPC; g++) acc ^= caps[g];
#pragma unroll
        for (int g = 0; g < 5; g++) acc += raw[g];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 * TW + threadIdx.x] = acc;
    if (lane0 == 0) {
        cyc[3 * (blockIdx.x * TW + w)] = t1 - t0;
        cyc[3 * (blockIdx.x * TW + w) + 1] = r1 - r0;
        cyc[3 * (blockIdx.x * TW + w) + 2] = done;
    }
}

__global__ void fill_kernel(uint8_t *d, uint64_t n) {
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i + 16 > n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    uint32_t wv[4];
    for (int k = 0; k < 4; k++) {
        uint32_t v = 0;
        for (int b = 0; b < 4; b++) v |= (uint32_t)"ACGT"[(x >> (2 * (4 * k + b))) & 3] << (8 * b);
        wv[k] = v;
    }
    *reinterpret_cast<uint4 *>(d + i) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const uint64_t n_tiles = 500000, n_bytes = n_tiles * TILE_BASES + 4096; // 4.6 GB
    uint8_t *d_b;
    if (hipMalloc(&d_b, n_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n_bytes / 16 + 255) / 256)), dim3(256), 0, 0, d_b, n_bytes);
    const int blocks = n_cu * HS_BPC;
    uint32_t *d_out;
    uint64_t *d_cyc;
    hipMalloc(&d_out, (size_t)blocks * 64 * TW * 4);
    hipMalloc(&d_cyc, (size_t)blocks * TW * 24);
    auto k = hs_kernel<31, 1>;
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * TW), TABLE_BYTES, 0, d_b, n_tiles, d_out, d_cyc, 42949672u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h((size_t)blocks * TW * 3);
        hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double cy = 0, rt = 0, dn = 0;
        for (size_t i = 0; i < h.size() / 3; i++) { cy += (double)h[3 * i]; rt += (double)h[3 * i + 1]; dn += (double)h[3 * i + 2]; }
        const double wps = TW * HS_BPC / 4.0;
        printf("%d x %d waves per CU (%.1f per SIMD): %.3f ms for %.2f Gbp -> %.0f Gbp/s; %.0f cyc/tile/wave, %.0f cyc/tile/SIMD, %.2f cyc/position/SIMD, clock %.0f MHz\n", HS_BPC, TW,
               wps, ms, n_tiles * (double)TILE_BASES / 1e9, n_tiles * (double)TILE_BASES / ms / 1e6, cy / dn, cy / dn / wps, cy / dn / wps / 144.0, cy / rt * 100.0);
    }
    return 0;
}
