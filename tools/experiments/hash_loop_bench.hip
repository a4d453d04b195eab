// hash_loop_bench.hip -- the tiled kernel's hash loop ALONE (the real code: hash_loop_static from s2k_tile_impl.h), one block of
// TW waves per CU over LDS-resident pseudo-random ACGT, timed with s_memtime: shader cycles per tile of 9216 positions per SIMD.
// What it answers: how many cycles a hash step costs when nothing else runs beside it, against the
// one-instruction loops of valu_rate.hip and against what the step costs inside the whole kernel.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I rust-seq2kminmers_amd/csrc [-DHX_...] tools/experiments/hash_loop_bench.hip -o hash_loop_bench
#include "s2k_tile_impl.h"
#include <stdio.h>
#include <vector>

using namespace s2k;
// (TW / S2K_WAVES_PER_SIMD left s2k_tile_impl.h in round 5 -- tw<HPC>() / waves_per_simd<HPC>() --: the bench keeps its own, -DS2K_TW=<waves per block>)
constexpr int TW = S2K_TW;
#ifndef S2K_WAVES_PER_SIMD
#define S2K_WAVES_PER_SIMD ((S2K_TW + 3) / 4)
#endif
#ifndef HX_BPC
#define HX_BPC 1 // blocks per CU (more than 16 waves per CU need more than one block)
#endif

template <int L, int LA, bool HPC>
__global__ __launch_bounds__(64 * TW, S2K_WAVES_PER_SIMD) void hl_kernel(uint32_t *out, uint64_t *cyc, int iters, uint32_t bound, int np, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using WL = WaveLdsT<HPC>;
    uint2 *tab = reinterpret_cast<uint2 *>(smem);
    const int lane0 = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int c = threadIdx.x; c < 256; c += 64 * TW) {
        uint32_t h = seed_h_scalar(c), r = seed_rc_scalar(c);
        tab[c] = make_uint2(h, rotl32(r, L - 1));
        tab[256 + c] = make_uint2(rotl32(h, L), rotr32(r, 1));
    }
#ifdef HX_SHARE // (occupancy experiment: more waves than tile buffers fit -- waves w and w + HX_SHARE read the same buffer)
    WL &S = *reinterpret_cast<WL *>(smem + TABLE_BYTES + (size_t)(w % HX_SHARE) * sizeof(WL));
#else
    WL &S = *reinterpret_cast<WL *>(smem + TABLE_BYTES + (size_t)w * sizeof(WL));
#endif
    uint8_t *D = S.buf + HS_OFF;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + (blockIdx.x * TW + w) * 0xBF58476D1CE4E5B9ull + lane0;
    for (int i = lane0; i < TILE_BASES + 128; i += 64) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        D[i] = "ACGT"[x & 3];
    }
    __syncthreads();
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        uint32_t caps[NPC], raw[5];
#pragma unroll
        for (int g = 0; g < NPC; g++) caps[g] = 0;
#pragma unroll
        for (int g = 0; g < 5; g++) raw[g] = 0;
        hash_loop_static<L, LA, false>(D, bound, lane, np, caps, raw);
        wave_sync();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NPC; g++) acc ^= caps[g];
#pragma unroll
        for (int g = 0; g < 5; g++) acc += raw[g];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 * TW + threadIdx.x] = acc;
    if (lane0 == 0) {
        cyc[blockIdx.x * TW + w] = t1 - t0;
        cyc[gridDim.x * TW + blockIdx.x * TW + w] = r1 - r0; // 100 MHz
    }
}

template <int L, int LA, bool HPC>
void run(const char *name, int n_cu, uint32_t *d_out, uint64_t *d_cyc, int np) {
    auto k = hl_kernel<L, LA, HPC>;
#ifdef HX_SHARE
    const int lds = TABLE_BYTES + HX_SHARE * (int)sizeof(WaveLdsT<HPC>);
#else
    const int lds = block_lds_bytes<HPC>();
#endif
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int iters = 3000;
    hipLaunchKernelGGL(k, dim3(n_cu * HX_BPC), dim3(64 * TW), lds, 0, d_out, d_cyc, 1000, 42949672u, np, 1u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(n_cu * HX_BPC), dim3(64 * TW), lds, 0, d_out, d_cyc, iters, 42949672u, np, 2u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h((size_t)n_cu * HX_BPC * TW * 2);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0, ravg = 0;
    for (size_t i = 0; i < h.size() / 2; i++) { avg += (double)h[i]; ravg += (double)h[h.size() / 2 + i]; }
    avg /= h.size() / 2; ravg /= h.size() / 2;
    const double per_tile_wave = avg / iters;                   // cycles one wave spends on one tile's hash loop
    const double per_tile_simd = per_tile_wave / (TW * HX_BPC / 4.0);    // TW * HX_BPC / 4 waves share a SIMD
    const int pos = 16 * np;
    printf("%-28s np=%d  %8.0f cyc/tile/wave  %7.0f cyc/tile/SIMD  %6.2f cyc/position/SIMD  %7.3f us/tile/SIMD  (%.3f ms wall, shader clock %.0f MHz)\n", name, np, per_tile_wave,
           per_tile_simd, per_tile_simd / pos, ravg / 100.0 / iters / (TW * HX_BPC / 4.0), ms, avg / ravg * 100.0);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    uint32_t *d_out;
    uint64_t *d_cyc;
    hipMalloc(&d_out, (size_t)n_cu * HX_BPC * 64 * TW * 4);
    hipMalloc(&d_cyc, (size_t)n_cu * HX_BPC * TW * 16);
    printf("# hash loop alone, %d x %d waves per CU, %d CUs", HX_BPC, TW, n_cu);
#ifdef HX_TAG
    printf(", variant %s", HX_TAG);
#endif
    printf("\n");
    run<31, 1, false>("L31 LA1 (Regular)", n_cu, d_out, d_cyc, 9);
    run<31, 2, false>("L31 LA2 (Hpc, full)", n_cu, d_out, d_cyc, 9);
    run<31, 2, false>("L31 LA2 (Hpc, np=7)", n_cu, d_out, d_cyc, 7);
    run<31, 4, false>("L31 LA4", n_cu, d_out, d_cyc, 9);
    return 0;
}
