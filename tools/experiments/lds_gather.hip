// lds_gather.hip -- cost of the hash loop's table look-ups on gfx950: ds_read_b64 / b32 gathers where the 64 lanes of a wave
// read one of 4 (or 16, or 64) distinct entries, against broadcast and linear reads.  Cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[16384];
    const int lane = threadIdx.x & 63;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    const uint32_t base = (uint32_t)(uintptr_t)(lds_u8 *)buf;
    for (int i = threadIdx.x; i < 4096; i += 256) ((uint32_t *)buf)[i] = i * seed;
    __syncthreads();
    uint32_t a[8];
    uint32_t x = (threadIdx.x + blockIdx.x * 256) * 2654435761u + seed;
    const uint8_t acgt[4] = {0x41, 0x43, 0x47, 0x54};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = x >> 16;
        if (MODE == 0 || MODE == 1) a[j] = base + 8 * acgt[r & 3];             // the kernel's table: 4 live entries of 256
        if (MODE == 2) a[j] = base;                                              // broadcast
        if (MODE == 3 || MODE == 4) a[j] = base + lane * 8 + j * 512;            // linear
        if (MODE == 5) a[j] = base + 8 * (r & 15);                               // 16 consecutive entries (pair table)
        if (MODE == 6) a[j] = base + 8 * (r & 3);                                // 4 consecutive entries
        if (MODE == 7) a[j] = base + 4 * acgt[r & 3];                            // b32 of the kernel's table
        if (MODE == 8) a[j] = base + 8 * (r & 255);                              // 256 random entries
        if (MODE == 9) a[j] = base + lane * 144 + j * 16;                        // the piece reads (b128, stride 144)
        if (MODE == 10) a[j] = base + 8 * (r & 3) + 32 * (lane & 15);            // 4 entries, replicated 16x (one copy per lane%16)
        if (MODE == 11) a[j] = base + 8 * (r & 3) + 32 * (lane & 3);             // replicated 4x
    }
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6 || MODE == 8 || MODE == 10 || MODE == 11) asm volatile("ds_read_b64 %0, %1" : "=v"(*(uint64_t *)&v0) : "v"(a[j]));
            if (MODE == 1 || MODE == 4 || MODE == 7) asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(a[j]));
            if (MODE == 9) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(a[j])); v0 = t.x; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3;
}
typedef void (*kern_t)(uint32_t *, int, uint32_t);
int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    uint32_t *d_out;
    (void)hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4);
    struct { const char *n; kern_t f; } tests[] = {
        {"ds_read_b64 gather, 4 live entries of the 256-entry table (A C G T)", k<0>},
        {"ds_read_b32 gather, same addresses", k<1>},
        {"ds_read_b64 broadcast (one address)", k<2>},
        {"ds_read_b64 linear (lane*8)", k<3>},
        {"ds_read_b32 linear (lane*8)", k<4>},
        {"ds_read_b64 gather, 16 consecutive entries", k<5>},
        {"ds_read_b64 gather, 4 consecutive entries", k<6>},
        {"ds_read_b32 gather, 4-byte entries of A C G T", k<7>},
        {"ds_read_b64 gather, 256 random entries", k<8>},
        {"ds_read_b128 pieces, lane stride 144 B", k<9>},
        {"ds_read_b64 gather, 4 entries x 16 replicas (lane%16)", k<10>},
        {"ds_read_b64 gather, 4 entries x 4 replicas (lane%4)", k<11>}};
    const int iters = 20000;
    printf("wall-clock cycles at 2.4 GHz per wave-instruction per CU\n");
    for (auto &t : tests) {
        printf("%-70s", t.n);
        for (int wps : {1, 2, 3}) {
            const int blocks = n_cu * wps;
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d_out, 10, 1u);
            (void)hipDeviceSynchronize();
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  %dw/SIMD: %6.2f", wps, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps * 4));
        }
        printf("\n");
    }
    return 0;
}
