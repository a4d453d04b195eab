// lds_rate.hip -- LDS store cost on gfx950: aligned / unaligned ds_write_b32, ds_write_b8, ds_write_b16, ds_read_b32 table reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint64_t *cyc, int iters, uint32_t s) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[4 * 3072];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    uint32_t base = (uint32_t)(uintptr_t)(lds_u8 *)buf + w * 3072;
    // per-lane region of ~36 bytes (like the compaction: consecutive lanes write consecutive ~27-byte runs)
    uint32_t a = base + lane * 27 + (MODE == 0 ? (4 - ((lane * 27) & 3)) & 3 : 0); // MODE 0: force 4-byte alignment
    uint32_t v = s * lane;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0 || MODE == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(j * 3));
            if (MODE == 0) {}
            if (MODE == 2) asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(j * 3));
            if (MODE == 3) asm volatile("ds_write_b16 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(j * 2));
            if (MODE == 4) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"((uint64_t)v), "n"(j * 3));
        }
        v += i;
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x] + v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef void (*kern_t)(uint32_t *, uint64_t *, int, uint32_t);
int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    uint32_t *d_out; uint64_t *d_cyc;
    (void)hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4);
    (void)hipMalloc(&d_cyc, (size_t)n_cu * 8 * 8);
    struct { const char *n; kern_t f; } tests[] = {{"ds_write_b32 4B-aligned (lanes 27-28 B apart)", k<0>}, {"ds_write_b32 unaligned (27 B lane stride)", k<1>},
        {"ds_write_b8 (27 B lane stride)", k<2>}, {"ds_write_b16 (27 B stride, odd addresses too)", k<3>}, {"ds_write_b64 unaligned", k<4>}};
    const int iters = 20000;
    printf("wall-clock cycles at 2.4 GHz per wave-instruction per CU\n");
    for (auto &t : tests) {
        printf("%-50s", t.n);
        for (int wps : {1, 2, 3}) {
            const int blocks = n_cu * wps;
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 10, 1u);
            (void)hipDeviceSynchronize();
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1u);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  %dw/SIMD: %6.2f", wps, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps * 4));
        }
        printf("\n");
    }
    return 0;
}
