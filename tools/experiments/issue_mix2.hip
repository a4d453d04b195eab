// issue_mix2.hip -- cost of mixed streams of "dear" (v_alignbit_b32: shifts, rotates, min/max, compares, selects, SDWA ...) and "cheap"
// (v_xor_b32: and/or/xor/add/sub/mov, v_bitop3_b32) vector instructions, all independent (16 registers round-robin), at 2/3/4/8 waves per SIMD.
// Question: does replacing one dear instruction by two cheap ones pay inside a stream of dear instructions?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>

#define D(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 31\n\t"
#define C(i) "v_xor_b32 %" #i ", %" #i ", %16\n\t"
#define B(i) "v_bitop3_b32 %" #i ", %" #i ", %16, %17 bitop3:0x96\n\t"
#define M(i) "v_min_u32 %" #i ", %" #i ", %16\n\t"

#define KERNEL(NAME, ASMTEXT)                                                                                      \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint64_t *cyc, int iters, uint32_t s) {         \
        uint32_t a[16];                                                                                           \
        for (int i = 0; i < 16; i++) a[i] = threadIdx.x * (2 * i + 3) + s;                                        \
        uint32_t b = s ^ threadIdx.x, c = s + 7;                                                                  \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                               \
        for (int i = 0; i < iters; i++) {                                                                         \
            asm volatile(ASMTEXT ASMTEXT                                                                          \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                           "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                         : "v"(b), "v"(c), "s"(s)                                                                 \
                         : "memory");                                                                             \
        }                                                                                                         \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                               \
        uint32_t x = 0;                                                                                           \
        for (int i = 0; i < 16; i++) x ^= a[i];                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = x;                                                                  \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }

// 16 instructions per text (x2 per loop body)
KERNEL(d16, D(0) D(1) D(2) D(3) D(4) D(5) D(6) D(7) D(8) D(9) D(10) D(11) D(12) D(13) D(14) D(15))
KERNEL(c16, C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15))
KERNEL(b16, B(0) B(1) B(2) B(3) B(4) B(5) B(6) B(7) B(8) B(9) B(10) B(11) B(12) B(13) B(14) B(15))
KERNEL(dc, D(0) C(1) D(2) C(3) D(4) C(5) D(6) C(7) D(8) C(9) D(10) C(11) D(12) C(13) D(14) C(15))
KERNEL(dcc, D(0) C(1) C(2) D(3) C(4) C(5) D(6) C(7) C(8) D(9) C(10) C(11) D(12) C(13) C(14) D(15))       /* 6 D + 10 C */
KERNEL(dccc, D(0) C(1) C(2) C(3) D(4) C(5) C(6) C(7) D(8) C(9) C(10) C(11) D(12) C(13) C(14) C(15))       /* 4 D + 12 C */
KERNEL(ddc, D(0) D(1) C(2) D(3) D(4) C(5) D(6) D(7) C(8) D(9) D(10) C(11) D(12) D(13) C(14) D(15))       /* 11 D + 5 C */
KERNEL(dddc, D(0) D(1) D(2) C(3) D(4) D(5) D(6) C(7) D(8) D(9) D(10) C(11) D(12) D(13) D(14) C(15))       /* 12 D + 4 C */
KERNEL(d8c8, D(0) D(1) D(2) D(3) D(4) D(5) D(6) D(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15))       /* blocks */
KERNEL(db, D(0) B(1) D(2) B(3) D(4) B(5) D(6) B(7) D(8) B(9) D(10) B(11) D(12) B(13) D(14) B(15))
KERNEL(dm, D(0) M(1) D(2) M(3) D(4) M(5) D(6) M(7) D(8) M(9) D(10) M(11) D(12) M(13) D(14) M(15))
// dependent pairs: the cheap instruction consumes the dear one's result (as in rotate -> xor3)
#define DCdep(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 31\n\tv_xor_b32 %" #i ", %" #i ", %16\n\t"
KERNEL(dcdep, DCdep(0) DCdep(1) DCdep(2) DCdep(3) DCdep(4) DCdep(5) DCdep(6) DCdep(7))

typedef void (*kern_t)(uint32_t *, uint64_t *, int, uint32_t);
struct Test { const char *name; kern_t k; int nd, nc; };

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    uint32_t *d_out;
    uint64_t *d_cyc;
    const int max_blocks = n_cu * 8;
    hipMalloc(&d_out, (size_t)max_blocks * 256 * 4);
    hipMalloc(&d_cyc, (size_t)max_blocks * 8);
    std::vector<uint64_t> h(max_blocks);
    Test tests[] = {{"16 dear", k_d16, 16, 0}, {"16 cheap (xor)", k_c16, 0, 16}, {"16 cheap (bitop3)", k_b16, 0, 16}, {"D C alternating", k_dc, 8, 8},
                    {"D C C", k_dcc, 6, 10}, {"D C C C", k_dccc, 4, 12}, {"D D C", k_ddc, 11, 5}, {"D D D C", k_dddc, 12, 4},
                    {"8 D then 8 C", k_d8c8, 8, 8}, {"D bitop3 alternating", k_db, 8, 8}, {"D v_min alternating", k_dm, 16, 0},
                    {"D->C dependent pairs", k_dcdep, 8, 8}};
    const int iters = 20000;
    printf("cycles per 16 instructions per SIMD (nd dear + nc cheap)\n%-26s", "");
    for (int wps : {2, 3, 4, 8}) printf("  %dw/SIMD", wps);
    printf("\n");
    for (auto &t : tests) {
        printf("%-22s %2d+%2d", t.name, t.nd, t.nc);
        for (int wps : {2, 3, 4, 8}) {
            const int blocks = n_cu * wps;
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 2000, 1u);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1u);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d_cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < blocks; i++) avg += (double)h[i];
            avg /= blocks;
            printf("  %7.2f", avg / ((double)iters * 2) / wps);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
