// cnd_rate.hip -- when is v_cndmask_b32 slow?  Alone (VCC never written in the loop) it costs 13 cycles per wave-instruction per SIMD at three
// waves per SIMD instead of 3 (valu_rate.hip); right behind the compare that wrote VCC it costs 3.  Variants: how far behind the compare, how many
// readers of one VCC, a mask in an SGPR pair instead of VCC, other VCC readers (v_addc).  cycles per wave-instruction per SIMD at 3 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/experiments/cnd_rate.hip -o tools/experiments/cnd_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define C(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define CS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define CMPV "v_cmp_ge_u32 vcc, %10, %0\n\t"
#define CMPS "v_cmp_ge_u32_e64 s[20:21], %10, %0\n\t"
#define MOV(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n\t"

#define KERNEL(NAME, BODYASM)                                                                                     \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint64_t *cyc, int iters, uint32_t s) {          \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t b = s ^ threadIdx.x, c = s + 7;                                                                  \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                               \
        for (int i = 0; i < iters; i++) {                                                                         \
            asm volatile(BODYASM BODYASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c), "s"(s) : "vcc", "s20", "s21");                                         \
        }                                                                                                         \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                               \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }

KERNEL(cmp_cnd, CMPV C(0) CMPV C(1) CMPV C(2) CMPV C(3))                                   // 8 instructions
KERNEL(cmp_x_cnd, CMPV MOV(4) C(0) CMPV MOV(5) C(1) CMPV MOV(6) C(2) CMPV MOV(7) C(3))       // 12: one instruction between compare and select
KERNEL(cmp_xx_cnd, CMPV MOV(4) MOV(5) C(0) CMPV MOV(6) MOV(7) C(1) CMPV MOV(4) MOV(5) C(2) CMPV MOV(6) MOV(7) C(3)) // 16: two between
KERNEL(cmp_cnd_cnd, CMPV C(0) C(1) CMPV C(2) C(3) CMPV C(4) C(5) CMPV C(6) C(7))              // 12: two selects per compare
KERNEL(cmp_cnd_x_cnd, CMPV C(0) MOV(4) C(1) CMPV C(2) MOV(5) C(3) CMPV C(0) MOV(6) C(1) CMPV C(2) MOV(7) C(3)) // 16
KERNEL(cmp_cnd_addc, CMPV C(0) ADDC(4) CMPV C(1) ADDC(5) CMPV C(2) ADDC(6) CMPV C(3) ADDC(7)) // 12: the hash loop's chain
KERNEL(cmps_cnds_cnds, CMPS CS(0) CS(1) CMPS CS(2) CS(3) CMPS CS(4) CS(5) CMPS CS(6) CS(7))   // 12: mask in an SGPR pair, two selects
KERNEL(cmps_x_cnds, CMPS MOV(4) CS(0) CMPS MOV(5) CS(1) CMPS MOV(6) CS(2) CMPS MOV(7) CS(3))  // 12
KERNEL(cnds_only, CS(0) CS(1) CS(2) CS(3) CS(4) CS(5) CS(6) CS(7))                           // 8: SGPR mask never written
KERNEL(cnd_only, C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7))                                    // 8: VCC never written

typedef void (*kern_t)(uint32_t *, uint64_t *, int, uint32_t);
struct Test { const char *name; kern_t k; int n; };

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount, wps = 3, blocks = n_cu * wps, iters = 3000;
    uint32_t *d_out;
    uint64_t *d_cyc;
    (void)hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&d_cyc, (size_t)blocks * 8);
    std::vector<uint64_t> h(blocks);
    Test tests[] = {{"cmp, cnd", k_cmp_cnd, 8}, {"cmp, x, cnd", k_cmp_x_cnd, 12}, {"cmp, x, x, cnd", k_cmp_xx_cnd, 16}, {"cmp, cnd, cnd", k_cmp_cnd_cnd, 12},
                    {"cmp, cnd, x, cnd", k_cmp_cnd_x_cnd, 16}, {"cmp, cnd, addc", k_cmp_cnd_addc, 12}, {"cmp->sgpr, cnd(s), cnd(s)", k_cmps_cnds_cnds, 12},
                    {"cmp->sgpr, x, cnd(s)", k_cmps_x_cnds, 12}, {"cnd(s) only", k_cnds_only, 8}, {"cnd(vcc) only", k_cnd_only, 8}};
    printf("%-28s cycles per group per SIMD / per instruction (three waves per SIMD)\n", "");
    for (auto &t : tests) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 200, 1u);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1u);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d_cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < blocks; i++) avg += (double)h[i];
        avg /= blocks;
        const double per_group = avg / ((double)iters * 2) / wps;
        printf("%-28s %8.2f / %5.2f   (%d instructions)\n", t.name, per_group, per_group / t.n, t.n);
        fflush(stdout);
    }
    return 0;
}
