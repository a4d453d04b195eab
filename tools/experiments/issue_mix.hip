// issue_mix.hip -- what does an instruction that is NOT a vector-ALU instruction cost a SIMD that is busy with vector-ALU work?
// Body = 8 independent v_alignbit_b32 (the "dear" class of valu_rate.hip) + 8 instructions of the class under test, interleaved;
// printed: shader cycles per body per SIMD at 2 / 3 / 4 waves per SIMD, and the difference to the bare body = what 8 of them cost.
// Build: hipcc -O3 --offload-arch=gfx950 tools/experiments/issue_mix.hip -o tools/experiments/issue_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define V(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 31\n\t"
#define BODY(X) V(0) X V(1) X V(2) X V(3) X V(4) X V(5) X V(6) X V(7) X

#define KERNEL(NAME, X, ...)                                                                                      \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint64_t *cyc, int iters, uint32_t s) {         \
        __shared__ uint32_t lds[1024];                                                                            \
        lds[threadIdx.x] = s;                                                                                     \
        __syncthreads();                                                                                          \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, \
                 a7 = a0 * 19;                                                                                    \
        uint32_t b = (threadIdx.x & 3) * 8, c = s + 7;                                                            \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                               \
        for (int i = 0; i < iters; i++) {                                                                         \
            asm volatile(BODY(X) BODY(X) BODY(X) BODY(X)                                                          \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)         \
                         : "v"(b), "v"(c), "s"(s)                                                                 \
                         : __VA_ARGS__);                                                                          \
        }                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ lds[(threadIdx.x + 1) & 255]; \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }

KERNEL(bare, "", "memory")
KERNEL(salu, "s_add_u32 s20, s20, 1\n\t", "memory", "s20", "scc")
KERNEL(snop, "s_nop 0\n\t", "memory")
KERNEL(swait, "s_waitcnt lgkmcnt(0)\n\t", "memory")
KERNEL(smov, "s_mov_b32 s20, 5\n\t", "memory", "s20")
KERNEL(sand64, "s_and_b64 s[20:21], s[20:21], exec\n\t", "memory", "s20", "s21", "scc")
KERNEL(vmovc, "v_mov_b32 v40, %8\n\t", "memory", "v40")
KERNEL(vxor, "v_xor_b32 v40, v40, %8\n\t", "memory", "v40")
KERNEL(vdear, "v_alignbit_b32 v40, v40, v40, 3\n\t", "memory", "v40")
KERNEL(readlane, "v_readlane_b32 s20, %9, 3\n\t", "memory", "s20")
KERNEL(readfirst, "v_readfirstlane_b32 s20, %9\n\t", "memory", "s20")
KERNEL(writelane, "v_writelane_b32 v40, s20, 3\n\t", "memory", "v40")
KERNEL(dsread64, "ds_read_b64 v[40:41], %8\n\t", "memory", "v40", "v41")
KERNEL(dsread64w, "ds_read_b64 v[40:41], %8\n\ts_waitcnt lgkmcnt(2)\n\t", "memory", "v40", "v41")
KERNEL(dsread128, "ds_read_b128 v[40:43], %8\n\t", "memory", "v40", "v41", "v42", "v43")
KERNEL(dswrite8, "ds_write_b8 %8, %9\n\t", "memory")
KERNEL(dswrite32, "ds_write_b32 %8, %9\n\t", "memory")
KERNEL(cmpx, "v_cmp_ge_u32 vcc, %10, %9\n\t", "memory", "vcc")
KERNEL(saveexec, "s_and_saveexec_b64 s[20:21], vcc\n\ts_mov_b64 exec, s[20:21]\n\t", "memory", "s20", "s21", "scc")
KERNEL(branch_nt, "s_cbranch_scc1 1f\n\t1:\n\t", "memory")
KERNEL(dpp, "v_mov_b32_dpp v40, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n\t", "memory", "v40")
KERNEL(bperm, "ds_bpermute_b32 v40, %8, %9\n\t", "memory", "v40")

typedef void (*kern_t)(uint32_t *, uint64_t *, int, uint32_t);
struct Test { const char *name; kern_t k; };

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
    Test tests[] = {{"bare (8 v_alignbit)", k_bare}, {"+8 s_add_u32", k_salu}, {"+8 s_nop 0", k_snop}, {"+8 s_waitcnt (idle)", k_swait},
                    {"+8 s_mov_b32", k_smov}, {"+8 s_and_b64", k_sand64}, {"+8 v_mov (cheap)", k_vmovc}, {"+8 v_xor (cheap)", k_vxor},
                    {"+8 v_alignbit (dear)", k_vdear}, {"+8 v_readlane", k_readlane}, {"+8 v_readfirstlane", k_readfirst},
                    {"+8 v_writelane", k_writelane}, {"+8 ds_read_b64", k_dsread64}, {"+8 ds_read_b64+wait(2)", k_dsread64w},
                    {"+8 ds_read_b128", k_dsread128}, {"+8 ds_write_b8", k_dswrite8}, {"+8 ds_write_b32", k_dswrite32},
                    {"+8 v_cmp", k_cmpx}, {"+8 saveexec+restore", k_saveexec}, {"+8 s_cbranch (not taken)", k_branch_nt},
                    {"+8 v_mov_dpp", k_dpp}, {"+8 ds_bpermute", k_bperm}};
    const int iters = 20000;
    printf("cycles per body (8 v_alignbit + 8 x) per SIMD; in brackets: cycles per added instruction per SIMD\n%-26s", "");
    for (int wps : {2, 3, 4, 8}) printf("        %dw/SIMD", wps);
    printf("\n");
    double bare[4] = {0, 0, 0, 0};
    for (auto &t : tests) {
        printf("%-26s", t.name);
        int wi = 0;
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
            const double per_body = avg / ((double)iters * 4) / wps;
            if (t.k == k_bare) bare[wi] = per_body;
            printf("  %6.2f [%5.2f]", per_body, (per_body - bare[wi]) / 8.0);
            wi++;
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
