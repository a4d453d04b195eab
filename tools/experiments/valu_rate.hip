// valu_rate.hip -- issue rate of the integer VALU instructions the minimizer kernel is made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
// For every instruction: a loop of 8 independent chains x 32 instructions, run with 1/2/3/4/8 waves per SIMD on every CU;
// prints cycles per wave-instruction per SIMD (shader clock from s_memtime).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY4(I) I I I I

#define KERNEL(NAME, ASM, ...)                                                                          \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint64_t *cyc, int iters, uint32_t s) { \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17,   \
                 a7 = a0 * 19;                                                                          \
        uint32_t b = s ^ threadIdx.x, c = s + 7;                                                        \
        uint64_t t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();                                \
        for (int i = 0; i < iters; i++) {                                                               \
            BODY4(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                   \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                               : "v"(b), "v"(c), "s"(s)                                                 \
                               : __VA_ARGS__);)                                                                \
        }                                                                                               \
        uint64_t t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                                \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                     \
        if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[gridDim.x + blockIdx.x] = w1 - w0; }                                                \
    }

#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define A_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 31\n\t"
#define A_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n\t"
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define A_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define A_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n\t"
#define A_CMP(i) "v_cmp_ge_u32 vcc, %10, %" #i "\n\t"
#define A_SDWA(i) "v_lshlrev_b32_sdwa %" #i ", %9, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n\t"
#define A_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n\t"
#define A_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n\t"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n\t"
#define A_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n\t"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n\t"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n\t"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n\t"
#define A_PKADD16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n\t"
#define A_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 3\n\t"
#define A_MIX(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 31\n\tv_xor_b32 %" #i ", %" #i ", %8\n\t"

KERNEL(xor, A_XOR, "memory")
KERNEL(alignbit, A_ALIGN, "memory")
KERNEL(min, A_MIN, "memory")
KERNEL(add, A_ADD, "memory")
KERNEL(cndmask, A_CND, "memory")
KERNEL(addc, A_ADDC, "vcc")
KERNEL(cmp, A_CMP, "vcc")
KERNEL(sdwa, A_SDWA, "memory")
KERNEL(fma, A_FMA, "memory")
KERNEL(andor, A_ANDOR, "memory")
KERNEL(perm, A_PERM, "memory")
KERNEL(dot4, A_DOT4, "memory")
KERNEL(bcnt, A_BCNT, "memory")
KERNEL(lshlor, A_LSHLOR, "memory")
KERNEL(mullo, A_MULLO, "memory")
KERNEL(mul24, A_MUL24, "memory")
KERNEL(min3, A_MIN3, "memory")
KERNEL(bfe, A_BFE, "memory")
KERNEL(xad, A_XAD, "memory")
KERNEL(add3, A_ADD3, "memory")
KERNEL(or3, A_OR3, "memory")
KERNEL(mov, A_MOV, "memory")
KERNEL(pkadd16, A_PKADD16, "memory")
KERNEL(alignbyte, A_ALIGNBYTE, "memory")
KERNEL(rotxor, A_MIX, "memory")
#define A_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
KERNEL(cnd64, A_CND64, "s20", "s21")
#define A_TRACK(i) "v_cmp_ge_u32_e32 vcc, %10, %8\n\tv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n\tv_addc_co_u32_e32 %9, vcc, %9, %9, vcc\n\t"
#define A_TRACK2(i) "v_cmp_ge_u32_e32 vcc, %10, %8\n\tv_addc_co_u32_e32 %" #i ", vcc, %" #i ", %" #i ", vcc\n\t"
#define A_CMPS(i) "v_cmp_ge_u32_e64 s[20:21], %10, %" #i "\n\t"
#define A_CNDSDWA(i) "v_cndmask_b32_sdwa %" #i ", %" #i ", %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define A_MINMIN(i) "v_min_u32 %" #i ", %" #i ", %8\n\tv_xor_b32 %" #i ", %" #i ", %9\n\t"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n\t"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define A_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n\t"
#define A_MAX(i) "v_max_u32 %" #i ", %" #i ", %8\n\t"
#define A_LSHL64(i) "v_lshlrev_b64 v[40:41], 3, v[40:41]\n\t"
#define A_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n\t"
#define A_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n\t"
#define A_XORSDWA(i) "v_xor_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define A_PKXOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n\tv_xor_b32 %" #i ", %" #i ", %9\n\t"
KERNEL(track, A_TRACK, "vcc")
KERNEL(track2, A_TRACK2, "vcc")
KERNEL(cmps, A_CMPS, "s20", "s21")
KERNEL(cndsdwa, A_CNDSDWA, "memory")
KERNEL(minxor, A_MINMIN, "memory")
KERNEL(lshl, A_LSHL, "memory")
KERNEL(and_, A_AND, "memory")
KERNEL(sub, A_SUB, "memory")
#define A_SUBCO(i) "v_sub_co_u32 %8, vcc, %10, %" #i "\n\t"
KERNEL(subco, A_SUBCO, "vcc")
#define A_TRACK3(i) "v_sub_co_u32 %8, vcc, %10, %" #i "\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\tv_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n\t"
KERNEL(track3, A_TRACK3, "vcc")
KERNEL(max, A_MAX, "memory")
KERNEL(lshl64, A_LSHL64, "v40", "v41")
KERNEL(madu24, A_MADU24, "memory")
KERNEL(bfi, A_BFI, "memory")
KERNEL(xorsdwa, A_XORSDWA, "memory")
KERNEL(xorxor, A_PKXOR, "memory")

// round 3: operands from three different registers, and the hash step's own mix (one strand: rotate, three-input XOR, min, compare,
// select, add-with-carry) -- does an instruction cost more inside the kernel than in the one-instruction loops above?
#define A_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n\t"
#define A_ALIGN2(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n\t"
#define A_HSTEP(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 31\n\tv_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n\tv_min_u32 v40, %" #i ", %8\n\tv_cmp_ge_u32 vcc, %10, v40\n\tv_cndmask_b32 v41, v41, v40, vcc\n\tv_addc_co_u32 v42, vcc, v42, v42, vcc\n\t"
KERNEL(bitop3, A_BITOP3, "memory")
KERNEL(align2, A_ALIGN2, "memory")
KERNEL(hstep, A_HSTEP, "vcc", "v40", "v41", "v42")

// round 4: the 64-bit and multiply instructions the compiler makes of index arithmetic
#define A_MAD64(i) "v_mad_u64_u32 v[40:41], s[20:21], %" #i ", %8, v[40:41]\n\t"
#define A_LSHLADD64(i) "v_lshl_add_u64 v[40:41], v[40:41], 2, v[42:43]\n\t"
#define A_CMP64(i) "v_cmp_lt_u64_e32 vcc, v[40:41], v[42:43]\n\t"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n\t"
#define A_MOV64(i) "v_mov_b64 v[40:41], v[42:43]\n\t"
#define A_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n\t"
#define A_BFM(i) "v_bfm_b32 %" #i ", %" #i ", %8\n\t"
KERNEL(mad64, A_MAD64, "v40", "v41", "s20", "s21")
KERNEL(lshladd64, A_LSHLADD64, "v40", "v41", "v42", "v43")
KERNEL(cmp64, A_CMP64, "vcc", "v40", "v41", "v42", "v43")
KERNEL(mulhi, A_MULHI, "memory")
KERNEL(mov64, A_MOV64, "v40", "v41", "v42", "v43")
KERNEL(med3, A_MED3, "memory")
KERNEL(bfm, A_BFM, "memory")

// LDS table look-ups as in the hash loop: ds_read_b64 of one of 4 entries per lane
__global__ __launch_bounds__(256) void k_ldsb64(uint32_t *out, uint64_t *cyc, int iters, uint32_t s) {
    __shared__ uint2 tab[512];
    for (int i = threadIdx.x; i < 512; i += 256) tab[i] = make_uint2(i * 2654435761u, i * 40503u);
    __syncthreads();
    uint32_t idx[8];
    for (int j = 0; j < 8; j++) idx[j] = ((threadIdx.x * (j + 3) + s) & 3) * 2 + 0x41;
    uint32_t acc = 0, acc2 = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint2 v = tab[idx[j] + r];
                acc ^= v.x;
                acc2 ^= v.y;
                idx[j] = (idx[j] & ~6u) | (v.x & 6u);
            }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ acc2;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(uint32_t *, uint64_t *, int, uint32_t);
struct Test { const char *name; kern_t k; int ops_per_body; };

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, n_cu, prop.clockRate);
    uint32_t *d_out;
    {   // s_memtime vs wall_clock64 (100 MHz)
        struct K { static __global__ void cal(uint64_t *o) { uint64_t a = __builtin_amdgcn_s_memtime(), w = wall_clock64(); uint32_t x = threadIdx.x; for (int i = 0; i < 2000000; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x)); uint64_t b = __builtin_amdgcn_s_memtime(), w2 = wall_clock64(); o[0] = b - a; o[1] = w2 - w; o[2] = x; } };
        uint64_t *d; hipMalloc(&d, 64); hipLaunchKernelGGL(K::cal, dim3(1), dim3(64), 0, 0, d); uint64_t hh[3]; hipMemcpy(hh, d, 24, hipMemcpyDeviceToHost);
        printf("s_memtime ticks %llu per wall_clock64 ticks %llu (100 MHz) -> s_memtime = %.1f MHz; 2M dependent v_add: %.2f ticks each\n", (unsigned long long)hh[0], (unsigned long long)hh[1], 100.0 * hh[0] / hh[1], hh[0] / 2e6);
    }
    uint64_t *d_cyc;
    const int max_blocks = n_cu * 8;
    hipMalloc(&d_out, (size_t)max_blocks * 256 * 4);
    hipMalloc(&d_cyc, (size_t)max_blocks * 16);
    std::vector<uint64_t> h(2 * max_blocks);
    Test tests[] = {{"v_xor_b32", k_xor, 32}, {"v_alignbit_b32", k_alignbit, 32}, {"v_min_u32", k_min, 32}, {"v_add_u32", k_add, 32},
                    {"v_cndmask_b32(vcc)", k_cndmask, 32}, {"v_addc_co_u32", k_addc, 32}, {"v_cmp_ge_u32", k_cmp, 32},
                    {"v_lshlrev_b32_sdwa", k_sdwa, 32}, {"v_fma_f32", k_fma, 32}, {"v_and_or_b32", k_andor, 32},
                    {"v_perm_b32", k_perm, 32}, {"v_dot4_u32_u8", k_dot4, 32}, {"v_bcnt_u32_b32", k_bcnt, 32},
                    {"v_lshl_or_b32", k_lshlor, 32}, {"v_mul_lo_u32", k_mullo, 32}, {"v_mul_u32_u24", k_mul24, 32},
                    {"v_min3_u32", k_min3, 32}, {"v_bfe_u32", k_bfe, 32}, {"v_xad_u32", k_xad, 32}, {"v_add3_u32", k_add3, 32},
                    {"v_or3_b32", k_or3, 32}, {"v_mov_b32", k_mov, 32}, {"v_pk_add_u16", k_pkadd16, 32},
                    {"v_alignbyte_b32", k_alignbyte, 32}, {"alignbit+xor pair", k_rotxor, 64}, {"ds_read_b64 (4 addr)", k_ldsb64, 32},
                    {"v_cndmask_e64 sgpr", k_cnd64, 32}, {"cmp+cnd+addc (3)", k_track, 96}, {"cmp+addc (2)", k_track2, 64},
                    {"v_cmp_e64 ->sgpr", k_cmps, 32}, {"v_cndmask_sdwa", k_cndsdwa, 32}, {"min+xor (2)", k_minxor, 64},
                    {"v_lshlrev_b32", k_lshl, 32}, {"v_and_b32", k_and_, 32}, {"v_sub_u32", k_sub, 32}, {"v_max_u32", k_max, 32},
                    {"v_lshlrev_b64", k_lshl64, 32}, {"v_mad_u32_u24", k_madu24, 32}, {"v_bfi_b32", k_bfi, 32},
                    {"v_xor_b32_sdwa", k_xorsdwa, 32}, {"xor+xor (2)", k_xorxor, 64}, {"v_sub_co_u32", k_subco, 32}, {"sub_co+cnd+addc (3)", k_track3, 96},
                    {"v_bitop3_b32 (3 regs)", k_bitop3, 32}, {"v_alignbit (2 regs)", k_align2, 32}, {"hash step mix (6)", k_hstep, 192},
                    {"v_mad_u64_u32", k_mad64, 32}, {"v_lshl_add_u64", k_lshladd64, 32}, {"v_cmp_lt_u64", k_cmp64, 32}, {"v_mul_hi_u32", k_mulhi, 32},
                    {"v_mov_b64", k_mov64, 32}, {"v_med3_i32", k_med3, 32}, {"v_bfm_b32", k_bfm, 32}};
    const int iters = 20000;
    printf("%-22s", "cycles/wave-instr/SIMD");
    for (int wps : {1, 2, 3, 4, 8}) printf("  %dw/SIMD", wps);
    printf("   (s_memtime cycles / (instrs x waves-per-SIMD))\n");
    for (auto &t : tests) {
        printf("%-22s", t.name);
        for (int wps : {1, 2, 3, 4, 8}) {
            const int blocks = n_cu * wps; // 256 threads = 4 waves = one per SIMD
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 10, 1u);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1u);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d_cyc, (size_t)blocks * 16, hipMemcpyDeviceToHost);
            double avg = 0, wavg = 0;
            for (int i = 0; i < blocks; i++) { avg += (double)h[i]; wavg += (double)h[blocks + i]; }
            avg /= blocks; wavg /= blocks;
            const double instrs = (double)iters * t.ops_per_body;
            printf("  %7.2f", avg / (instrs * wps));
            if (wps == 8) printf("   [%.3f ms wall; block alive %.3f ms (100 MHz clock); shader clock %.0f MHz]", ms, wavg / 1e5, avg / wavg * 100.0);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
