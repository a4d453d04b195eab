// global_load_lds_dwordx4 issued from inline asm (M0 saved and restored inside the statement; the instruction offset moves BOTH the
// global and the LDS address): does lane i land at M0 + offset + 16 i?  Build: hipcc -O3 --offload-arch=gfx950 -o t lds_dma_asm_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* g, unsigned char* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned char* gg = g + w * 10240 + 16 * lane;
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds + 64 + w * 10240;
#pragma unroll
    for (int grp = 0; grp < 3; grp++) {
        const unsigned char* p = gg + 4096 * grp;
        const unsigned m = __builtin_amdgcn_readfirstlane(lds_base + 4096 * grp);
        unsigned save;
        if (grp < 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                         "s_mov_b32 m0, %0" : "=&s"(save) : "v"(p), "s"(m) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "s_mov_b32 m0, %0" : "=&s"(save) : "v"(p), "s"(m) : "memory");
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 10240; i += 64) out[w * 10240 + i] = lds[64 + w * 10240 + i];
}
int main() {
    const int W = 4, N = W * 10240 + 4096;
    std::vector<unsigned char> h(N), o(W * 10240, 0xEE);
    for (int i = 0; i < N; i++) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    unsigned char *dg, *dout;
    (void)hipMalloc(&dg, N); (void)hipMalloc(&dout, W * 10240);
    (void)hipMemcpy(dg, h.data(), N, hipMemcpyHostToDevice);
    (void)hipMemset(dout, 0xEE, W * 10240);
    k<<<1, 64 * W, 64 + W * 10240 + 64>>>(dg, dout);
    hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(o.data(), dout, W * 10240, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < W * 10240; i++)
        if (o[i] != h[i]) { if (bad < 5) printf("mismatch i%d got %02x want %02x\n", i, o[i], h[i]); bad++; }
    printf("status %s bad %d\n", hipGetErrorString(e), bad);
    return bad != 0;
}
