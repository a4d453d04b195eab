// Experiment kept for reference: semantics of global_load_lds_dwordx4 on gfx950 -- lane i of a load lands at
// LDS base + instruction offset + 16 i, also under a partial EXEC mask.  Build: hipcc -O3 --offload-arch=gfx950 -o t lds_direct_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* g, unsigned char* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char* D = lds + 4096 + w * 3072 + 16;
    const unsigned char* gg = g + w * 2048;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gg + 16 * lane),
                                     (__attribute__((address_space(3))) void*)D, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gg + 1024 + 16 * lane),
                                     (__attribute__((address_space(3))) void*)(D + 1024), 16, 0, 0);
    if (lane < 8)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gg + 2048 + 16 * lane),
                                         (__attribute__((address_space(3))) void*)(D + 2048), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 2048 + 128; i += 64) out[w * 4096 + i] = D[i];
}
int main() {
    const int W = 4, N = W * 2048 + 4096;
    std::vector<unsigned char> h(N), o(W * 4096, 0xEE);
    for (int i = 0; i < N; i++) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    unsigned char *dg, *dout;
    hipMalloc(&dg, N); hipMalloc(&dout, W * 4096);
    hipMemcpy(dg, h.data(), N, hipMemcpyHostToDevice);
    hipMemset(dout, 0xEE, W * 4096);
    k<<<1, 64 * W, 4096 + W * 3072 + 64>>>(dg, dout);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o.data(), dout, W * 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < W; w++)
        for (int i = 0; i < 2048 + 128; i++)
            if (o[w * 4096 + i] != h[w * 2048 + i]) { if (bad < 5) printf("mismatch w%d i%d got %02x want %02x\n", w, i, o[w*4096+i], h[w*2048+i]); bad++; }
    printf("status %s bad %d\n", hipGetErrorString(e), bad);
    return bad != 0;
}
