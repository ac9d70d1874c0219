// membench6.hip -- how fast can the CUs of ONE XCD read HBM, and what do the bytes in flight per CU buy?  (measurement tool, round 3:
// the column step of team_quad_kernel keeps one 64 KiB chunk in flight per CU and lands a chunk in 3 - 4 us = 16 - 20 GB/s per CU.)
// One 512-thread workgroup per CU; workgroups on XCDs outside `mask` leave at once.  Every live workgroup streams its own region by
// LDS-DMA (nt), DEPTH tiles of 64 KiB in flight (the tile is dropped: nothing is computed).  Prints GB/s for 1, 2, 4, 8 XCDs and DEPTH 1, 2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma_nt(const u32x4* p, unsigned lds) {
    const unsigned a = __builtin_amdgcn_readfirstlane(lds);
    unsigned saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
}

template <int DEPTH>
__global__ __launch_bounds__(512) void k_read(const u32x4* in, unsigned* sink, long long region16, int tiles, unsigned mask) {
    extern __shared__ u32x4 land[];  // DEPTH x 4096 x 16 bytes
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 15u;
    if (!((mask >> x) & 1u)) return;
    const unsigned tid = threadIdx.x;
    const u32x4* src = in + (long long)blockIdx.x * region16;
    const unsigned lds0 = (unsigned)(size_t)land;
    auto dma = [&](int t) {
        const u32x4* s = src + (long long)t * 4096 + tid;
        const unsigned base = lds0 + (unsigned)(t % DEPTH) * 65536u;
#pragma unroll
        for (int i = 0; i < 8; i++) dma_nt(s + i * 512, base + (unsigned)(i * 512 + (tid & ~63u)) * 16u);
    };
    for (int t = 0; t < DEPTH && t < tiles; t++) dma(t);
    for (int t = 0; t < tiles; t++) {
        // tile t has landed: at most (DEPTH - 1) x 8 younger requests may still fly
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        if (t + DEPTH < tiles) dma(t + DEPTH);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && land[5].x == 0x12345678u) sink[0] = 1;
}

int main() {
    const int nwg = 256, tiles = 256;                  // 256 x 64 KiB = 16 MiB per workgroup, 4 GiB in all
    const long long region16 = (long long)tiles * 4096;
    u32x4* in; unsigned* sink;
    CK(hipMalloc(&in, (size_t)nwg * region16 * 16));
    CK(hipMemset(in, 1, (size_t)nwg * region16 * 16));
    CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute((const void*)k_read<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_read<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned masks[4] = {0x01u, 0x03u, 0x0fu, 0xffu};
    for (int depth = 1; depth <= 2; depth++)
        for (int m = 0; m < 4; m++) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                if (depth == 1) hipLaunchKernelGGL(k_read<1>, dim3(nwg), dim3(512), 65536, 0, in, sink, region16, tiles, masks[m]);
                else hipLaunchKernelGGL(k_read<2>, dim3(nwg), dim3(512), 131072, 0, in, sink, region16, tiles, masks[m]);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const int nx = __builtin_popcount(masks[m]);
            const double bytes = (double)nx * 32 * tiles * 65536.0;
            printf("%d XCD(s) reading, %3d KiB in flight per CU: %7.1f GB/s = %6.1f GB/s per XCD = %5.1f GB/s per CU (%.3f ms)\n", nx, 64 * depth,
                   bytes / best / 1e6, bytes / best / 1e6 / nx, bytes / best / 1e6 / nx / 32, best);
        }
    return 0;
}
