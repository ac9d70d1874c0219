// membench7.hip -- how fast can the CUs of ONE XCD WRITE to HBM, in a burst and sustained?  (measurement tool, round 4: the result stores of
// team_quad_kernel are the most expensive single item of a transform -- profiles/r4_price_list.txt --: 8 MiB per XCD leave in one burst at
// the end of every transform and the store instructions take 5.5 - 7.5 us to ISSUE.)
// One 512-thread workgroup per CU; workgroups on XCDs outside `mask` leave at once.  Every live workgroup writes `tiles` tiles of 64 KiB
// (8 nt stores of 16 bytes per thread and tile, registers only: nothing is read) in one of two shapes:
//   contiguous: its own 64 KiB per tile;
//   strided:    the result pattern of n = 2^20 -- per tile 256 rows of 256 bytes, 8 KiB apart, the XCD's 32 workgroups side by side in a row
// and reports the rate of the whole launch (sustained: 256 tiles = 16 MiB per workgroup) and of a BURST of 4 tiles = 256 KiB per workgroup =
// 8 MiB per XCD -- one transform's results -- timed in the kernel from the first store's issue to the last store's completion.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_nt(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }

// clocks[block * 2 + 0 / 1] = wall clock (100 MHz) at the first store's issue / when the last store of the workgroup has completed
template <int STRIDED>
__global__ __launch_bounds__(512) void k_write(u32x4* out, long long* clocks, int tiles, unsigned mask, int pause_ticks) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 15u;
    if (!((mask >> x) & 1u)) return;
    const unsigned tid = threadIdx.x;
    const u32x4 v = {tid, 1u, 2u, 3u};
    // the XCD's workgroups side by side: a dense index within the XCD is not known -- the block index mod 32 stands in (round-robin
    // dispatch puts blocks b, b + 8, .. on one XCD, so b / 8 is dense there)
    const long long seat = blockIdx.x / 8, xcd_base = (long long)(blockIdx.x % 8) * ((long long)tiles * 32 * 4096);
    if (tid == 0) clocks[blockIdx.x * 2] = (long long)wall_clock64();
    for (int t = 0; t < tiles; t++) {
        if (STRIDED) {
            // tile t = rows 256 t .. 256 t + 255 of the XCD's region (rows of 8 KiB = 512 pieces), my 16 pieces (256 bytes) of each
            u32x4* row0 = out + xcd_base + (long long)t * 256 * 512 + seat * 16;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int piece = i * 512 + (int)tid;  // 4096 pieces of the tile: row = piece / 16, 16 pieces per row
                st_nt(row0 + (long long)(piece >> 4) * 512 + (piece & 15), v);
            }
        } else {
            u32x4* dst = out + xcd_base + (seat * tiles + t) * 4096 + tid;
#pragma unroll
            for (int i = 0; i < 8; i++) st_nt(dst + i * 512, v);
        }
        if (pause_ticks) {  // spread the stores: a pause of this many 100 MHz ticks behind every tile
            const long long t0 = (long long)wall_clock64();
            while ((long long)wall_clock64() - t0 < pause_ticks) __builtin_amdgcn_s_sleep(8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) clocks[blockIdx.x * 2 + 1] = (long long)wall_clock64();
}

int main() {
    const int nwg = 256;
    u32x4* out; long long* clocks;
    const size_t bytes = (size_t)nwg * 256 * 65536;  // 4 GiB
    CK(hipMalloc(&out, bytes));
    CK(hipMemset(out, 0, bytes));
    CK(hipMalloc(&clocks, nwg * 2 * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned masks[4] = {0x01u, 0x03u, 0x0fu, 0xffu};
    long long h[nwg * 2];
    for (int strided = 0; strided < 2; strided++)
        for (int tiles : {256, 4})
            for (int m = 0; m < 4; m++) {
                float best = 1e30f;
                double burst_us = 0;
                for (int rep = 0; rep < 4; rep++) {
                    CK(hipMemset(clocks, 0, nwg * 2 * sizeof(long long)));
                    CK(hipEventRecord(e0));
                    if (strided) hipLaunchKernelGGL(k_write<1>, dim3(nwg), dim3(512), 0, 0, out, clocks, tiles, masks[m], 0);
                    else hipLaunchKernelGGL(k_write<0>, dim3(nwg), dim3(512), 0, 0, out, clocks, tiles, masks[m], 0);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) {
                        best = ms;
                        CK(hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost));
                        long long first = 0, last = 0; int live = 0;
                        for (int b = 0; b < nwg; b++) if (h[2 * b + 1]) { if (!live || h[2 * b] < first) first = h[2 * b]; if (h[2 * b + 1] > last) last = h[2 * b + 1]; live++; }
                        burst_us = (double)(last - first) / 100.0;
                    }
                }
                const int nx = __builtin_popcount(masks[m]);
                const double total = (double)nx * 32 * tiles * 65536.0;
                printf("%-10s %3d tiles per CU (%4.0f MiB per XCD), %d XCD(s) writing: in-kernel %8.2f us = %7.1f GB/s = %6.1f GB/s per XCD = %5.1f GB/s per CU (launch %.3f ms)\n",
                       strided ? "strided" : "contiguous", tiles, 32.0 * tiles / 16, nx, burst_us, total / burst_us / 1e3, total / burst_us / 1e3 / nx,
                       total / burst_us / 1e3 / nx / 32, best);
            }
    return 0;
}
