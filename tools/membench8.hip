// membench8.hip -- what does the memory side sustain for team_quad_kernel's TRAFFIC MIX when nothing waits for anybody?  (measurement tool,
// round 4.)  Per transform of n = 2^20 fp32 a CU of the team moves four streams of 256 KiB each through its one vector-memory queue: HBM in
// (LDS-DMA, nt), window out (plain 16-byte stores into the XCD's L2), window in (LDS-DMA, sc1 nt), HBM out (nt stores).  The kernel below issues
// exactly those instructions, 64 KiB of each per "quarter", with the same LDS double buffering -- and NO arithmetic, NO team protocol, NO
// dependency between the streams except the LDS images' reuse: the data is garbage, the traffic is real.  One 512-thread workgroup per CU.
//   mode bit 0: HBM in   bit 1: window out   bit 2: window in   bit 3: HBM out      (15 = all four)
//   window bytes per XCD: a ring of `slots` units of 1 MiB (every workgroup writes half images of the workgroup 8 blocks on -- same XCD under
//   round-robin dispatch -- and reads its own, written `lag` units earlier)
// Prints the time per "transform" (4 quarters) per CU and the Gpoint/s the batch of BASELINE config 3 would run at that pace.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int POL>  // 0: HBM stream, nt; 1: window, sc1 nt
__device__ __forceinline__ void dma(const u32x4* p, unsigned lds) {
    const unsigned a = __builtin_amdgcn_readfirstlane(lds);
    unsigned saved;
    if (POL == 0)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1 nt\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
}
__device__ __forceinline__ void st_nt(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_plain(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
// cache-policy bits chosen at run time (wave-uniform): 0 plain, 1 nt, 2 sc1, 3 sc1 nt, 4 sc0 sc1, 5 sc0 sc1 nt, 6 sc0, 7 sc0 nt
__device__ __forceinline__ void st_pol(u32x4* p, u32x4 v, int pol) {
    switch (pol) {
        case 1: asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); break;
        case 2: asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); break;
        case 3: asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory"); break;
        case 4: asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); break;
        case 5: asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory"); break;
        case 6: asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory"); break;
        case 7: asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory"); break;
        default: asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); break;
    }
}
#define DMA_ASM(bits) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off " bits "\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory")
__device__ __forceinline__ void dma_pol(const u32x4* p, unsigned lds, int pol) {
    const unsigned a = __builtin_amdgcn_readfirstlane(lds);
    unsigned saved;
    switch (pol) {
        case 1: DMA_ASM("nt"); break;
        case 2: DMA_ASM("sc1"); break;
        case 3: DMA_ASM("sc1 nt"); break;
        case 4: DMA_ASM("sc0 sc1"); break;
        case 5: DMA_ASM("sc0 sc1 nt"); break;
        case 6: DMA_ASM("sc0"); break;
        case 7: DMA_ASM("sc0 nt"); break;
        default: DMA_ASM(""); break;
    }
}

__global__ __launch_bounds__(512) void k_mix(const u32x4* in, u32x4* out, u32x4* window, int transforms, int mode, int slots, int lag, long long* clocks, int p_in, int p_wst, int p_wld, int p_out) {
    extern __shared__ u32x4 land[];  // 2 x 64 KiB
    const unsigned tid = threadIdx.x;
    const unsigned lds0 = (unsigned)(size_t)land;
    const long long wg = blockIdx.x, nwg = gridDim.x;
    const u32x4 v = {tid, 1u, 2u, 3u};
    // the window of "my" XCD (blocks b, b + 8, ..): slot x 32 images of 4096 pieces; I write the image of block (b + 8) and read mine
    const long long xcd = wg % 8, seat = wg / 8, seat_to = (seat + 1) % (nwg / 8);
    u32x4* const win = window + xcd * ((long long)slots * (nwg / 8) * 2048);
    if (tid == 0) clocks[wg * 2] = (long long)wall_clock64();
    int im = 0;
    for (int t = 0; t < transforms; t++) {
        // my transform: rows of 8 KiB (512 pieces), my 16 pieces (256 bytes) of each: the column-step and result pattern of n = 2^20
        const long long tr = t;  // (every XCD walks its own transforms: xcd, xcd + 8, ..)
        const u32x4* src_t = in + ((tr * 8 + xcd) % 512) * (1024ll * 512) + seat * 16;
        u32x4* dst_t = out + ((tr * 8 + xcd) % 512) * (1024ll * 512) + seat * 16;
        for (int q = 0; q < 4; q++, im ^= 1) {
            const unsigned base = lds0 + (unsigned)im * 65536u;
            if (mode & 1) {  // HBM in: 256 rows (every fourth of the 1024) x 256 bytes
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int piece = i * 512 + (int)tid;
                    dma_pol(src_t + (long long)(4 * (piece >> 4) + q) * 512 + (piece & 15), base + (unsigned)(i * 512 + (tid & ~63u)) * 16u, p_in);
                }
            }
            // the window as a ring of `slots` units of 1 MiB per XCD (32 KiB per workgroup and unit): a quarter writes two units and
            // reads the two units written `lag` units earlier
            const long long h0 = ((long long)t * 4 + q) * 2;
            if (mode & 2) {  // window out: 2 x 32 KiB into a neighbour's half images, 1 KiB runs
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    u32x4* w = win + (((h0 + (i >> 2)) % slots) * (nwg / 8) + seat_to) * 2048 + tid;
                    st_pol(w + (i & 3) * 512, v, p_wst);
                }
            }
            if (mode & 4) {  // window in: my half images, written `lag` units ago
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const u32x4* w = win + (((h0 + (i >> 2) + slots - lag % slots) % slots) * (nwg / 8) + seat) * 2048 + tid;
                    dma_pol(w + (i & 3) * 512, base + (unsigned)(i * 512 + (tid & ~63u)) * 16u, p_wld);
                }
            }
            if (mode & 8) {  // HBM out: 256 rows x 256 bytes, 16-byte stores, nt
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int piece = i * 512 + (int)tid;
                    st_pol(dst_t + (long long)(4 * (piece >> 4) + q) * 512 + (piece & 15), v, p_out);
                }
            }
            // the image written two quarters ago is reused next: everything older than this quarter's instructions has completed
            switch (__builtin_popcount(mode & 15)) {
                case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
            }
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) clocks[wg * 2 + 1] = (long long)wall_clock64();
}

int main() {
    const int nwg = 256, transforms = 64;  // 64 "transforms" per XCD = BASELINE config 3's 512 over 8 XCDs
    u32x4 *in, *out, *window; long long* clocks;
    const size_t tbytes = 512ull * 1024 * 512 * 16;  // 512 transforms of 8 MiB
    CK(hipMalloc(&in, tbytes)); CK(hipMalloc(&out, tbytes));
    CK(hipMemset(in, 1, tbytes)); CK(hipMemset(out, 0, tbytes));
    CK(hipMalloc(&window, 8ull * 8 * 32 * 32768)); CK(hipMemset(window, 0, 8ull * 8 * 32 * 32768));
    CK(hipMalloc(&clocks, nwg * 2 * sizeof(long long)));
    CK(hipFuncSetAttribute((const void*)k_mix, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { int mode, slots, lag; const char* name; } cases[] = {
        {15, 4, 4, "all four streams, window 4 MiB per XCD"}, {15, 3, 3, "all four streams, window 3 MiB per XCD"},
        {15, 2, 2, "all four streams, window 2 MiB per XCD"}, {15, 1, 1, "all four streams, window 1 MiB per XCD"},
        {15, 6, 6, "all four streams, window 6 MiB per XCD"}, {15, 8, 8, "all four streams, window 8 MiB per XCD"},
        {15, 4, 2, "all four, ring of 4 MiB, read 2 units behind"}, {15, 4, 1, "all four, ring of 4 MiB, read 1 unit behind"},
        {9, 2, 2, "HBM in + HBM out only"}, {6, 4, 4, "window out + in only (4 MiB)"}, {6, 2, 2, "window out + in only (2 MiB)"},
        {1, 2, 2, "HBM in only"}, {8, 2, 2, "HBM out only"}, {7, 4, 4, "HBM in + window (4 MiB), no HBM out"}, {14, 4, 4, "window (4 MiB) + HBM out, no HBM in"},
        {7, 2, 2, "HBM in + window (2 MiB), no HBM out"}, {14, 2, 2, "window (2 MiB) + HBM out, no HBM in"}};
    for (auto& c : cases) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mix, dim3(nwg), dim3(512), 131072, 0, in, out, window, transforms, c.mode, c.slots, c.lag, clocks, 1, 0, 3, 1);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double us_per_transform = best * 1e3 / transforms;
        printf("%-48s %7.3f ms per 512 transforms = %6.2f us per transform and XCD = %6.1f Gpoint/s at that pace (%4.1f %% of the 8 TB/s roof)\n", c.name, best,
               us_per_transform, 512.0 * 1048576 / best / 1e6, 512.0 * 1048576 * 16 / best / 1e6 / 80);
    }
    // cache-policy bits of the four streams at 4 MiB and at 3 MiB of window (the kernel's: in nt, window stores plain, window loads sc1 nt, out nt)
    const char* pn[8] = {"plain", "nt", "sc1", "sc1 nt", "sc0 sc1", "sc0 sc1 nt", "sc0", "sc0 nt"};
    for (int slots : {4, 3}) {
        printf("---- policy matrix, window %d MiB: in | window store | window load | out : ms\n", slots);
        for (int p_in : {1, 0, 3, 5})
            for (int p_wst : {0, 1})
                for (int p_wld : {3, 2, 1, 0})
                    for (int p_out : {1, 0, 2, 3, 4, 5}) {
                        float best = 1e30f;
                        for (int rep = 0; rep < 3; rep++) {
                            CK(hipEventRecord(e0));
                            hipLaunchKernelGGL(k_mix, dim3(nwg), dim3(512), 131072, 0, in, out, window, transforms, 15, slots, slots, clocks, p_in, p_wst, p_wld, p_out);
                            CK(hipEventRecord(e1));
                            CK(hipEventSynchronize(e1));
                            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                            if (rep > 0 && ms < best) best = ms;
                        }
                        printf("%-10s | %-5s | %-10s | %-10s : %6.3f\n", pn[p_in], pn[p_wst], pn[p_wld], pn[p_out], best);
                    }
    }
    return 0;
}
