// membench5.hip -- does a hand-over window stay in an XCD's 4 MiB L2 while the CUs of that XCD stream HBM through it?
// (measurement tool, not part of the product).  One 512-thread workgroup per CU.  Per iteration every workgroup
//   - streams 64 KiB in from its own HBM region and 64 KiB out to its own HBM region (cache-policy bits per MODE),
//   - writes its 1/32 slice of its XCD's window (plain 16-byte stores) and reads another slice (sc1 loads),
// i.e. per XCD 2 MiB in + 2 MiB out + W written + W read, the shape of one row phase of the team kernel.
// Run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`: a resident window shows up as traffic = the streams only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ u32x4 ld_stream(const u32x4* p) {
    u32x4 v;
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE> __device__ __forceinline__ void st_stream(u32x4* p, u32x4 v) {
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 ld_window(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// LDS-DMA flavours of the same loads (what the team kernel issues): 64 lanes x 16 bytes land at M0 + 16 * lane
template <int MODE> __device__ __forceinline__ void dma_stream(const u32x4* p, unsigned lds) {
    const unsigned a = __builtin_amdgcn_readfirstlane(lds);
    unsigned saved;
    if (MODE == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
    if (MODE == 1) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
    if (MODE == 2) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
    if (MODE == 3) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1 nt\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
    if (MODE == 4) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1 nt\n\ts_mov_b32 m0, %0" : "=&s"(saved) : "v"(p), "s"(a) : "memory");
}

// in / out: 256 regions of `region` bytes; win: 8 windows of 4 MiB; seats: 8 counters
template <int LM, int SM, int DMA>
__global__ __launch_bounds__(512) void k_phase(const u32x4* in, u32x4* out, u32x4* win, unsigned* seats, unsigned* sink,
                                               long long region16, int iters, int wslice16 /* 16-byte units per workgroup slice */, int K, int pat) {
    __shared__ unsigned s_seat, s_xcc;
    __shared__ u32x4 s_land[DMA ? 4096 : 1];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 7u;
        s_xcc = x;
        s_seat = atomicAdd(&seats[x], 1u) & 31u;
    }
    __syncthreads();
    const unsigned seat = s_seat, xcc = s_xcc, tid = threadIdx.x;
    const u32x4* src = in + (long long)blockIdx.x * region16;
    u32x4* dst = out + (long long)blockIdx.x * region16;
    u32x4* const w0 = win + (long long)xcc * (4ll << 20) / 16;
    const unsigned lds0 = (unsigned)(size_t)s_land + (tid & ~63u) * 16u;
    u32x4 acc = {0, 0, 0, 0};
    const long long tiles = region16 / 4096;  // 64 KiB tiles in a region
    // pat != 0: the team kernel's n = 2^20 access shape instead of contiguous tiles: the XCD's 32 workgroups share one
    // 8 MiB "transform" (1024 rows of 8 KiB) per 4 iterations; per iteration a workgroup moves 1024 segments of 64 bytes,
    // one per row, at block b of the row's 128 blocks.  pat 1: b = 32 * (it % 4) + seat (the kernel's mapping: a phase
    // covers a 2 KiB band of every row); pat 2: b = 4 * seat + it % 4 (a phase covers every second 64 bytes of the row).
    const long long xreg16 = 32 * region16;  // the XCD's share of in / out
    const u32x4* xsrc = in + (long long)xcc * xreg16;
    u32x4* xdst = out + (long long)xcc * xreg16;
    for (int it = 0; it < iters; it++) {
        const long long t = (it % tiles) * 4096;
        u32x4* const w = w0 + (long long)(it % K) * 32 * wslice16;  // K windows in turn: read what was written K iterations ago, then rewrite it
        u32x4 v[8];
        const int ph = it & 3;
        const int blk = pat == 2 ? 4 * (int)seat + ph : 32 * ph + (int)seat;
        const long long tr16 = (long long)((it >> 2) % (int)(xreg16 / (512 * 1024))) * (512 * 1024);  // 8 MiB transforms
        const long long seg0 = tr16 + (long long)(tid >> 2) * 512 + blk * 4 + (tid & 3);  // row tid/4, 16-byte unit of its segment
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u32x4* a = pat ? xsrc + seg0 + (long long)i * 128 * 512 : src + t + i * 512 + tid;
            if (DMA) { dma_stream<LM>(a, lds0 + i * 8192u); v[i] = u32x4{(unsigned)it, tid, 0u, 0u}; }
            else v[i] = ld_stream<LM>(a);
        }
        // window: write own slice, read the slice of the seat 7 further on (written one iteration ago)
        u32x4 x[16];
        const int per = wslice16 / 512;  // 16-byte stores per thread (0 .. 16)
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (i < per) {
                if (DMA) { dma_stream<2>(w + (long long)((seat + 7u) & 31u) * wslice16 + i * 512 + tid, lds0 + (i & 7) * 8192u); x[i] = u32x4{0u, 0u, 0u, 0u}; }
                else x[i] = ld_window(w + (long long)((seat + 7u) & 31u) * wslice16 + i * 512 + tid);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) { if (i < 8) acc ^= v[i]; if (i < per) acc ^= x[i]; }
#pragma unroll
        for (int i = 0; i < 8; i++) { u32x4 o = v[i] + acc; st_stream<SM>(pat ? xdst + seg0 + (long long)i * 128 * 512 : dst + t + i * 512 + tid, o); }
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (i < per) { u32x4 o = v[i & 7] ^ acc; asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(w + (long long)seat * wslice16 + i * 512 + tid), "v"(o) : "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DMA) { __syncthreads(); acc ^= s_land[tid]; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <int LM, int SM, int DMA>
void run(const char* name, const u32x4* in, u32x4* out, u32x4* win, unsigned* seats, unsigned* sink, long long region, hipEvent_t e0, hipEvent_t e1, int K = 1, int pat = 0) {
    const int iters = 400;
    for (int wk : {0, 16, 32, 64}) {  // KiB per workgroup slice: window = 32 slices = 0, 0.5, 1, 2, 3, 4 MiB per XCD
        if (K * wk > 128) continue;  // 4 MiB of window space per XCD
        CK(hipMemset(seats, 0, 64));
        k_phase<LM, SM, DMA><<<256, 512>>>(in, out, win, seats, sink, region / 16, 20, wk * 1024 / 16, K, pat);
        CK(hipMemset(seats, 0, 64));
        CK(hipEventRecord(e0));
        k_phase<LM, SM, DMA><<<256, 512>>>(in, out, win, seats, sink, region / 16, iters, wk * 1024 / 16, K, pat);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_it = ms * 1e3 / iters;
        const double stream = 2.0 * 256 * 65536 * iters, wbytes = 2.0 * 256 * wk * 1024 * iters;
        printf("%-28s pat %d  %d x window %4.1f MiB/XCD: %6.2f us per iteration | streams %7.1f GB/s | window %7.1f GB/s | expected traffic if resident: %.3f GB, if not: %.3f GB\n",
               name, pat, K, wk * 32 / 1024.0, us_it, stream / ms / 1e6, wbytes / ms / 1e6, stream / 1e9, (stream + wbytes) / 1e9);
    }
}

int main() {
    const long long region = 32ll << 20;  // per workgroup, in and out: 8 GiB each
    u32x4 *in, *out, *win; unsigned *seats, *sink;
    CK(hipMalloc(&in, 256 * region)); CK(hipMalloc(&out, 256 * region)); CK(hipMalloc(&win, 32ll << 20));
    CK(hipMalloc(&seats, 64)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(in, 1, 256 * region)); CK(hipMemset(out, 2, 256 * region)); CK(hipMemset(win, 3, 32ll << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pat : {0, 1, 2}) {
        run<0, 0, 1>("DMA plain / plain", in, out, win, seats, sink, region, e0, e1, 2, pat);
        run<1, 1, 1>("DMA nt / nt", in, out, win, seats, sink, region, e0, e1, 2, pat);
        run<1, 0, 1>("DMA nt / plain", in, out, win, seats, sink, region, e0, e1, 2, pat);
    }
    CK(hipGetLastError());
    return 0;
}
