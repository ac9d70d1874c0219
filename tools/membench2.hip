// membench2.hip -- which copy-kernel STRUCTURE reaches the HBM ceiling on MI355X (measurement tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct alignas(16) V16 { unsigned w[4]; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// A: classic grid-stride copy, UNROLL x 16 B per thread per iteration
template <int UNROLL>
__global__ __launch_bounds__(256) void gs_copy(const V16* in, V16* out, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        V16 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) out[i + u * stride] = v[u];
    }
    for (; i < n16; i += stride) out[i] = in[i];
}

__global__ __launch_bounds__(256) void gs_read(const V16* in, unsigned* sink, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { V16 v = in[i]; acc ^= v.w[0] ^ v.w[1] ^ v.w[2] ^ v.w[3]; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void gs_write(V16* out, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    V16 v; v.w[0] = threadIdx.x; v.w[1] = 1; v.w[2] = 2; v.w[3] = 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = v;
}

// B: tile copy. THREADS threads move a tile of THREADS*E*16 bytes: E loads then E stores per thread.
//    persistent = 0: one tile per workgroup.  persistent = 1: grid-stride over tiles, next tile's loads are issued
//    before the current tile's stores (software pipelining), as the FFT kernel would prefetch.
template <int THREADS, int E, int PIPE>
__global__ __launch_bounds__(THREADS) void tile_copy(const V16* in, V16* out, long long ntiles, int lds_bytes_dummy) {
    extern __shared__ unsigned char smem[];
    if (lds_bytes_dummy < 0) smem[threadIdx.x] = 1;  // keep the LDS allocation alive
    const int tid = threadIdx.x;
    if (!PIPE) {
        for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const long long base = t * (long long)(THREADS * E);
            V16 v[E];
#pragma unroll
            for (int e = 0; e < E; e++) v[e] = in[base + tid + e * THREADS];
#pragma unroll
            for (int e = 0; e < E; e++) out[base + tid + e * THREADS] = v[e];
        }
    } else {
        long long t = blockIdx.x;
        V16 cur[E], nxt[E];
        if (t < ntiles) {
            const long long base = t * (long long)(THREADS * E);
#pragma unroll
            for (int e = 0; e < E; e++) cur[e] = in[base + tid + e * THREADS];
        }
        for (; t < ntiles; t += gridDim.x) {
            const long long tn = t + gridDim.x;
            if (tn < ntiles) {
                const long long nb = tn * (long long)(THREADS * E);
#pragma unroll
                for (int e = 0; e < E; e++) nxt[e] = in[nb + tid + e * THREADS];
            }
            const long long base = t * (long long)(THREADS * E);
#pragma unroll
            for (int e = 0; e < E; e++) out[base + tid + e * THREADS] = cur[e];
#pragma unroll
            for (int e = 0; e < E; e++) cur[e] = nxt[e];
        }
    }
}

static hipEvent_t e0, e1;
template <class F> static float timeit(F f) {
    f();
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError());
    return ms;
}

int main() {
    const long long total = 4ll << 30;
    const long long n16 = total / 16;
    V16 *a, *b; unsigned* sink;
    CK(hipMalloc(&a, total)); CK(hipMalloc(&b, total)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, total)); CK(hipMemset(b, 2, total));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        ms = timeit([&] { gs_copy<1><<<blocks, 256>>>(a, b, n16); }); printf("gs_copy<1>  blocks=%5d  %7.1f GB/s (r+w)\n", blocks, 2.0 * total / ms / 1e6);
        ms = timeit([&] { gs_copy<4><<<blocks, 256>>>(a, b, n16); }); printf("gs_copy<4>  blocks=%5d  %7.1f GB/s (r+w)\n", blocks, 2.0 * total / ms / 1e6);
        ms = timeit([&] { gs_copy<8><<<blocks, 256>>>(a, b, n16); }); printf("gs_copy<8>  blocks=%5d  %7.1f GB/s (r+w)\n", blocks, 2.0 * total / ms / 1e6);
    }
    ms = timeit([&] { gs_read<<<256 * 16, 256>>>(a, sink, n16); }); printf("gs_read   %7.1f GB/s\n", 1.0 * total / ms / 1e6);
    ms = timeit([&] { gs_write<<<256 * 16, 256>>>(b, n16); }); printf("gs_write  %7.1f GB/s\n", 1.0 * total / ms / 1e6);
    CK(hipFuncSetAttribute((const void*)tile_copy<512, 16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_copy<512, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_copy<256, 16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_copy<256, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int lds : {0, 40 * 1024, 72 * 1024, 150 * 1024}) {  // LDS footprint limits workgroups per CU: 0 -> by threads, 40K -> 4, 72K -> 2, 150K -> 1
        {
            const long long nt = n16 / (512 * 16);
            ms = timeit([&] { tile_copy<512, 16, 0><<<(unsigned)nt, 512, lds>>>(a, b, nt, lds); }); printf("tile 512x16 one-tile-per-WG   lds=%6d  %7.1f GB/s\n", lds, 2.0 * total / ms / 1e6);
            for (int per_cu : {1, 2, 4}) {
                ms = timeit([&] { tile_copy<512, 16, 0><<<256 * per_cu, 512, lds>>>(a, b, nt, lds); }); printf("tile 512x16 persistent x%d      lds=%6d  %7.1f GB/s\n", per_cu, lds, 2.0 * total / ms / 1e6);
                ms = timeit([&] { tile_copy<512, 16, 1><<<256 * per_cu, 512, lds>>>(a, b, nt, lds); }); printf("tile 512x16 persistent+pipe x%d lds=%6d  %7.1f GB/s\n", per_cu, lds, 2.0 * total / ms / 1e6);
            }
        }
        {
            const long long nt = n16 / (256 * 16);
            ms = timeit([&] { tile_copy<256, 16, 0><<<(unsigned)nt, 256, lds>>>(a, b, nt, lds); }); printf("tile 256x16 one-tile-per-WG   lds=%6d  %7.1f GB/s\n", lds, 2.0 * total / ms / 1e6);
            for (int per_cu : {2, 4, 8}) {
                ms = timeit([&] { tile_copy<256, 16, 1><<<256 * per_cu, 256, lds>>>(a, b, nt, lds); }); printf("tile 256x16 persistent+pipe x%d lds=%6d  %7.1f GB/s\n", per_cu, lds, 2.0 * total / ms / 1e6);
            }
        }
    }
    return 0;
}
