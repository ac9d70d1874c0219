// membench3.hip -- does the 256 MiB Infinity Cache keep a scratch buffer between two kernels? (measurement tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
struct alignas(16) V16 { unsigned w[4]; };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_copy(const V16* in, V16* out, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_read(const V16* in, unsigned* sink, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { V16 v = in[i]; acc ^= v.w[0] ^ v.w[1] ^ v.w[2] ^ v.w[3]; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(V16* out, long long n16, unsigned tag) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    V16 v; v.w[0] = threadIdx.x; v.w[1] = tag; v.w[2] = 2; v.w[3] = 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = v;
}
int main() {
    const long long total = 4ll << 30;
    V16 *a, *b, *s; unsigned* sink;
    CK(hipMalloc(&a, total)); CK(hipMalloc(&b, total)); CK(hipMalloc(&s, 1ll << 30)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, total)); CK(hipMemset(b, 2, total)); CK(hipMemset(s, 3, 1ll << 30));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int G = 1024;
    for (long long mb : {16, 32, 64, 128, 192, 256, 512, 1024}) {
        const long long S = mb << 20, n16 = S / 16;
        float ms;
        // (a) repeated reads of the same S bytes
        const int reps = (int)((8ll << 30) / S);
        k_read<<<G, 256>>>(s, sink, n16);
        CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) k_read<<<G, 256>>>(s, sink, n16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double rr = (double)S * reps / ms / 1e6;
        // (b) write S then read S, alternating
        CK(hipEventRecord(e0)); for (int r = 0; r < reps / 2; r++) { k_write<<<G, 256>>>(s, n16, r); k_read<<<G, 256>>>(s, sink, n16); } CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double wr = (double)S * 2 * (reps / 2) / ms / 1e6;
        // (c) the FFT pipeline shape: HBM stream -> scratch (kernel 1), scratch -> HBM stream (kernel 2); 4 GiB through an S-byte scratch
        const int chunks = (int)(total / S);
        CK(hipEventRecord(e0));
        for (int c = 0; c < chunks; c++) { k_copy<<<G, 256>>>(a + c * n16, s, n16); k_copy<<<G, 256>>>(s, b + c * n16, n16); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double pipe_alg = 2.0 * total / ms / 1e6;
        printf("scratch %5lld MiB: re-read %7.1f GB/s | write-then-read %7.1f GB/s | pipeline in->scratch->out: %.3f ms, algorithmic %7.1f GB/s (moved %7.1f GB/s)\n", mb, rr, wr, ms, pipe_alg, 2 * pipe_alg);
    }
    CK(hipGetLastError());
    return 0;
}
