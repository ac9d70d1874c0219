// membench4.hip -- what one CU's vector-memory pipeline carries, by access shape (measurement tool).
//   One 512-thread workgroup per CU streams 64 KiB tiles, back to back, the way the team kernel does:
//     dma   LDS-DMA (global_load_lds_dwordx4), 8 instructions per thread per tile, tile = ROWS rows of SEG bytes,
//           rows `stride` bytes apart (SEG = 64: the n = 2^20 column tile; 128: n = 2^18; 65536: a contiguous row tile)
//     st    plain 16-byte stores of the same shapes
//   from a 4 GiB buffer (HBM), a 64 MiB one (Infinity Cache) and a 2 MiB one per XCD-sized slice (L2).
// Prints GB/s per CU and chip-wide.  hipcc --offload-arch=gfx950 -O3 tools/membench4.hip -o /tmp/membench4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
struct alignas(16) V16 { unsigned w[4]; };

__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
    const unsigned a = __builtin_amdgcn_readfirstlane(lds);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(a) : "memory");
}

// tile t of this workgroup: rows of `seg` bytes; chunk g = i*512 + tid -> row g / (seg/16), piece g % (seg/16)
template <int STORE>
__global__ __launch_bounds__(512) void k_stream(unsigned char* base, long long region, int log2seg16, long long row_stride,
                                                int rows, int tiles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const int tid = threadIdx.x;
    const int piece = tid & ((1 << log2seg16) - 1), row0 = tid >> log2seg16;
    const int rows_per_i = 512 >> log2seg16;
    const long long seg = 16ll << log2seg16;
    const long long block = (long long)rows * row_stride;           // bytes spanned by the rows of one tile
    const long long sweeps = row_stride / seg;                       // tiles that share one block of rows
    unsigned char* mine = base + (long long)blockIdx.x * region;     // every workgroup streams its own region (region 0: all share 8 MiB)
    if (region == 0) region = 8ll << 20;
    V16 v; v.w[0] = tid; v.w[1] = 1; v.w[2] = 2; v.w[3] = 3;
    for (int t = 0; t < tiles; t++) {
        const long long tb = ((t / sweeps) * block) % region + (t % sweeps) * seg;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            unsigned char* p = mine + tb + (long long)(row0 + i * rows_per_i) * row_stride + piece * 16;
            if (STORE) *reinterpret_cast<V16*>(p) = v;
            else dma16(p, lds0 + (unsigned)(i * 512 + tid) * 16u);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // at most one more tile in flight, as in the kernel
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!STORE && lds[tid] == 0x7f && lds[tid + 1] == 0x7e) sink[0] = 1;
}

int main() {
    const long long big = 4ll << 30;
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 1, big));
    CK(hipFuncSetAttribute((const void*)k_stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)k_stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Shape { const char* name; int log2seg16; long long row_stride; int rows; };
    const Shape shapes[] = {
        {"64-B segments, 1024 rows 8 KiB apart (n=2^20 column tile)", 2, 8192, 1024},
        {"128-B segments, 512 rows 4 KiB apart (n=2^18 column tile)", 3, 4096, 512},
        {"256-B segments, 256 rows 2 KiB apart", 4, 2048, 256},
        {"contiguous 64 KiB (row tile / hand-over window)", 9, 8192, 8},
    };
    // region per workgroup: 16 MiB (4 GiB in all: HBM), 256 KiB x shapes that fit (64 MiB in all: Infinity Cache)
    const struct { const char* name; long long region; } spans[] = {{"HBM, 16 MiB per workgroup", 16ll << 20}, {"8 MiB shared by all (Infinity Cache / L2)", 0}};
    const int tiles = 256;
    for (const auto& sp : spans)
        for (const auto& sh : shapes)
            for (int store = 0; store < 2; store++) {
                float ms;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    // region 0: every workgroup walks the same 8 MiB (block % region with region = 8 MiB, no per-workgroup offset)
                    const long long region = sp.region ? sp.region : (8ll << 20);
                    unsigned char* b = buf;
                    if (store) hipLaunchKernelGGL(k_stream<1>, dim3(256), dim3(512), 131072, 0, b, sp.region ? region : 0, sh.log2seg16, sh.row_stride, sh.rows, tiles, sink);
                    else hipLaunchKernelGGL(k_stream<0>, dim3(256), dim3(512), 131072, 0, b, sp.region ? region : 0, sh.log2seg16, sh.row_stride, sh.rows, tiles, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                }
                const double gbs = 65536.0 * tiles * 256 / ms / 1e6;
                printf("%-42s %-60s %-5s %6.1f GB/s per CU  %7.1f GB/s chip  (%.2f us per tile)\n", sp.name, sh.name, store ? "store" : "dma",
                       gbs / 256, gbs, ms * 1e3 / tiles);
            }
    CK(hipGetLastError());
    return 0;
}
