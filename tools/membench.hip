// membench.hip -- HBM access-pattern ceilings for the FFT tile kernels (measurement tool, not product).
// Each workgroup (512 threads) moves one 128 KiB tile: `rows` row segments of `seg` bytes.
//   mode 0: contiguous read  -> contiguous write
//   mode 1: strided   read  -> contiguous write   (pass A read side)
//   mode 2: contiguous read -> strided   write    (pass B write side)
//   mode 3: strided   read  -> strided   write    (pass A in place)
// "strided": tile t of matrix m covers bytes [ct*seg, (ct+1)*seg) of every row; row pitch = pitch bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

struct alignas(16) V16 { unsigned w[4]; };

__global__ __launch_bounds__(512) void tile_copy(const char* in, char* out, int seg, int pitch, int rows_per_mat,
                                                 int mode, long long mat_bytes) {
    const int tile_bytes = 128 * 1024;
    const int rows = tile_bytes / seg;               // rows per tile
    const int tiles_per_mat_row = pitch / seg;       // column tiles across a row
    const int row_groups = rows_per_mat / rows;      // tiles stacked vertically
    long long t = blockIdx.x;
    const int ct = t % tiles_per_mat_row; t /= tiles_per_mat_row;
    const int rg = t % row_groups; t /= row_groups;
    const long long m = t;
    const int lanes_per_row = seg / 16;
    const int tid = threadIdx.x;
    V16 v[16];
    const long long contig = (m * (long long)row_groups * tiles_per_mat_row + (long long)rg * tiles_per_mat_row + ct) * tile_bytes;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int g = tid + e * 512;  // 16-byte chunk index in tile
        long long off;
        if (mode == 1 || mode == 3) {
            const int r = g / lanes_per_row, c = g % lanes_per_row;
            off = m * mat_bytes + ((long long)rg * rows + r) * pitch + (long long)ct * seg + c * 16;
        } else {
            off = contig + (long long)g * 16;
        }
        v[e] = *(const V16*)(in + off);
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int g = tid + e * 512;
        long long off;
        if (mode == 2 || mode == 3) {
            const int r = g / lanes_per_row, c = g % lanes_per_row;
            off = m * mat_bytes + ((long long)rg * rows + r) * pitch + (long long)ct * seg + c * 16;
        } else {
            off = contig + (long long)g * 16;
        }
        *(V16*)(out + off) = v[e];
    }
}

__global__ void fill(float* p, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = (float)(i & 1023);
}

int main(int argc, char** argv) {
    const long long total = 4ll << 30;  // 4 GiB buffers
    char *a, *b;
    hipMalloc(&a, total); hipMalloc(&b, total);
    fill<<<4096, 256>>>((float*)a, total / 4); fill<<<4096, 256>>>((float*)b, total / 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int pitch = 8192;               // 1024 complex64 per row
    const int rows_per_mat = 1024;        // 8 MiB matrices (one 2^20-point transform)
    const long long mat_bytes = (long long)pitch * rows_per_mat;
    const long long nmats = total / mat_bytes;
    printf("mode seg_bytes  GB/s(read+write)  ms\n");
    for (int mode = 0; mode < 4; mode++) {
        for (int seg = 128; seg <= 8192; seg *= 2) {
            if (mode == 0 && seg != 128) continue;
            const long long grid = total / (128 * 1024);
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                tile_copy<<<(unsigned)grid, 512>>>(a, b, seg, pitch, rows_per_mat, mode, mat_bytes);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) printf("%d %5d  %8.1f  %.3f\n", mode, seg, 2.0 * total / ms / 1e6, ms);
            }
        }
    }
    // chunked launches like the engine: 64 MiB per launch (8 matrices), mode 3 at seg 128, and ping-pong through a 64 MiB scratch
    {
        const long long chunk = 64ll << 20;
        const unsigned grid = (unsigned)(chunk / (128 * 1024));
        char* scratch; hipMalloc(&scratch, chunk);
        for (int variant = 0; variant < 3; variant++) {
            hipEventRecord(e0);
            int launches = 0;
            for (long long off = 0; off + chunk <= total; off += chunk) {
                if (variant == 0) {  // A: strided read -> contiguous scratch ; B: contiguous scratch -> strided write
                    tile_copy<<<grid, 512>>>(a + off, scratch, 128, pitch, rows_per_mat, 1, mat_bytes);
                    tile_copy<<<grid, 512>>>(scratch, b + off, 128, pitch, rows_per_mat, 2, mat_bytes);
                } else if (variant == 1) {  // same but scratch is a full-size buffer region (no MALL reuse)
                    tile_copy<<<grid, 512>>>(a + off, b + off, 128, pitch, rows_per_mat, 1, mat_bytes);
                    tile_copy<<<grid, 512>>>(b + off, a + off, 128, pitch, rows_per_mat, 2, mat_bytes);
                } else {  // contiguous both through scratch: the ceiling of a 2-kernel pipeline with MALL-resident scratch
                    tile_copy<<<grid, 512>>>(a + off, scratch, 128, pitch, rows_per_mat, 0, mat_bytes);
                    tile_copy<<<grid, 512>>>(scratch, b + off, 128, pitch, rows_per_mat, 0, mat_bytes);
                }
                launches += 2;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("pipeline variant %d: %d launches, %.3f ms, algorithmic (1 read + 1 write of 4 GiB) %.1f GB/s\n", variant, launches, ms, 2.0 * total / ms / 1e6);
        }
    }
    return 0;
}
