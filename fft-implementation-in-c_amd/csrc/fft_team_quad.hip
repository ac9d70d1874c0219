// fft_team_quad.hip -- the device instantiations of team_quad_kernel (fft_team_quad.h): fp32, E = 16 values per thread and chunk,
// 512-thread workgroups, one per CU.
#include "fft_team_quad.h"
#include "fft_wide_row.h"

namespace fftk {
template __global__ void team_quad_kernel<float, 16, 4, 10, 5, 2>(TeamParams<float>);  // n = 2^20: 1024 x 1024, teams of 32 (a whole XCD), two window slots
template __global__ void team_quad_kernel<float, 16, 4, 10, 5, 1>(TeamParams<float>);  // ... with one (experiments: FFT_HIP_QUAD_SLOTS20=1; traffic 1.08 x, but -20 %)
template __global__ void team_quad_kernel<float, 16, 3, 9, 3, 1>(TeamParams<float>);   // n = 2^18: 512 x 512, teams of 8, one window slot
template __global__ void team_quad_kernel<float, 16, 2, 8, 1, 1>(TeamParams<float>);    // n = 2^16: 256 x 256, teams of 2, one window slot
template __global__ void wide_row_kernel<float, 13>(WideParams<float>);  // single-pass n = 8192 (fft_wide_row.h)
template __global__ void wide_row_kernel<float, 14>(WideParams<float>);  // single-pass n = 16384
template __global__ void wide_row_kernel<double, 13, 16>(WideParams<double>);  // fp64 n = 8192: 512 threads, one 128 KiB image in place
}
