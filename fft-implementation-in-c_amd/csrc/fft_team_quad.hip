// fft_team_quad.hip -- the device instantiations of team_quad_kernel (fft_team_quad.h): fp32, E = 16 values per thread and chunk,
// 512-thread workgroups, one per CU.
#include "fft_team_quad.h"
#include "fft_wide_row.h"

namespace fftk {
#define FFT_QUAD_DEFINE(T, ...) template __global__ void team_quad_kernel<T, __VA_ARGS__>(TeamParams<T>);
FFT_QUAD_INSTANCES(FFT_QUAD_DEFINE)
#undef FFT_QUAD_DEFINE
template __global__ void wide_row_kernel<float, 13>(WideParams<float>);  // single-pass n = 8192 (fft_wide_row.h)
template __global__ void wide_row_kernel<float, 14>(WideParams<float>);  // single-pass n = 16384
template __global__ void wide_row_kernel<double, 13, 16>(WideParams<double>);  // fp64 n = 8192: 512 threads, one 128 KiB image in place
}
