// fft_team_quad.hip -- the device instantiation of team_quad_kernel (fft_team_quad.h): fp32 n = 2^20, E = 16, teams of 32.
#include "fft_team_quad.h"

namespace fftk {
template __global__ void team_quad_kernel<float, 16, 10, 5>(TeamParams<float>);  // one 512-thread workgroup per CU, teams of 32
template __global__ void team_quad_kernel<float, 16, 10, 6>(TeamParams<float>);  // two 256-thread workgroups per CU, teams of 64
}
