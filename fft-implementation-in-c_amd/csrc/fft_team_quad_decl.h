// fft_team_quad_decl.h -- declaration of team_quad_kernel (fft_team_quad.h).  The device build compiles the kernel in a
// translation unit of its own (fft_team_quad.hip: seconds instead of the backend's minutes); the planner (fft_engine.h) only
// needs the declaration to launch it.  The CPU emulation includes the body directly.
#pragma once

#include "fft_team.h"

namespace fftk {

#if defined(FFT_EMU)
#define FFT_QUAD_BOUNDS(E, LOG2L, LOG2TS)
#else
#define FFT_QUAD_BOUNDS(E, LOG2L, LOG2TS) __launch_bounds__(((1 << (LOG2L)) >> (LOG2TS)) * (E), (((1 << (LOG2L)) >> (LOG2TS)) * (E)) / 256)
#endif

template <typename T, int E, int LOG2L, int LOG2TS>
FFT_KERNEL void FFT_QUAD_BOUNDS(E, LOG2L, LOG2TS) team_quad_kernel(TeamParams<T> p);

#if !defined(FFT_EMU)
extern template __global__ void team_quad_kernel<float, 16, 10, 5>(TeamParams<float>);
extern template __global__ void team_quad_kernel<float, 16, 10, 6>(TeamParams<float>);
#endif

}  // namespace fftk
