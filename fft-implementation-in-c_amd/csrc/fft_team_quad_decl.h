// fft_team_quad_decl.h -- declaration of team_quad_kernel (fft_team_quad.h).  The device build compiles the kernel in a
// translation unit of its own (fft_team_quad.hip: seconds instead of the backend's minutes); the planner (fft_engine.h) only
// needs the declaration to launch it.  The CPU emulation includes the body directly.
#pragma once

#include "fft_team.h"

namespace fftk {

#if defined(FFT_EMU)
#define FFT_QUAD_BOUNDS(LOG2R2, LOG2L, LOG2TS)
#else
// NTHR = (L / TS) * R2 threads, all resident on one CU: NTHR / 256 waves per SIMD
#define FFT_QUAD_BOUNDS(LOG2R2, LOG2L, LOG2TS) __launch_bounds__((((1 << (LOG2L)) >> (LOG2TS)) << (LOG2R2)), ((((1 << (LOG2L)) >> (LOG2TS)) << (LOG2R2))) / 256)
#endif
template <typename T, int E, int LOG2R2, int LOG2L, int LOG2TS, int SLOTS>
FFT_KERNEL void FFT_QUAD_BOUNDS(LOG2R2, LOG2L, LOG2TS) team_quad_kernel(TeamParams<T> p);

#if !defined(FFT_EMU)
extern template __global__ void team_quad_kernel<float, 16, 4, 10, 5, 2>(TeamParams<float>);  // n = 2^20: M = 16 x 16, teams of 32 (a whole XCD), two window slots
extern template __global__ void team_quad_kernel<float, 16, 4, 10, 5, 1>(TeamParams<float>);  // ... one (experiments)
extern template __global__ void team_quad_kernel<float, 16, 3, 9, 3, 1>(TeamParams<float>);   // n = 2^18: M = 16 x 8, teams of 8, one window slot
extern template __global__ void team_quad_kernel<float, 16, 2, 8, 1, 1>(TeamParams<float>);   // n = 2^16: M = 16 x 4, teams of 2, one window slot
#endif

}  // namespace fftk
