// fft_team_quad_decl.h -- declaration of team_quad_kernel (fft_team_quad.h).  The device build compiles the kernel in a
// translation unit of its own (fft_team_quad.hip: seconds instead of the backend's minutes); the planner (fft_engine.h) only
// needs the declaration to launch it.  The CPU emulation includes the body directly.
#pragma once

#include "fft_team.h"

namespace fftk {

#if defined(FFT_EMU)
#define FFT_QUAD_BOUNDS(LOG2RA, LOG2L2, LOG2TS)
#else
// NTHR = (L2 / TS) * RA threads, all resident on one CU: NTHR / 256 waves per SIMD
#define FFT_QUAD_BOUNDS(LOG2RA, LOG2L2, LOG2TS) __launch_bounds__((((1 << (LOG2L2)) >> (LOG2TS)) << (LOG2RA)), ((((1 << (LOG2L2)) >> (LOG2TS)) << (LOG2RA))) / 256)
#endif
template <typename T, int E, int LOG2RA, int LOG2RB, int LOG2L1, int LOG2L2, int LOG2TS, int SLOTS>
FFT_KERNEL void FFT_QUAD_BOUNDS(LOG2RA, LOG2L2, LOG2TS) team_quad_kernel(TeamParams<T> p);

#if !defined(FFT_EMU)
// <T, E, log2 RA, log2 RB, log2 L1, log2 L2, log2 TS, window slots>; the list of fft_team_quad.hip
#define FFT_QUAD_INSTANCES(X)                                                                                         \
    X(float, 16, 4, 4, 10, 10, 5, 2) /* n = 2^20: 1024 x 1024, teams of 32 (a whole XCD), two window slots */          \
    X(float, 16, 4, 4, 10, 10, 5, 1) /* ... with one (experiments: FFT_HIP_QUAD_SLOTS=1; traffic 1.08 x, but -20 %) */ \
    X(float, 16, 4, 4, 10, 10, 5, 3) /* ... ONE image per seat and the pair protocol (per-seat counters, no team-wide wait) */ \
    X(float, 16, 4, 3, 10, 9, 4, 2)  /* n = 2^19: 1024 x 512, teams of 16 */                                           \
    X(float, 16, 4, 3, 10, 9, 4, 1)                                                                                    \
    X(float, 16, 4, 3, 10, 9, 4, 3)                                                                                    \
    X(float, 16, 3, 3, 9, 9, 3, 1)   /* n = 2^18: 512 x 512, teams of 8, one window slot */                            \
    X(float, 16, 3, 3, 9, 9, 3, 2)   /* ... two (experiments) */                                                       \
    X(float, 16, 3, 3, 9, 9, 3, 3)   /* ... pair protocol */                                                           \
    X(float, 16, 3, 2, 9, 8, 2, 1)   /* n = 2^17: 512 x 256, teams of 4 */                                             \
    X(float, 16, 3, 2, 9, 8, 2, 3)                                                                                     \
    X(float, 16, 2, 2, 8, 8, 1, 1)   /* n = 2^16: 256 x 256, teams of 2 */                                             \
    X(float, 16, 2, 1, 8, 7, 0, 1)   /* n = 2^15: 256 x 128, one CU per transform */                                   \
    X(double, 8, 3, 1, 8, 6, 0, 1)   /* fp64 n = 2^14: 256 x 64, one CU per transform */                               \
    X(double, 8, 3, 2, 8, 7, 1, 1)   /* fp64 n = 2^15: 256 x 128, teams of 2 */                                        \
    X(double, 8, 3, 3, 8, 8, 2, 1)   /* fp64 n = 2^16: 256 x 256, teams of 4 */
// Measured and NOT instantiated (round 4, profiles/r4_ab_fp64_quad_e16.txt): fp64 n = 2^17 ... 2^19 with 16 values per thread and chunk on
// 256-thread workgroups (one wave per SIMD, 512 registers per wave: <double, 16, 4, 3, 10, 9, 5, 2>, <double, 16, 3, 3, 9, 9, 4, 1 / 2>,
// <double, 16, 3, 2, 9, 8, 3, 1>).  Correct (parity-green on the device), but 41 - 73 registers still spill and one wave per SIMD hides no
// latency: 75 / 75 / 84 Gpoint/s against 88 - 91 / 97 - 101 / 103 - 107 of round 2's team_defer_kernel, which keeps those sizes.
#define FFT_QUAD_EXTERN(T, ...) extern template __global__ void team_quad_kernel<T, __VA_ARGS__>(TeamParams<T>);
FFT_QUAD_INSTANCES(FFT_QUAD_EXTERN)
#undef FFT_QUAD_EXTERN
#endif

}  // namespace fftk
