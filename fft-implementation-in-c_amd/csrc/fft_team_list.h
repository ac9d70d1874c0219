// fft_team_list.h -- the device instantiations of team_fft_kernel (fft_team.h): geometry baked in per
// (precision, tiles per workgroup, elements per thread).  MI355X: 8 XCDs x 32 CUs, 64 KiB tiles.
//   fp32: n = 2^20 = 1024 x 1024 (NT 4), 2^19 = 512 x 1024 (NT 2), 2^18 = 512 x 512 (NT 1);
//         E = 16 (512 threads, radix-16 stages) or E = 8 (1024 threads, radix-8 stages)
//   fp64: n = 2^19 =  512 x 1024 (NT 4), 2^18 = 512 x  512 (NT 2), 2^17 = 256 x 512 (NT 1); E = 8
#pragma once
#include "fft_team.h"

namespace fftk {
template <typename T, int NT>
struct TeamGeo {
    static constexpr int value = 0;
};
template <> struct TeamGeo<float, 4> { static constexpr int value = FFT_TEAM_GEO(10, 10, 3, 3, 5); };
template <> struct TeamGeo<float, 2> { static constexpr int value = FFT_TEAM_GEO(9, 10, 4, 3, 5); };
template <> struct TeamGeo<float, 1> { static constexpr int value = FFT_TEAM_GEO(9, 9, 4, 4, 5); };
template <> struct TeamGeo<double, 4> { static constexpr int value = FFT_TEAM_GEO(9, 10, 3, 2, 5); };
template <> struct TeamGeo<double, 2> { static constexpr int value = FFT_TEAM_GEO(9, 9, 3, 3, 5); };
template <> struct TeamGeo<double, 1> { static constexpr int value = FFT_TEAM_GEO(8, 9, 4, 3, 5); };
}  // namespace fftk
