// fft_team_list.h -- the device instantiations of team_fft_kernel (fft_team.h): four tiles per workgroup, sixteen
// (fp32) / eight (fp64) elements per thread, geometry baked in per (precision, log2 n).  MI355X: 8 XCDs x 32 CUs,
// 512-thread workgroups, 64 KiB tiles; a team is 2^(log2 n - 15) (fp32) / 2^(log2 n - 14) (fp64) CUs of one XCD.
#pragma once
#include "fft_team.h"

namespace fftk {
// FFT_TEAM_GEO(log2 L1, log2 L2, log2 CA, log2 CB, log2 TS); 0 = not built
template <typename T, int LOG2N>
struct TeamGeo {
    static constexpr int value = 0;
};
template <> struct TeamGeo<float, 20> { static constexpr int value = FFT_TEAM_GEO(10, 10, 3, 3, 5); };  // 1024 x 1024, whole XCD
template <> struct TeamGeo<float, 19> { static constexpr int value = FFT_TEAM_GEO(9, 10, 4, 3, 4); };   //  512 x 1024, 16 CUs
template <> struct TeamGeo<float, 18> { static constexpr int value = FFT_TEAM_GEO(9, 9, 4, 4, 3); };    //  512 x  512,  8 CUs
template <> struct TeamGeo<float, 17> { static constexpr int value = FFT_TEAM_GEO(8, 9, 5, 4, 2); };    //  256 x  512,  4 CUs
template <> struct TeamGeo<float, 16> { static constexpr int value = FFT_TEAM_GEO(8, 8, 5, 5, 1); };    //  256 x  256,  2 CUs
template <> struct TeamGeo<double, 19> { static constexpr int value = FFT_TEAM_GEO(9, 10, 3, 2, 5); };
template <> struct TeamGeo<double, 18> { static constexpr int value = FFT_TEAM_GEO(9, 9, 3, 3, 4); };
template <> struct TeamGeo<double, 17> { static constexpr int value = FFT_TEAM_GEO(8, 9, 4, 3, 3); };
template <> struct TeamGeo<double, 16> { static constexpr int value = FFT_TEAM_GEO(8, 8, 4, 4, 2); };
template <> struct TeamGeo<double, 15> { static constexpr int value = FFT_TEAM_GEO(7, 8, 5, 4, 1); };
// sizes whose ASPLIT variant (128-byte column segments) is instantiated too
template <typename T, int LOG2N>
struct TeamAsplitBuilt {
    static constexpr bool value = false;
};
template <> struct TeamAsplitBuilt<float, 20> { static constexpr bool value = true; };
// sizes whose PAIR variant of team_defer_kernel (paired row tiles, 128-byte result segments) is instantiated: the fp32
// geometries with CB = 8 rows per row tile
template <typename T, int LOG2N>
struct TeamPairBuilt {
    static constexpr bool value = false;
};
template <> struct TeamPairBuilt<float, 20> { static constexpr bool value = true; };
template <> struct TeamPairBuilt<float, 19> { static constexpr bool value = true; };
}  // namespace fftk
