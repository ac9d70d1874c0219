// fft_rows_o2.hip -- explicit instantiation of the single-pass (rows in, rows out) tile kernels; built with -O2
// (see fft_rows_list.h).  The backend TU declares the same list `extern template`.
#include "fft_kernels.h"
#include "fft_rows_list.h"

namespace fftk {
#define FFT_INSTANTIATE(T, E, FAM) \
    template __global__ void tile_fft_kernel<T, E, 1, FAM, LOAD_LCONTIG, STORE_LCONTIG, false, 0>(TileParams<T>);
FFT_ROWS_LIST(FFT_INSTANTIATE)
#undef FFT_INSTANTIATE
#define FFT_INSTANTIATE_FIXED(T, LOG2L, LOG2C) \
    template __global__ void tile_fft_kernel<T, 4, 1, FAM_R4, LOAD_LCONTIG, STORE_LCONTIG, false, ((LOG2L) << 8) | (LOG2C)>(TileParams<T>);
FFT_ROWS_FIXED_LIST(FFT_INSTANTIATE_FIXED)
#undef FFT_INSTANTIATE_FIXED
#define FFT_INSTANTIATE_FIXED8(T, LOG2L, LOG2C) \
    template __global__ void tile_fft_kernel<T, 8, 1, FAM_SR16, LOAD_LCONTIG, STORE_LCONTIG, false, ((LOG2L) << 8) | (LOG2C)>(TileParams<T>);
FFT_ROWS_FIXED8_LIST(FFT_INSTANTIATE_FIXED8)
#undef FFT_INSTANTIATE_FIXED8
}  // namespace fftk
