/*
 * fft_algorithms.h -- per-algorithm entry points, device-backed.
 *
 * The reference declares one in-place routine per algorithm
 * (include/fft_algorithms.h:12-20: void f(complex_t* x, int n, fft_direction)).
 * Here the same names with a `_gpu` suffix run the corresponding butterfly
 * family of the HIP engine on a HOST array (H2D, execute, D2H); they return
 * 0 / -1 instead of calling exit() (the reference exits from
 * CHECK_POWER_OF_TWO, fft_common.h:123-127).  All compute the same DFT.
 */
#ifndef FFT_ALGORITHMS_H
#define FFT_ALGORITHMS_H

#include "fft_common.h"

#ifdef __cplusplus
extern "C" {
#endif

int radix2_dit_fft_gpu(complex_t* x, int n, fft_direction dir);  /* bit-reversal kernel + radix-2 DIT stage kernels */
int radix2_fft_gpu(complex_t* x, int n, fft_direction dir);      /* LDS Stockham radix-2 */
int radix4_fft_gpu(complex_t* x, int n, fft_direction dir);      /* LDS Stockham radix-4 */
int split_radix_fft_gpu(complex_t* x, int n, fft_direction dir); /* LDS Stockham, split-radix codelets */
int bluestein_fft_gpu(complex_t* x, int n, fft_direction dir);   /* chirp-z over the power-of-two engine, any n */

#ifdef __cplusplus
}
#endif

#endif /* FFT_ALGORITHMS_H */
