/*
 * fft_apps.h -- the reference's FFT consumers, device-backed (additive; the reference keeps these functions inside
 * its applications/ demo programs, they are not part of its library).
 *
 * Same names, argument order and results as
 *   fft_convolution, circular_convolution      applications/convolution.c:34-96
 *   compute_periodogram, autocorrelation_fft,
 *   cross_correlation_fft                      applications/power_spectrum.c:58-86, 133-190
 * with a `_gpu` suffix, host arrays in and out, and int 0 / -1 (or NULL) instead of exit().  Each is ONE fused plan on the
 * device (fft_hip.h: fft_gpu_plan_fused_hip): forward transform, element-wise step and inverse transform, with the zero
 * padding, the spectral product and the truncation riding on the FFT passes.  Batched, device-resident use goes through
 * fft_gpu_plan_fused_hip / fft_gpu_execute_fused_hip directly.
 */
#ifndef FFT_APPS_H
#define FFT_APPS_H

#include "fft_common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* y[0 .. nx + nh - 2] = x * h (linear convolution); y has room for nx + nh - 1 values */
int fft_convolution_gpu(const complex_t* x, int nx, const complex_t* h, int nh, complex_t* y);
/* y = x (*) h, all of length n (a power of two, as in the reference) */
int circular_convolution_gpu(const complex_t* x, const complex_t* h, int n, complex_t* y);
/* one-sided Hann periodogram: a malloc'd array of n/2 + 1 doubles (n a power of two), freed by the caller; NULL on failure */
double* compute_periodogram_gpu(const complex_t* signal, int n, double sample_rate);
/* first n lags; a malloc'd (allocate_complex_array) array freed by the caller; NULL on failure */
complex_t* autocorrelation_fft_gpu(const complex_t* signal, int n);
complex_t* cross_correlation_fft_gpu(const complex_t* x, const complex_t* y, int n);

#ifdef __cplusplus
}
#endif

#endif /* FFT_APPS_H */
