/*
 * fft_apps.c -- host-array front ends of the fused consumers (include/fft_apps.h): the reference's
 * applications/convolution.c:34-96 and applications/power_spectrum.c:58-86, 133-190 as one fused device plan each.
 */
#include <stdio.h>
#include <stdlib.h>

#include "../../include/fft_apps.h"
#include "../../include/fft_gpu.h"
#include "../../include/fft_hip.h"

static int ensure_gpu(void) {
    if (fft_gpu_get_backend() == FFT_GPU_HIP) return 0;
    return fft_gpu_init(FFT_GPU_AUTO);
}

/* plan, upload, execute, download: x (nx) [, y (nx)] -> out (out_bytes) */
static int run_fused(fft_gpu_fused_t kind, const complex_t* x, const complex_t* y, int nx, const complex_t* h, int nh, void* out, size_t out_bytes,
                     double sample_rate) {
    if (!x || nx <= 0 || !out || ensure_gpu() != 0) return -1;
    int rc = -1;
    fft_gpu_plan_t plan = fft_gpu_plan_fused_hip(kind, nx, nh, h, 1, FFT_PREC_F64);
    fft_gpu_memory_t dx = plan ? fft_gpu_alloc((size_t)nx) : NULL;
    fft_gpu_memory_t dy = (plan && y) ? fft_gpu_alloc((size_t)nx) : NULL;
    fft_gpu_memory_t dout = plan ? fft_gpu_alloc_bytes_hip(out_bytes) : NULL;
    if (plan && dx && dout && (!y || dy)) {
        fft_gpu_copy_h2d(dx, x, (size_t)nx);
        if (y) fft_gpu_copy_h2d(dy, y, (size_t)nx);
        if (fft_gpu_execute_fused_hip(plan, fft_gpu_memory_ptr(dx), y ? fft_gpu_memory_ptr(dy) : NULL, fft_gpu_memory_ptr(dout), sample_rate) == 0 &&
            fft_gpu_plan_sync(plan) == 0 && fft_gpu_copy_d2h_bytes_hip(out, dout, out_bytes) == 0)
            rc = 0;
    }
    fft_gpu_free(dx);
    fft_gpu_free(dy);
    fft_gpu_free(dout);
    fft_gpu_destroy_plan(plan);
    return rc;
}

int fft_convolution_gpu(const complex_t* x, int nx, const complex_t* h, int nh, complex_t* y) {
    if (!h || nh <= 0) return -1;
    return run_fused(FFT_GPU_FUSED_CONV_LINEAR, x, NULL, nx, h, nh, y, (size_t)(nx + nh - 1) * sizeof(complex_t), 1.0);
}

int circular_convolution_gpu(const complex_t* x, const complex_t* h, int n, complex_t* y) {
    if (!h) return -1;
    if (!is_power_of_two(n)) {
        fprintf(stderr, "Error: Size %d is not a power of two\n", n);
        return -1;
    }
    return run_fused(FFT_GPU_FUSED_CONV_CIRCULAR, x, NULL, n, h, n, y, (size_t)n * sizeof(complex_t), 1.0);
}

double* compute_periodogram_gpu(const complex_t* signal, int n, double sample_rate) {
    if (n <= 0 || !is_power_of_two(n)) {
        fprintf(stderr, "Error: Size %d is not a power of two\n", n);
        return NULL;
    }
    double* psd = (double*)malloc(((size_t)n / 2 + 1) * sizeof(double));
    if (!psd) return NULL;
    if (run_fused(FFT_GPU_FUSED_PSD, signal, NULL, n, NULL, 0, psd, ((size_t)n / 2 + 1) * sizeof(double), sample_rate) != 0) {
        free(psd);
        return NULL;
    }
    return psd;
}

complex_t* autocorrelation_fft_gpu(const complex_t* signal, int n) {
    complex_t* acf = n > 0 ? allocate_complex_array(n) : NULL;
    if (!acf) return NULL;
    if (run_fused(FFT_GPU_FUSED_AUTOCORR, signal, NULL, n, NULL, 0, acf, (size_t)n * sizeof(complex_t), 1.0) != 0) {
        free_complex_array(acf);
        return NULL;
    }
    return acf;
}

complex_t* cross_correlation_fft_gpu(const complex_t* x, const complex_t* y, int n) {
    complex_t* ccf = (n > 0 && y) ? allocate_complex_array(n) : NULL;
    if (!ccf) return NULL;
    if (run_fused(FFT_GPU_FUSED_XCORR, x, y, n, NULL, 0, ccf, (size_t)n * sizeof(complex_t), 1.0) != 0) {
        free_complex_array(ccf);
        return NULL;
    }
    return ccf;
}
