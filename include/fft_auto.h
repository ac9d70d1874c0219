/*
 * fft_auto.h -- planner / one-shot API in front of the MI355X HIP engine.
 *
 * Same signatures, flag bits and return conventions as the reference's
 * include/fft_auto.h:17-194.  In this build EVERY plan executes on the HIP
 * engine (the reference's CPU algorithms are not part of this library; a
 * maintainer integrating it keeps their algorithms/core and only gains the
 * ALGO_GPU_HIP route -- see INTEGRATION.md).  No device => NULL / -1, loudly.
 */
#ifndef FFT_AUTO_H
#define FFT_AUTO_H

#include "fft_common.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fft_plan* fft_plan_t;

/* planning flags: values as in the reference (fft_auto.h:17-29) */
typedef enum {
    FFT_ESTIMATE = 0,
    FFT_MEASURE = 1,
    FFT_PATIENT = 2,
    FFT_EXHAUSTIVE = 3,
    FFT_WISDOM_ONLY = 4,
    FFT_REAL_INPUT = 1 << 5,
    FFT_REAL_OUTPUT = 1 << 6,
    FFT_UNALIGNED = 1 << 7,
    FFT_CONSERVE_MEMORY = 1 << 8,
    FFT_PREFER_GPU = 1 << 9,
    FFT_THREADED = 1 << 10
} fft_flags_t;

/* hardware capability bits: values as in the reference (fft_auto.h:145-154) */
typedef enum {
    FFT_HW_CPU_SSE = 1 << 0,
    FFT_HW_CPU_AVX = 1 << 1,
    FFT_HW_CPU_AVX2 = 1 << 2,
    FFT_HW_CPU_AVX512 = 1 << 3,
    FFT_HW_CPU_NEON = 1 << 4,
    FFT_HW_GPU_CUDA = 1 << 5,
    FFT_HW_GPU_MPS = 1 << 6,
    FFT_HW_GPU_OPENCL = 1 << 7,
    FFT_HW_GPU_HIP = 1 << 8 /* additive */
} fft_hardware_t;

/* reference fft_auto.h:43-85 */
fft_plan_t fft_plan_dft_1d(int n, complex_t* in, complex_t* out, int sign, unsigned flags);
void fft_execute(fft_plan_t plan);
void fft_execute_dft(fft_plan_t plan, complex_t* in, complex_t* out);
void fft_destroy_plan(fft_plan_t plan);
int fft_auto(complex_t* in, complex_t* out, int n, int sign);

/* reference fft_auto.h:88-124 (NULL / a use-after-free there, fft_auto.c:391-415; real here).  r2c: n reals ->
 * n/2 + 1 bins; c2r: n/2 + 1 bins -> n reals scaled by 1/n; 2D: row-major rows x cols, inverse scaled once by
 * 1/(rows*cols).  All executed with fft_execute(). */
fft_plan_t fft_plan_r2c_1d(int n, double* in, complex_t* out, unsigned flags);
fft_plan_t fft_plan_c2r_1d(int n, complex_t* in, double* out, unsigned flags);
fft_plan_t fft_plan_dft_2d(int rows, int cols, complex_t* in, complex_t* out, int sign, unsigned flags);
/* additive: which schedule FFT_MEASURE / FFT_PATIENT / FFT_EXHAUSTIVE kept for a plan (fft_gpu_algo_t of fft_hip.h; -1 =
 * the plan was not measured), and a way to drop the plans fft_auto() keeps between calls */
int fft_plan_measured_algo(fft_plan_t plan);
/* additive: fft_execute() / fft_execute_dft() are void (reference fft_auto.h:63-70); the status of a plan's last execute is
 * kept here: 0 done, 1 the copy to the device failed, 2 the transform failed, 3 the copy back failed (the output array was not
 * written; a message went to stderr) */
int fft_plan_last_error(fft_plan_t plan);
void fft_auto_cleanup(void);
char* fft_export_wisdom_to_string(void);
int fft_import_wisdom_from_string(const char* wisdom);

unsigned fft_get_hardware_capabilities(void);
void fft_plan_with_nthreads(int nthreads);

/* 64-byte aligned host allocations (reference fft_auto.c:352-383) */
complex_t* fft_alloc_complex(size_t n);
double* fft_alloc_real(size_t n);
void fft_free(void* p);

const char* fft_version(void);

#ifdef __cplusplus
}
#endif

#endif /* FFT_AUTO_H */
