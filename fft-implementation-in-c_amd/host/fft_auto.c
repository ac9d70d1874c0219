/*
 * fft_auto.c -- FFTW-style planner / one-shot API in front of the HIP engine.
 *
 * Same public behaviour as the reference's algorithms/auto/fft_auto.c for the
 * GPU route (:175-238 plan, :241-284 execute, :287-302 execute_dft, :305-322
 * destroy, :325-333 fft_auto): the plan borrows the host in/out arrays, owns
 * one device plan and one or two device buffers (gpu_out == gpu_in when the
 * transform is in place, :227), and fft_execute does H2D -> execute -> D2H
 * (:278-280).  Differences, all deliberate:
 *   - every plan runs on the HIP engine; this library contains no CPU FFT, so
 *     there is nothing to fall back to (no device => NULL, with a message);
 *   - the dead O(n) twiddle / bit-reversal precompute (:199-212) is gone;
 *   - fft_execute_dft does not mutate the plan (the reference swaps plan->in/out
 *     and is not re-entrant, :291-301).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fft_algorithms.h"
#include "../../include/fft_auto.h"
#include "../../include/fft_gpu.h"

#ifndef FFT_VERSION
#define FFT_VERSION "2.0.0-mi355x"
#endif

struct fft_plan {
    int n;
    complex_t* in;  /* borrowed */
    complex_t* out; /* borrowed */
    fft_direction dir;
    unsigned flags;
    fft_gpu_plan_t gpu_plan; /* owned */
    fft_gpu_memory_t gpu_in; /* owned */
    fft_gpu_memory_t gpu_out; /* owned unless == gpu_in */
};

static int g_num_threads = 0;

static int ensure_gpu(void) {
    if (fft_gpu_get_backend() == FFT_GPU_HIP) return 0;
    return fft_gpu_init(FFT_GPU_AUTO);
}

fft_plan_t fft_plan_dft_1d(int n, complex_t* in, complex_t* out, int sign, unsigned flags) {
    if (n <= 0 || !in || !out) return NULL;
    if (ensure_gpu() != 0) {
        fprintf(stderr, "fft_plan_dft_1d: no MI355X/HIP device -- this build has no CPU path\n");
        return NULL;
    }
    fft_plan_t plan = (fft_plan_t)calloc(1, sizeof(struct fft_plan));
    if (!plan) return NULL;
    plan->n = n;
    plan->in = in;
    plan->out = out;
    plan->dir = (sign < 0) ? FFT_FORWARD : FFT_INVERSE;
    plan->flags = flags;
    plan->gpu_plan = fft_gpu_plan_1d(n, 1, plan->dir);
    plan->gpu_in = fft_gpu_alloc((size_t)n);
    plan->gpu_out = (in == out) ? plan->gpu_in : fft_gpu_alloc((size_t)n);
    if (!plan->gpu_plan || !plan->gpu_in || !plan->gpu_out) {
        fft_destroy_plan(plan);
        return NULL;
    }
    return plan;
}

static void run_plan(fft_plan_t plan, complex_t* in, complex_t* out) {
    fft_gpu_copy_h2d(plan->gpu_in, in, (size_t)plan->n);
    /* a plan made for distinct arrays can still be executed in place and vice versa */
    fft_gpu_execute(plan->gpu_plan, plan->gpu_in, plan->gpu_out);
    fft_gpu_copy_d2h(out, plan->gpu_out, (size_t)plan->n);
}

void fft_execute(fft_plan_t plan) {
    if (!plan) return;
    run_plan(plan, plan->in, plan->out);
}

void fft_execute_dft(fft_plan_t plan, complex_t* in, complex_t* out) {
    if (!plan || !in || !out) return;
    run_plan(plan, in, out);
}

void fft_destroy_plan(fft_plan_t plan) {
    if (!plan) return;
    fft_gpu_destroy_plan(plan->gpu_plan);
    if (plan->gpu_out && plan->gpu_out != plan->gpu_in) fft_gpu_free(plan->gpu_out);
    fft_gpu_free(plan->gpu_in);
    free(plan);
}

int fft_auto(complex_t* in, complex_t* out, int n, int sign) {
    fft_plan_t plan = fft_plan_dft_1d(n, in, out, sign, FFT_ESTIMATE | FFT_PREFER_GPU);
    if (!plan) return -1;
    fft_execute(plan);
    fft_destroy_plan(plan);
    return 0;
}

unsigned fft_get_hardware_capabilities(void) {
    unsigned caps = 0;
#if defined(__x86_64__) || defined(__i386__)
    __builtin_cpu_init();
    if (__builtin_cpu_supports("sse2")) caps |= FFT_HW_CPU_SSE;
    if (__builtin_cpu_supports("avx")) caps |= FFT_HW_CPU_AVX;
    if (__builtin_cpu_supports("avx2")) caps |= FFT_HW_CPU_AVX2;
    if (__builtin_cpu_supports("avx512f")) caps |= FFT_HW_CPU_AVX512;
#endif
    if (fft_gpu_available()) caps |= FFT_HW_GPU_HIP;
    return caps;
}

void fft_plan_with_nthreads(int nthreads) { g_num_threads = nthreads > 0 ? nthreads : 0; }

complex_t* fft_alloc_complex(size_t n) {
    void* p = NULL;
    if (posix_memalign(&p, 64, (n ? n : 1) * sizeof(complex_t)) != 0) return NULL;
    return (complex_t*)p;
}

double* fft_alloc_real(size_t n) {
    void* p = NULL;
    if (posix_memalign(&p, 64, (n ? n : 1) * sizeof(double)) != 0) return NULL;
    return (double*)p;
}

void fft_free(void* p) { free(p); }

const char* fft_version(void) { return FFT_VERSION; }

/* ---- stubs kept as stubs (reference fft_auto.c:391-426); out of scope for this path ---- */
fft_plan_t fft_plan_r2c_1d(int n, double* in, complex_t* out, unsigned flags) {
    (void)n; (void)in; (void)out; (void)flags;
    return NULL;
}
fft_plan_t fft_plan_c2r_1d(int n, complex_t* in, double* out, unsigned flags) {
    (void)n; (void)in; (void)out; (void)flags;
    return NULL;
}
fft_plan_t fft_plan_dft_2d(int rows, int cols, complex_t* in, complex_t* out, int sign, unsigned flags) {
    (void)rows; (void)cols; (void)in; (void)out; (void)sign; (void)flags;
    return NULL;
}
char* fft_export_wisdom_to_string(void) { return strdup("# FFT Wisdom v2.0.0 (mi355x: plans are deterministic, nothing to save)\n"); }
int fft_import_wisdom_from_string(const char* wisdom) { return wisdom ? 1 : 0; }

/* ---- per-algorithm host-array entry points (include/fft_algorithms.h) ---- */
static int run_algo(complex_t* x, int n, fft_direction dir, fft_gpu_algo_t algo, int need_pow2) {
    if (!x || n <= 0) {
        fprintf(stderr, "Error: invalid input (NULL array or n <= 0)\n");
        return -1;
    }
    if (need_pow2 && !is_power_of_two(n)) {
        fprintf(stderr, "Error: Size %d is not a power of two\n", n);
        return -1;
    }
    if (ensure_gpu() != 0) return -1;
    fft_gpu_plan_t plan = fft_gpu_plan_1d_ex(n, 1, dir, FFT_PREC_F64, algo);
    fft_gpu_memory_t buf = fft_gpu_alloc((size_t)n);
    int rc = -1;
    if (plan && buf) {
        fft_gpu_copy_h2d(buf, x, (size_t)n);
        fft_gpu_execute(plan, buf, buf);
        fft_gpu_copy_d2h(x, buf, (size_t)n);
        rc = 0;
    }
    fft_gpu_free(buf);
    fft_gpu_destroy_plan(plan);
    return rc;
}

int radix2_dit_fft_gpu(complex_t* x, int n, fft_direction dir) { return run_algo(x, n, dir, FFT_GPU_ALGO_RADIX2_GLOBAL, 1); }
int radix2_fft_gpu(complex_t* x, int n, fft_direction dir) { return run_algo(x, n, dir, FFT_GPU_ALGO_RADIX2, 1); }
int radix4_fft_gpu(complex_t* x, int n, fft_direction dir) { return run_algo(x, n, dir, FFT_GPU_ALGO_RADIX4, 1); }
int split_radix_fft_gpu(complex_t* x, int n, fft_direction dir) { return run_algo(x, n, dir, FFT_GPU_ALGO_SPLIT_RADIX, 1); }
int bluestein_fft_gpu(complex_t* x, int n, fft_direction dir) {
    /* any n: a power of two is padded like any other length (m = next_pow2(2n-1), bluestein.c:87) */
    return run_algo(x, n, dir, FFT_GPU_ALGO_BLUESTEIN, 0);
}
