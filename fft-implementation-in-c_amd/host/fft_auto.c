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
 *     and is not re-entrant, :291-301);
 *   - FFT_MEASURE and above really measure (the reference's TODO, :232-235): the candidate schedules for this n
 *     are timed on the device at plan time and the fastest is kept;
 *   - the borrowed host arrays are page-locked for the plan's lifetime (hipHostRegister), so every execute's
 *     H2D / D2H runs at pinned-memory speed; FFT_CONSERVE_MEMORY skips that;
 *   - fft_auto() keeps its plans (the reference builds and destroys a plan per call, :325-333): a small cache
 *     keyed by (n, direction) holds the device plan, its stream and its device buffer, so a repeated call costs
 *     H2D + execute + D2H and nothing else (no hipMalloc, no stream, no table upload);
 *   - fft_plan_r2c_1d / fft_plan_c2r_1d / fft_plan_dft_2d are real (NULL / a use-after-free in the reference,
 *     :391-415): n/2 + 1 bins, c2r scaled by 1/n, the 2D inverse scaled once by 1/(rows*cols).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fft_algorithms.h"
#include "../../include/fft_auto.h"
#include "../../include/fft_gpu.h"
#include "../../include/fft_hip.h"

#ifndef FFT_VERSION
#define FFT_VERSION "2.0.0-mi355x"
#endif

enum { KIND_C2C = 0, KIND_R2C = 1, KIND_C2R = 2, KIND_2D = 3 };

struct fft_plan {
    int kind;
    int n;          /* transform length (2D: rows * cols) */
    int rows, cols; /* 2D */
    void* in;       /* borrowed: complex_t* or double* by kind */
    void* out;      /* borrowed */
    size_t in_bytes, out_bytes;
    fft_direction dir;
    unsigned flags;
    fft_gpu_plan_t gpu_plan; /* owned */
    fft_gpu_memory_t gpu_in; /* owned */
    fft_gpu_memory_t gpu_out; /* owned unless == gpu_in */
    int pinned_in, pinned_out; /* the borrowed arrays are page-locked by this plan */
    int measured_algo;         /* FFT_MEASURE: the schedule that won (fft_gpu_algo_t), -1 = not measured */
    int last_error;            /* run_plan: 0, or the step of the last execute that failed */
};

static int g_num_threads = 0;

static int ensure_gpu(void) {
    if (fft_gpu_get_backend() == FFT_GPU_HIP) return 0;
    return fft_gpu_init(FFT_GPU_AUTO);
}

/* FFT_MEASURE (reference TODO fft_auto.c:232-235): time the candidate schedules for a single transform of length n on the
 * device -- AUTO's pick and the explicit radix-4 / radix-2 families for a power of two -- and keep the fastest plan. */
static fft_gpu_plan_t measured_plan(int n, fft_direction dir, fft_gpu_memory_t buf, int* winner) {
    static const fft_gpu_algo_t candidates[3] = {FFT_GPU_ALGO_AUTO, FFT_GPU_ALGO_RADIX4, FFT_GPU_ALGO_RADIX2};
    const int n_cand = is_power_of_two(n) && n >= 4 ? 3 : 1;
    fft_gpu_plan_t best = NULL;
    float best_ms = 0.f;
    for (int c = 0; c < n_cand; c++) {
        fft_gpu_plan_t p = fft_gpu_plan_1d_ex(n, 1, dir, FFT_PREC_F64, candidates[c]);
        if (!p) continue;
        float ms = 0.f;
        void* d = fft_gpu_memory_ptr(buf);
        /* one untimed execute (first-launch costs), then 4 timed ones between HIP events on the plan's stream */
        if (fft_gpu_execute_timed(p, d, d, 1, &ms) != 0 || fft_gpu_execute_timed(p, d, d, 4, &ms) != 0) {
            fft_gpu_destroy_plan(p);
            continue;
        }
        if (!best || ms < best_ms) {
            if (best) fft_gpu_destroy_plan(best);
            best = p;
            best_ms = ms;
            *winner = (int)candidates[c];
        } else {
            fft_gpu_destroy_plan(p);
        }
    }
    return best;
}

static fft_plan_t new_plan(int kind, int n, void* in, void* out, size_t in_bytes, size_t out_bytes, fft_direction dir, unsigned flags) {
    fft_plan_t plan = (fft_plan_t)calloc(1, sizeof(struct fft_plan));
    if (!plan) return NULL;
    plan->kind = kind;
    plan->n = n;
    plan->in = in;
    plan->out = out;
    plan->in_bytes = in_bytes;
    plan->out_bytes = out_bytes;
    plan->dir = dir;
    plan->flags = flags;
    plan->measured_algo = -1;
    return plan;
}

/* device buffers (one when the transform is in place on equally sized arrays) and page-locking of the borrowed arrays */
static int plan_buffers(fft_plan_t plan) {
    plan->gpu_in = fft_gpu_alloc_bytes_hip(plan->in_bytes);
    plan->gpu_out = (plan->in == plan->out && plan->in_bytes == plan->out_bytes) ? plan->gpu_in : fft_gpu_alloc_bytes_hip(plan->out_bytes);
    if (!plan->gpu_in || !plan->gpu_out) return -1;
    if (!(plan->flags & FFT_CONSERVE_MEMORY)) {
        /* in == out (an in-place r2c writes n/2 + 1 complex values over n reals): ONE page-locked range that covers both uses.
         * A range the runtime already knows (another plan on the same array) is left alone -- the copies then go the pageable
         * way, through a bounce buffer where the known range is shorter than this plan's (fft_gpu_copy_*_bytes_hip). */
        const size_t in_reg = (plan->out == plan->in && plan->out_bytes > plan->in_bytes) ? plan->out_bytes : plan->in_bytes;
        if (!fft_gpu_host_is_registered_hip(plan->in)) plan->pinned_in = fft_gpu_host_register_hip(plan->in, in_reg) == 0;
        if (plan->out != plan->in && !fft_gpu_host_is_registered_hip(plan->out))
            plan->pinned_out = fft_gpu_host_register_hip(plan->out, plan->out_bytes) == 0;
    }
    return 0;
}

fft_plan_t fft_plan_dft_1d(int n, complex_t* in, complex_t* out, int sign, unsigned flags) {
    if (n <= 0 || !in || !out) return NULL;
    if (ensure_gpu() != 0) {
        fprintf(stderr, "fft_plan_dft_1d: no MI355X/HIP device -- this build has no CPU path\n");
        return NULL;
    }
    const size_t bytes = (size_t)n * sizeof(complex_t);
    fft_plan_t plan = new_plan(KIND_C2C, n, in, out, bytes, bytes, (sign < 0) ? FFT_FORWARD : FFT_INVERSE, flags);
    if (!plan) return NULL;
    if (plan_buffers(plan) != 0) {
        fft_destroy_plan(plan);
        return NULL;
    }
    const unsigned effort = flags & 7u; /* FFT_ESTIMATE 0 .. FFT_WISDOM_ONLY 4 (fft_auto.h:17-22) */
    if (effort >= FFT_MEASURE && effort <= FFT_EXHAUSTIVE) plan->gpu_plan = measured_plan(n, plan->dir, plan->gpu_in, &plan->measured_algo);
    if (!plan->gpu_plan) plan->gpu_plan = fft_gpu_plan_1d(n, 1, plan->dir);
    if (!plan->gpu_plan) {
        fft_destroy_plan(plan);
        return NULL;
    }
    return plan;
}

/* which schedule FFT_MEASURE kept (fft_gpu_algo_t), -1 when the plan was not measured (additive) */
int fft_plan_measured_algo(fft_plan_t plan) { return plan ? plan->measured_algo : -1; }

/* 0, or which step failed (1 copy in, 2 transform, 3 copy out): fft_execute() is void in the reference's API, so the status
 * is kept on the plan (fft_plan_last_error) and reported on stderr -- never silently dropped */
static int run_plan(fft_plan_t plan, const void* in, void* out) {
    plan->last_error = 0;
    if (fft_gpu_copy_h2d_bytes_hip(plan->gpu_in, in, plan->in_bytes) != 0) plan->last_error = 1;
    /* a plan made for distinct arrays can still be executed in place and vice versa */
    if (!plan->last_error) {
        fft_gpu_execute(plan->gpu_plan, plan->gpu_in, plan->gpu_out);
        if (fft_gpu_plan_sync_hip(plan->gpu_plan) != 0) plan->last_error = 2;
    }
    if (!plan->last_error && fft_gpu_copy_d2h_bytes_hip(out, plan->gpu_out, plan->out_bytes) != 0) plan->last_error = 3;
    if (plan->last_error)
        fprintf(stderr, "fft_execute: step %d of copy-in / transform / copy-out failed; the output array was not written\n", plan->last_error);
    return plan->last_error;
}

/* status of the plan's last execute: 0 done, 1 / 2 / 3 the copy in / the transform / the copy out failed (additive) */
int fft_plan_last_error(fft_plan_t plan) { return plan ? plan->last_error : -1; }

void fft_execute(fft_plan_t plan) {
    if (!plan) return;
    (void)run_plan(plan, plan->in, plan->out);
}

void fft_execute_dft(fft_plan_t plan, complex_t* in, complex_t* out) {
    if (!plan || !in || !out) return;
    if (plan->kind != KIND_C2C && plan->kind != KIND_2D) {
        fprintf(stderr, "fft_execute_dft: complex-to-complex plans only\n");
        return;
    }
    (void)run_plan(plan, in, out);
}

void fft_destroy_plan(fft_plan_t plan) {
    if (!plan) return;
    fft_gpu_destroy_plan(plan->gpu_plan);
    if (plan->gpu_out && plan->gpu_out != plan->gpu_in) fft_gpu_free(plan->gpu_out);
    fft_gpu_free(plan->gpu_in);
    if (plan->pinned_in) (void)fft_gpu_host_unregister_hip(plan->in);
    if (plan->pinned_out) (void)fft_gpu_host_unregister_hip(plan->out);
    free(plan);
}

/* ---- fft_auto(): one-shot transforms with kept plans ------------------------------------------------------------ */
enum { AUTO_CACHE_SLOTS = 8 };
struct auto_slot {
    int n;
    fft_direction dir;
    fft_gpu_plan_t plan;
    fft_gpu_memory_t buf;
    unsigned long last_use;
};
static struct auto_slot g_auto[AUTO_CACHE_SLOTS];
static unsigned long g_auto_clock = 0;
static pthread_mutex_t g_auto_lock = PTHREAD_MUTEX_INITIALIZER;

/* drop every kept plan (additive; also what a caller does before fft_gpu_cleanup()) */
void fft_auto_cleanup(void) {
    pthread_mutex_lock(&g_auto_lock);
    for (int i = 0; i < AUTO_CACHE_SLOTS; i++) {
        if (g_auto[i].plan) fft_gpu_destroy_plan(g_auto[i].plan);
        if (g_auto[i].buf) fft_gpu_free(g_auto[i].buf);
        memset(&g_auto[i], 0, sizeof(g_auto[i]));
    }
    pthread_mutex_unlock(&g_auto_lock);
}

int fft_auto(complex_t* in, complex_t* out, int n, int sign) {
    if (n <= 0 || !in || !out) return -1;
    if (ensure_gpu() != 0) {
        fprintf(stderr, "fft_auto: no MI355X/HIP device -- this build has no CPU path\n");
        return -1;
    }
    const fft_direction dir = (sign < 0) ? FFT_FORWARD : FFT_INVERSE;
    int rc = -1;
    pthread_mutex_lock(&g_auto_lock); /* the slot's buffer is in use until the D2H below: one fft_auto at a time */
    struct auto_slot* s = NULL;
    struct auto_slot* victim = &g_auto[0];
    for (int i = 0; i < AUTO_CACHE_SLOTS; i++) {
        if (g_auto[i].plan && g_auto[i].n == n && g_auto[i].dir == dir) s = &g_auto[i];
        if (!g_auto[i].plan || g_auto[i].last_use < victim->last_use) {
            if (!victim->plan && g_auto[i].plan) continue; /* an empty slot beats any used one */
            victim = &g_auto[i];
        }
    }
    if (!s) {
        if (victim->plan) fft_gpu_destroy_plan(victim->plan);
        if (victim->buf) fft_gpu_free(victim->buf);
        memset(victim, 0, sizeof(*victim));
        victim->plan = fft_gpu_plan_1d(n, 1, dir);
        victim->buf = fft_gpu_alloc((size_t)n);
        if (!victim->plan || !victim->buf) {
            if (victim->plan) fft_gpu_destroy_plan(victim->plan);
            if (victim->buf) fft_gpu_free(victim->buf);
            memset(victim, 0, sizeof(*victim));
            pthread_mutex_unlock(&g_auto_lock);
            return -1;
        }
        victim->n = n;
        victim->dir = dir;
        s = victim;
    }
    s->last_use = ++g_auto_clock;
    fft_gpu_copy_h2d(s->buf, in, (size_t)n);
    fft_gpu_execute(s->plan, s->buf, s->buf);
    if (fft_gpu_copy_d2h_bytes_hip(out, s->buf, (size_t)n * sizeof(complex_t)) == 0) rc = 0;
    pthread_mutex_unlock(&g_auto_lock);
    return rc;
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

/* ---- real-input, real-output and 2D plans (NULL / broken in the reference, fft_auto.c:391-415) ---- */
fft_plan_t fft_plan_r2c_1d(int n, double* in, complex_t* out, unsigned flags) {
    if (n <= 0 || !in || !out || ensure_gpu() != 0) return NULL;
    fft_plan_t plan = new_plan(KIND_R2C, n, in, out, (size_t)n * sizeof(double), ((size_t)n / 2 + 1) * sizeof(complex_t), FFT_FORWARD,
                               flags | FFT_REAL_INPUT);
    if (!plan) return NULL;
    plan->gpu_plan = fft_gpu_plan_r2c_1d_hip(n, 1, FFT_PREC_F64);
    if (!plan->gpu_plan || plan_buffers(plan) != 0) {
        fft_destroy_plan(plan);
        return NULL;
    }
    return plan;
}
fft_plan_t fft_plan_c2r_1d(int n, complex_t* in, double* out, unsigned flags) {
    if (n <= 0 || !in || !out || ensure_gpu() != 0) return NULL;
    fft_plan_t plan = new_plan(KIND_C2R, n, in, out, ((size_t)n / 2 + 1) * sizeof(complex_t), (size_t)n * sizeof(double), FFT_INVERSE,
                               flags | FFT_REAL_OUTPUT);
    if (!plan) return NULL;
    plan->gpu_plan = fft_gpu_plan_c2r_1d_hip(n, 1, FFT_PREC_F64);
    if (!plan->gpu_plan || plan_buffers(plan) != 0) {
        fft_destroy_plan(plan);
        return NULL;
    }
    return plan;
}
fft_plan_t fft_plan_dft_2d(int rows, int cols, complex_t* in, complex_t* out, int sign, unsigned flags) {
    if (rows <= 0 || cols <= 0 || (long long)rows * cols > (1ll << 30) || !in || !out || ensure_gpu() != 0) return NULL;
    const size_t bytes = (size_t)rows * (size_t)cols * sizeof(complex_t);
    fft_plan_t plan = new_plan(KIND_2D, rows * cols, in, out, bytes, bytes, (sign < 0) ? FFT_FORWARD : FFT_INVERSE, flags);
    if (!plan) return NULL;
    plan->rows = rows;
    plan->cols = cols;
    plan->gpu_plan = fft_gpu_plan_2d(rows, cols, plan->dir);
    if (!plan->gpu_plan || plan_buffers(plan) != 0) {
        fft_destroy_plan(plan);
        return NULL;
    }
    return plan;
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
