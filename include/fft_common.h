/*
 * fft_common.h -- shared types for the MI355X FFT engine.
 *
 * API-compatible with the reference's include/fft_common.h (types, enum
 * values and helper names a caller of the FFT path touches), written from
 * scratch:
 *   complex_t            reference include/fft_common.h:28   (interleaved double re,im)
 *   fft_direction        reference include/fft_common.h:31-34 (FORWARD=-1, INVERSE=+1)
 *   is_power_of_two / next_power_of_two / log2_int   :37-56
 *   bit_reverse          :59-77  -- here correct for EVERY log2n (the reference's
 *                                   fast path returns 0 for log2n <= 4)
 *   twiddle_factor       :89-98
 *   fft_timer_t, timer_start/stop                    :101-114 (CPU-time stopwatch of the demos and benchmarks)
 *   CHECK_NULL, CHECK_POWER_OF_TWO                   :117-127 (caller-side guards; they terminate the CALLER's program
 *                                   exactly as the reference's do -- the library itself never exits; the _RET forms
 *                                   below return instead)
 *   print_complex(_array), generate_sine_wave / square_wave / impulse   :130-164
 *   compute_magnitude / phase / power_spectrum       :166-196 (malloc'd arrays owned by the caller)
 * so that the reference's own harnesses (examples/demo_v2_features.c, benchmarks/) build against this header.
 * Additive: complex32_t (interleaved float re,im) for the fp32 entry points.
 */
#ifndef FFT_COMMON_H
#define FFT_COMMON_H

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

#ifdef __cplusplus
/* C++ translation units (the HIP shim) see the same 16-byte / 8-byte layout. */
typedef struct { double re, im; } complex_t;
typedef struct { float re, im; } complex32_t;
#else
#include <complex.h>
typedef double complex complex_t;
typedef float complex complex32_t;
#endif

#ifndef PI
#define PI 3.14159265358979323846
#endif
#ifndef TWO_PI
#define TWO_PI (2.0 * PI)
#endif

typedef enum {
    FFT_FORWARD = -1, /* exponent exp(-2 pi i jk/N) */
    FFT_INVERSE = 1   /* exponent exp(+2 pi i jk/N), result scaled by 1/N */
} fft_direction;

static inline int is_power_of_two(int n) { return n > 0 && (n & (n - 1)) == 0; }

/* smallest power of two >= n (n itself when it already is one) */
static inline int next_power_of_two(int n) {
    unsigned v = (unsigned)n - 1u;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return (int)(v + 1u);
}

static inline int log2_int(int n) {
    int l = 0;
    while (n > 1) { n >>= 1; l++; }
    return l;
}

/* reverse the low log2n bits of x */
static inline unsigned int bit_reverse(unsigned int x, int log2n) {
    unsigned int v = x;
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    v = (v >> 16) | (v << 16);
    return log2n > 0 ? v >> (32 - log2n) : 0u;
}

#ifndef __cplusplus
static inline complex_t* allocate_complex_array(int n) {
    return (complex_t*)calloc((size_t)(n > 0 ? n : 0), sizeof(complex_t));
}
static inline void free_complex_array(complex_t* a) { free(a); }

/* W_n^k for the given direction, exact at the quarter points */
static inline complex_t twiddle_factor(int k, int n, fft_direction dir) {
    if (k == 0) return 1.0;
    if (2 * k == n) return -1.0;
    if (4 * k == n) return dir == FFT_FORWARD ? -I : I;
    if (4 * k == 3 * n) return dir == FFT_FORWARD ? I : -I;
    double a = (double)dir * TWO_PI * (double)k / (double)n;
    return cos(a) + I * sin(a);
}

/* ---- stopwatch on the process CPU clock (what the reference's demos time with) */
typedef struct {
    clock_t start;
    clock_t end;
    double elapsed_ms;
} fft_timer_t;

static inline void timer_start(fft_timer_t* t) { t->start = clock(); }

static inline void timer_stop(fft_timer_t* t) {
    t->end = clock();
    t->elapsed_ms = 1000.0 * (double)(t->end - t->start) / (double)CLOCKS_PER_SEC;
}

/* ---- caller-side guards.  Same names and behaviour as the reference's: a failed check ends the calling PROGRAM. */
#define CHECK_NULL(ptr, msg)                          \
    do {                                              \
        if (!(ptr)) {                                 \
            fprintf(stderr, "Error: %s\n", (msg));    \
            exit(EXIT_FAILURE);                       \
        }                                             \
    } while (0)

#define CHECK_POWER_OF_TWO(n)                                                   \
    do {                                                                        \
        if (!is_power_of_two(n)) {                                              \
            fprintf(stderr, "Error: Size %d is not a power of two\n", (int)(n)); \
            exit(EXIT_FAILURE);                                                 \
        }                                                                       \
    } while (0)

/* the same checks for code that must not exit (library style: report and return `ret`) */
#define FFT_CHECK_NULL_RET(ptr, msg, ret)             \
    do {                                              \
        if (!(ptr)) {                                 \
            fprintf(stderr, "Error: %s\n", (msg));    \
            return ret;                               \
        }                                             \
    } while (0)

#define FFT_CHECK_POWER_OF_TWO_RET(n, ret)                                      \
    do {                                                                        \
        if (!is_power_of_two(n)) {                                              \
            fprintf(stderr, "Error: Size %d is not a power of two\n", (int)(n)); \
            return ret;                                                         \
        }                                                                       \
    } while (0)

/* ---- printing, test signals, spectra */
static inline void print_complex(complex_t c) {
    double re = creal(c), im = cimag(c);
    if (fabs(re) < 1e-10) re = 0;  /* no "-0.000" */
    if (fabs(im) < 1e-10) im = 0;
    printf("(%.3f, %.3fi)", re, im);
}

static inline void print_complex_array(const char* label, complex_t* arr, int n) {
    printf("%s: ", label);
    for (int i = 0; i < n; i++) {
        print_complex(arr[i]);
        putchar(' ');
    }
    putchar('\n');
}

/* real sine of frequency `freq` sampled at `fs` */
static inline void generate_sine_wave(complex_t* signal, int n, double freq, double fs) {
    for (int i = 0; i < n; i++) signal[i] = sin(TWO_PI * freq * (double)i / fs);
}

/* +1 for the first half of every period of (int)(fs / freq) samples, -1 for the second */
static inline void generate_square_wave(complex_t* signal, int n, double freq, double fs) {
    const int period = (int)(fs / freq);
    for (int i = 0; i < n; i++) signal[i] = (period > 0 && (i % period) < period / 2) ? 1.0 : -1.0;
}

static inline void generate_impulse(complex_t* signal, int n) {
    for (int i = 0; i < n; i++) signal[i] = 0.0;
    if (n > 0) signal[0] = 1.0;
}

/* |X[k]|, arg X[k], |X[k]|^2 / n: a new malloc'd array of n doubles each, freed by the caller */
static inline double* compute_magnitude(complex_t* fft_result, int n) {
    double* mag = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    CHECK_NULL(mag, "Failed to allocate magnitude array");
    for (int i = 0; i < n; i++) mag[i] = cabs(fft_result[i]);
    return mag;
}

static inline double* compute_phase(complex_t* fft_result, int n) {
    double* phase = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    CHECK_NULL(phase, "Failed to allocate phase array");
    for (int i = 0; i < n; i++) phase[i] = carg(fft_result[i]);
    return phase;
}

static inline double* compute_power_spectrum(complex_t* fft_result, int n) {
    double* power = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    CHECK_NULL(power, "Failed to allocate power spectrum array");
    for (int i = 0; i < n; i++) {
        const double m = cabs(fft_result[i]);
        power[i] = m * m / (double)n;
    }
    return power;
}
#endif

#endif /* FFT_COMMON_H */
