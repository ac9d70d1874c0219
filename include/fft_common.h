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
 * Additive: complex32_t (interleaved float re,im) for the fp32 entry points.
 */
#ifndef FFT_COMMON_H
#define FFT_COMMON_H

#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

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
#endif

#endif /* FFT_COMMON_H */
