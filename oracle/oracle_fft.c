/*
 * oracle_fft.c -- CPU restatement of the reference's 1D complex FFT path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
 * calls this file.  It is the checker that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg compare the HIP path against.
 *
 * Parity status: PINNED.  Every function below that has a reference
 * counterpart is checked bit-for-bit (recurrence variants) against the real
 * reference compiled from /root/reference by oracle/Makefile (target `ref`,
 * output oracle/_ref/libref.so), see tests/test_oracle.py::test_against_real_reference_when_present, and against
 * the golden vectors under tests/golden/ that were generated from that build
 * (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off  (contraction OFF is required for
 * bit-exactness with the reference, which is C99 without -ffast-math).
 *
 * Layout: complex numbers are interleaved (re, im) -- identical to the
 * reference's `typedef double complex complex_t` (include/fft_common.h:28).
 * All routines work in place, like the reference (algorithms/core/ *.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_PI 3.14159265358979323846 /* include/fft_common.h:24 */
#define ORACLE_TWO_PI (2.0 * ORACLE_PI)  /* include/fft_common.h:25 */

/* ---- size helpers: include/fft_common.h:37-56 --------------------------- */
int oracle_is_power_of_two(int n) { return n > 0 && (n & (n - 1)) == 0; }

int oracle_next_power_of_two(int n) {
    n--;
    n |= n >> 1; n |= n >> 2; n |= n >> 4; n |= n >> 8; n |= n >> 16;
    return n + 1;
}

int oracle_log2_int(int n) {
    int l = 0;
    while (n >>= 1) l++;
    return l;
}

/*
 * Bit reversal of the low log2n bits.  Restates the GENERAL loop of
 * include/fft_common.h:70-76, which is the branch taken for log2n > 8 (all
 * BASELINE configs).  The reference's fast path for log2n <= 8 (:61-68) is
 * correct for log2n in 5..8 and returns garbage for log2n <= 4 (SURVEY.md
 * fact 3); this restatement is the mathematically correct permutation for
 * every log2n, which equals the reference wherever the reference is right.
 */
unsigned oracle_bit_reverse(unsigned x, int log2n) {
    unsigned r = 0;
    for (int i = 0; i < log2n; i++) {
        r = (r << 1) | (x & 1u);
        x >>= 1;
    }
    return r;
}

/* The reference's function verbatim in behaviour, bug included
 * (include/fft_common.h:59-77) -- only used to document/pin the bug. */
unsigned oracle_bit_reverse_asref(unsigned x, int log2n) {
    if (log2n <= 8) {
        x = ((x & 0xAAAA) >> 1) | ((x & 0x5555) << 1);
        x = ((x & 0xCCCC) >> 2) | ((x & 0x3333) << 2);
        x = ((x & 0xF0F0) >> 4) | ((x & 0x0F0F) << 4);
        if (log2n > 4) x = ((x & 0xFF00) >> 8) | ((x & 0x00FF) << 8);
        return x >> (16 - log2n);
    }
    return oracle_bit_reverse(x, log2n);
}

void oracle_bit_reverse_table(uint32_t* table, int log2n) {
    uint32_t n = 1u << log2n;
    for (uint32_t i = 0; i < n; i++) table[i] = oracle_bit_reverse(i, log2n);
}

/* ---- twiddle_factor(k, n, dir): include/fft_common.h:89-98 -------------- */
void oracle_twiddle_factor(int k, int n, int dir, double* re, double* im) {
    if (k == 0) { *re = 1.0; *im = 0.0; return; }
    if (k * 4 == n) { *re = 0.0; *im = (dir < 0) ? -1.0 : 1.0; return; }
    if (k * 2 == n) { *re = -1.0; *im = 0.0; return; }
    if (k * 4 == 3 * n) { *re = 0.0; *im = (dir < 0) ? 1.0 : -1.0; return; }
    double angle = dir * ORACLE_TWO_PI * k / n;
    *re = cos(angle); /* cexp(I*angle) == cos + i sin (real part of arg is 0) */
    *im = sin(angle);
}

static void permute_bitrev(double* x, int n, int log2n) {
    /* algorithms/core/radix2_dit.c:70-77 */
    for (int i = 0; i < n; i++) {
        int j = (int)oracle_bit_reverse((unsigned)i, log2n);
        if (i < j) {
            double tr = x[2 * i], ti = x[2 * i + 1];
            x[2 * i] = x[2 * j]; x[2 * i + 1] = x[2 * j + 1];
            x[2 * j] = tr; x[2 * j + 1] = ti;
        }
    }
}

/*
 * radix2_dit_fft: algorithms/core/radix2_dit.c:59-120.
 * Bit-reversal, log2 n in-place stages with the twiddle RECURRENCE
 * w *= w_m (:109), inverse scaled by a true division x[i] /= n (:115-119).
 * Complex products are written out exactly as __muldc3 evaluates them
 * (ac - bd, ad + bc), so the result is bit-identical to the reference built
 * without -ffast-math.  Returns -1 where the reference would exit(1) (:61).
 */
int oracle_radix2_dit(double* x, int n, int dir) {
    if (!oracle_is_power_of_two(n)) return -1;
    int log2n = oracle_log2_int(n);
    permute_bitrev(x, n, log2n);
    for (int stage = 1; stage <= log2n; stage++) {
        int m = 1 << stage, half = m >> 1;
        double mr, mi;
        oracle_twiddle_factor(1, m, dir, &mr, &mi);
        for (int k = 0; k < n; k += m) {
            double wr = 1.0, wi = 0.0;
            for (int j = 0; j < half; j++) {
                int t = k + j, u = t + half;
                double ur = x[2 * u], ui = x[2 * u + 1];
                double pr = ur * wr - ui * wi;
                double pi = ur * wi + ui * wr;
                double tr = x[2 * t], ti = x[2 * t + 1];
                x[2 * u] = tr - pr; x[2 * u + 1] = ti - pi;
                x[2 * t] = tr + pr; x[2 * t + 1] = ti + pi;
                double nr = wr * mr - wi * mi;
                double ni = wr * mi + wi * mr;
                wr = nr; wi = ni;
            }
        }
    }
    if (dir > 0) {
        double dn = (double)n;
        for (int i = 0; i < 2 * n; i++) x[i] /= dn;
    }
    return 0;
}

/*
 * radix4_fft (algorithms/core/radix4.c:83-134) and split_radix_fft
 * (algorithms/core/split_radix.c:23-70) are the same radix-2 pipeline in the
 * reference; radix4 scales the inverse by multiplying with 1.0/n (:128-133)
 * instead of dividing.  mode: 0 = split_radix (divide), 1 = radix4 (multiply).
 */
int oracle_radix4_or_split(double* x, int n, int dir, int mode) {
    if (!oracle_is_power_of_two(n)) return -1;
    if (n <= 1) return 0;
    int log2n = oracle_log2_int(n);
    permute_bitrev(x, n, log2n);
    for (int stage = 1; stage <= log2n; stage++) {
        int m = 1 << stage, half = m >> 1;
        double mr, mi;
        oracle_twiddle_factor(1, m, dir, &mr, &mi);
        for (int k = 0; k < n; k += m) {
            double wr = 1.0, wi = 0.0;
            for (int j = 0; j < half; j++) {
                int t = k + j, u = t + half;
                double ur = x[2 * u], ui = x[2 * u + 1];
                double pr = ur * wr - ui * wi;
                double pi = ur * wi + ui * wr;
                double tr = x[2 * t], ti = x[2 * t + 1];
                x[2 * t] = tr + pr; x[2 * t + 1] = ti + pi;
                x[2 * u] = tr - pr; x[2 * u + 1] = ti - pi;
                double nr = wr * mr - wi * mi;
                double ni = wr * mi + wi * mr;
                wr = nr; wi = ni;
            }
        }
    }
    if (dir > 0) {
        if (mode == 1) {
            double s = 1.0 / n;
            for (int i = 0; i < 2 * n; i++) x[i] *= s;
        } else {
            double dn = (double)n;
            for (int i = 0; i < 2 * n; i++) x[i] /= dn;
        }
    }
    return 0;
}

/*
 * radix2_dif_fft: algorithms/core/radix2_dif.c:15-59.  Gentleman-Sande:
 * stages from m = n down to 2 with x[u] = (a - b) * w, THEN bit-reversal,
 * then 1/n on the inverse.
 */
int oracle_radix2_dif(double* x, int n, int dir) {
    if (!oracle_is_power_of_two(n)) return -1;
    int log2n = oracle_log2_int(n);
    for (int stage = log2n; stage >= 1; stage--) {
        int m = 1 << stage, half = m >> 1;
        double mr, mi;
        oracle_twiddle_factor(1, m, dir, &mr, &mi);
        for (int k = 0; k < n; k += m) {
            double wr = 1.0, wi = 0.0;
            for (int j = 0; j < half; j++) {
                int t = k + j, u = t + half;
                double ar = x[2 * t], ai = x[2 * t + 1];
                double br = x[2 * u], bi = x[2 * u + 1];
                x[2 * t] = ar + br; x[2 * t + 1] = ai + bi;
                double dr = ar - br, di = ai - bi;
                x[2 * u] = dr * wr - di * wi;
                x[2 * u + 1] = dr * wi + di * wr;
                double nr = wr * mr - wi * mi;
                double ni = wr * mi + wi * mr;
                wr = nr; wi = ni;
            }
        }
    }
    permute_bitrev(x, n, log2n);
    if (dir > 0) {
        double dn = (double)n;
        for (int i = 0; i < 2 * n; i++) x[i] /= dn;
    }
    return 0;
}

/*
 * bluestein_fft: algorithms/core/bluestein.c:51-65 (chirp) and :79-155.
 * chirp[k] = exp(i * (-dir) * pi * k*k / n) with k*k formed in int then
 * promoted -- the reference writes `-dir * PI * k * k / n`, which evaluates
 * left to right in double: ((-dir * PI) * k) * k / n.  Three radix-2 FFTs of
 * length m = next_pow2(2n-1); the inverse one carries the 1/m.
 */
int oracle_bluestein(double* x, int n, int dir) {
    if (!x || n <= 0) return -1;
    int m = oracle_next_power_of_two(2 * n - 1);
    double* a = (double*)calloc((size_t)m * 2, sizeof(double));
    double* b = (double*)calloc((size_t)m * 2, sizeof(double));
    double* chirp = (double*)calloc((size_t)n * 2, sizeof(double));
    if (!a || !b || !chirp) { free(a); free(b); free(chirp); return -1; }
    for (int k = 0; k < n; k++) {
        double phase = -dir * ORACLE_PI * k * k / n;
        chirp[2 * k] = cos(phase);
        chirp[2 * k + 1] = sin(phase);
    }
    for (int k = 0; k < n; k++) { /* a = x * conj(chirp)  (:107-109) */
        double xr = x[2 * k], xi = x[2 * k + 1];
        double cr = chirp[2 * k], ci = -chirp[2 * k + 1];
        a[2 * k] = xr * cr - xi * ci;
        a[2 * k + 1] = xr * ci + xi * cr;
    }
    for (int k = 0; k < n; k++) { /* (:116-121) */
        b[2 * k] = chirp[2 * k]; b[2 * k + 1] = chirp[2 * k + 1];
        if (k > 0) {
            b[2 * (m - k)] = chirp[2 * k];
            b[2 * (m - k) + 1] = chirp[2 * k + 1];
        }
    }
    oracle_radix2_dit(a, m, -1);
    oracle_radix2_dit(b, m, -1);
    for (int k = 0; k < m; k++) { /* (:128-130) */
        double ar = a[2 * k], ai = a[2 * k + 1];
        double br = b[2 * k], bi = b[2 * k + 1];
        a[2 * k] = ar * br - ai * bi;
        a[2 * k + 1] = ar * bi + ai * br;
    }
    oracle_radix2_dit(a, m, +1);
    for (int k = 0; k < n; k++) { /* (:139-141) */
        double ar = a[2 * k], ai = a[2 * k + 1];
        double cr = chirp[2 * k], ci = -chirp[2 * k + 1];
        x[2 * k] = ar * cr - ai * ci;
        x[2 * k + 1] = ar * ci + ai * cr;
    }
    if (dir > 0) {
        double dn = (double)n;
        for (int k = 0; k < 2 * n; k++) x[k] /= dn;
    }
    free(a); free(b); free(chirp);
    return 0;
}

/*
 * Accuracy truth (no reference counterpart; stated so): the same radix-2 DIT
 * schedule but with every twiddle taken from a table computed in long double
 * (exact to fp64 rounding) instead of the w *= w_m recurrence, which loses
 * ~N*eps.  Used to show the GPU result is at least as
 * close to the true DFT as the reference is.
 */
int oracle_radix2_dit_exact(double* x, int n, int dir) {
    if (!oracle_is_power_of_two(n)) return -1;
    int log2n = oracle_log2_int(n);
    double* tw = (double*)malloc(sizeof(double) * (size_t)(n > 1 ? n : 2));
    if (!tw) return -1;
    for (int k = 0; k < n / 2; k++) {
        long double ang = (long double)dir * 2.0L * 3.141592653589793238462643383279502884L * k / n;
        tw[2 * k] = (double)cosl(ang);
        tw[2 * k + 1] = (double)sinl(ang);
    }
    permute_bitrev(x, n, log2n);
    for (int stage = 1; stage <= log2n; stage++) {
        int m = 1 << stage, half = m >> 1, step = n / m;
        for (int k = 0; k < n; k += m) {
            for (int j = 0; j < half; j++) {
                int t = k + j, u = t + half;
                double wr = tw[2 * j * step], wi = tw[2 * j * step + 1];
                double ur = x[2 * u], ui = x[2 * u + 1];
                double pr = ur * wr - ui * wi, pi = ur * wi + ui * wr;
                double tr = x[2 * t], ti = x[2 * t + 1];
                x[2 * u] = tr - pr; x[2 * u + 1] = ti - pi;
                x[2 * t] = tr + pr; x[2 * t + 1] = ti + pi;
            }
        }
    }
    if (dir > 0) {
        double dn = (double)n;
        for (int i = 0; i < 2 * n; i++) x[i] /= dn;
    }
    free(tw);
    return 0;
}

/* O(n^2) DFT, algorithms/dft/naive_dft.c:55-97 in spirit (ground truth for
 * tiny n incl. n = 4, 8, 16 where the reference's bit_reverse is broken).
 * Uses exact integer phase reduction (j*k mod n) and long double sincos. */
int oracle_naive_dft(double* x, int n, int dir) {
    if (!x || n <= 0) return -1;
    double* y = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    if (!y) return -1;
    for (int k = 0; k < n; k++) {
        long double sr = 0, si = 0;
        for (int j = 0; j < n; j++) {
            long long p = ((long long)j * k) % n;
            long double ang = (long double)dir * 2.0L * 3.141592653589793238462643383279502884L * p / n;
            long double c = cosl(ang), s = sinl(ang);
            sr += x[2 * j] * c - x[2 * j + 1] * s;
            si += x[2 * j] * s + x[2 * j + 1] * c;
        }
        if (dir > 0) { sr /= n; si /= n; }
        y[2 * k] = (double)sr; y[2 * k + 1] = (double)si;
    }
    memcpy(x, y, sizeof(double) * 2 * (size_t)n);
    free(y);
    return 0;
}

/* ---- fp32 CPU comparators (timing baselines) ------------
 * Interleaved fp32 radix-2 DIT with the reference's loop structure
 * (radix2_dit.c:59-120) but twiddles from an fp64-generated table, so the
 * results are right (unlike optimizations/simd_fft.c). */
int oracle_radix2_dit_f32(float* x, int n, int dir) {
    if (!oracle_is_power_of_two(n)) return -1;
    int log2n = oracle_log2_int(n);
    float* tw = (float*)malloc(sizeof(float) * (size_t)(n > 1 ? n : 2));
    if (!tw) return -1;
    for (int k = 0; k < n / 2; k++) {
        double ang = dir * ORACLE_TWO_PI * k / n;
        tw[2 * k] = (float)cos(ang);
        tw[2 * k + 1] = (float)sin(ang);
    }
    for (int i = 0; i < n; i++) {
        int j = (int)oracle_bit_reverse((unsigned)i, log2n);
        if (i < j) {
            float tr = x[2 * i], ti = x[2 * i + 1];
            x[2 * i] = x[2 * j]; x[2 * i + 1] = x[2 * j + 1];
            x[2 * j] = tr; x[2 * j + 1] = ti;
        }
    }
    for (int stage = 1; stage <= log2n; stage++) {
        int m = 1 << stage, half = m >> 1, step = n / m;
        for (int k = 0; k < n; k += m) {
            for (int j = 0; j < half; j++) {
                int t = k + j, u = t + half;
                float wr = tw[2 * j * step], wi = tw[2 * j * step + 1];
                float ur = x[2 * u], ui = x[2 * u + 1];
                float pr = ur * wr - ui * wi, pi = ur * wi + ui * wr;
                float tr = x[2 * t], ti = x[2 * t + 1];
                x[2 * u] = tr - pr; x[2 * u + 1] = ti - pi;
                x[2 * t] = tr + pr; x[2 * t + 1] = ti + pi;
            }
        }
    }
    if (dir > 0) {
        float s = 1.0f / (float)n;
        for (int i = 0; i < 2 * n; i++) x[i] *= s;
    }
    free(tw);
    return 0;
}

/* fp32 SoA (split real[] / imag[]) radix-2 DIT, FOUR butterflies per step -- the structure of the reference's
 * optimizations/simd_fft.c:92-95 (complex_float_split_t), :114-140 (butterfly_sse2: 4 butterflies per _mm_load_ps) and
 * :143-230 (fft_radix2_sse2: bit reversal on the split arrays, then log2 n stages), but with a CORRECT twiddle for each of
 * the four lanes (the reference applies one twiddle to four adjacent butterflies and advances it by one step,
 * simd_fft.c:179-189; its self-test prints a 9.27 maximum error).  Twiddles per stage are gathered
 * into contiguous split arrays so the 4-wide loop below is straight loads / multiplies / stores that gcc turns into
 * SSE / AVX code at the reference's flags.  re / im: n floats each, 16-byte aligned.  Timing comparator and checker. */
int oracle_radix2_soa4_f32(float* re, float* im, int n, int dir) {
    if (!oracle_is_power_of_two(n)) return -1;
    const int log2n = oracle_log2_int(n);
    float* wr = (float*)malloc(sizeof(float) * (size_t)(n > 1 ? n : 2));
    if (!wr) return -1;
    float* wi = wr + n / 2 + (n < 2);
    for (int i = 0; i < n; i++) {
        const int j = (int)oracle_bit_reverse((unsigned)i, log2n);
        if (i < j) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int stage = 1; stage <= log2n; stage++) {
        const int m = 1 << stage, half = m >> 1;
        for (int j = 0; j < half; j++) {  /* this stage's twiddles W_m^j, contiguous */
            const double ang = dir * ORACLE_TWO_PI * (double)j / (double)m;
            wr[j] = (float)cos(ang);
            wi[j] = (float)sin(ang);
        }
        for (int k = 0; k < n; k += m) {
            float* tr = re + k;
            float* ti = im + k;
            float* ur = re + k + half;
            float* ui = im + k + half;
            int j = 0;
            for (; j + 4 <= half; j += 4) {  /* four butterflies, four twiddles */
                float pr[4], pi[4];
                for (int l = 0; l < 4; l++) {
                    pr[l] = ur[j + l] * wr[j + l] - ui[j + l] * wi[j + l];
                    pi[l] = ur[j + l] * wi[j + l] + ui[j + l] * wr[j + l];
                }
                for (int l = 0; l < 4; l++) {
                    ur[j + l] = tr[j + l] - pr[l];
                    ui[j + l] = ti[j + l] - pi[l];
                    tr[j + l] += pr[l];
                    ti[j + l] += pi[l];
                }
            }
            for (; j < half; j++) {  /* the first two stages: fewer than four butterflies per block */
                const float pr = ur[j] * wr[j] - ui[j] * wi[j], pi = ur[j] * wi[j] + ui[j] * wr[j];
                ur[j] = tr[j] - pr;
                ui[j] = ti[j] - pi;
                tr[j] += pr;
                ti[j] += pi;
            }
        }
    }
    if (dir > 0) {
        const float s = 1.0f / (float)n;
        for (int i = 0; i < n; i++) { re[i] *= s; im[i] *= s; }
    }
    free(wr);
    return 0;
}

/* batch of split-array transforms: transform b at re + b*n, im + b*n; OpenMP over the batch index */
int oracle_soa4_batch_f32(float* re, float* im, int n, long batch, int dir) {
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(| : rc)
#endif
    for (long b = 0; b < batch; b++) rc |= (oracle_radix2_soa4_f32(re + (size_t)b * (size_t)n, im + (size_t)b * (size_t)n, n, dir) != 0);
    return rc ? -1 : 0;
}

/* Batched drivers: transform b occupies [b*n, (b+1)*n) -- the contiguous
 * layout of cufftPlanMany in gpu/fft_cuda.cu:152-156.  algo: 0 dit,
 * 1 dif, 2 split_radix, 3 radix4, 4 bluestein, 5 exact-twiddle dit, 6 naive.
 * Parallel over the batch index with OpenMP when built with -fopenmp (the
 * "task parallel" strategy of optimizations/parallel_fft.c:424-427). */
int oracle_fft_batch(double* x, int n, long batch, int dir, int algo) {
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(| : rc)
#endif
    for (long b = 0; b < batch; b++) {
        double* p = x + 2 * (size_t)b * (size_t)n;
        int r;
        switch (algo) {
            case 0: r = oracle_radix2_dit(p, n, dir); break;
            case 1: r = oracle_radix2_dif(p, n, dir); break;
            case 2: r = oracle_radix4_or_split(p, n, dir, 0); break;
            case 3: r = oracle_radix4_or_split(p, n, dir, 1); break;
            case 4: r = oracle_bluestein(p, n, dir); break;
            case 5: r = oracle_radix2_dit_exact(p, n, dir); break;
            case 6: r = oracle_naive_dft(p, n, dir); break;
            default: r = -1;
        }
        rc |= (r != 0);
    }
    return rc ? -1 : 0;
}

int oracle_fft_batch_f32(float* x, int n, long batch, int dir) {
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(| : rc)
#endif
    for (long b = 0; b < batch; b++)
        rc |= (oracle_radix2_dit_f32(x + 2 * (size_t)b * (size_t)n, n, dir) != 0);
    return rc ? -1 : 0;
}

/* ---- deterministic synthetic inputs ----------------------
 * Two-tone complex sinusoid with an analytic spectrum: transform b has
 *   x_b[j] = exp(2 pi i (f_b j mod N)/N) + 0.5 exp(2 pi i (g_b j mod N)/N)
 *   f_b = (1 + 7 b) mod N,  g_b = (N/3 + 13 b) mod N  (g_b += 1 if == f_b)
 * so X_b[f_b] = N, X_b[g_b] = N/2 and 0 elsewhere (forward transform).
 * Mirrors the reference's known-transform test, tests/test_all.c:290-351. */
void oracle_two_tone_bins(long n, long b, long* f, long* g) {
    long fb = (1 + 7 * b) % n;
    long gb = (n / 3 + 13 * b) % n;
    if (gb == fb) gb = (gb + 1) % n;
    *f = fb; *g = gb;
}

void oracle_gen_two_tone(double* x, long n, long b0, long batch) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long bb = 0; bb < batch; bb++) {
        long f, g;
        oracle_two_tone_bins(n, b0 + bb, &f, &g);
        double* p = x + 2 * (size_t)bb * (size_t)n;
        for (long j = 0; j < n; j++) {
            uint64_t pf = ((uint64_t)f * (uint64_t)j) % (uint64_t)n;
            uint64_t pg = ((uint64_t)g * (uint64_t)j) % (uint64_t)n;
            double af = ORACLE_TWO_PI * (double)pf / (double)n;
            double ag = ORACLE_TWO_PI * (double)pg / (double)n;
            p[2 * j] = cos(af) + 0.5 * cos(ag);
            p[2 * j + 1] = sin(af) + 0.5 * sin(ag);
        }
    }
}

void oracle_gen_two_tone_f32(float* x, long n, long b0, long batch) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long bb = 0; bb < batch; bb++) {
        long f, g;
        oracle_two_tone_bins(n, b0 + bb, &f, &g);
        float* p = x + 2 * (size_t)bb * (size_t)n;
        for (long j = 0; j < n; j++) {
            uint64_t pf = ((uint64_t)f * (uint64_t)j) % (uint64_t)n;
            uint64_t pg = ((uint64_t)g * (uint64_t)j) % (uint64_t)n;
            double af = ORACLE_TWO_PI * (double)pf / (double)n;
            double ag = ORACLE_TWO_PI * (double)pg / (double)n;
            p[2 * j] = (float)(cos(af) + 0.5 * cos(ag));
            p[2 * j + 1] = (float)(sin(af) + 0.5 * sin(ag));
        }
    }
}

/* LCG noise, uniform [-0.5, 0.5) in re and im -- the reproducible stand-in
 * for the reference's rand()/RAND_MAX - 0.5 (tests/test_all.c:161,210). */
void oracle_gen_lcg(double* x, long n, long b0, long batch) {
    for (long bb = 0; bb < batch; bb++) {
        uint64_t s = 0x9E3779B97F4A7C15ull ^ (uint64_t)(b0 + bb);
        double* p = x + 2 * (size_t)bb * (size_t)n;
        for (long j = 0; j < 2 * n; j++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            p[j] = (double)(s >> 40) / 16777216.0 - 0.5;
        }
    }
}
