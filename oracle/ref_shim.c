/* ref_shim.c -- exports the reference's `static inline` helpers from
 * include/fft_common.h (which have no linkable symbol) so tests can pin the
 * oracle against them.  Compiled only by `make ref` against the header where
 * it lies under /root/reference; contains no reference code itself. */
#include "fft_common.h"

unsigned ref_bit_reverse(unsigned x, int log2n) { return bit_reverse(x, log2n); }
int ref_next_power_of_two(int n) { return next_power_of_two(n); }
int ref_log2_int(int n) { return log2_int(n); }
int ref_is_power_of_two(int n) { return is_power_of_two(n); }
void ref_twiddle_factor(int k, int n, int dir, double* re, double* im) {
    complex_t w = twiddle_factor(k, n, (fft_direction)dir);
    *re = creal(w);
    *im = cimag(w);
}
