/*
 * fft_utils.h -- the reference's text interchange format for complex arrays (utils/fft_utils.c:77-145), so that data
 * written by the reference's tools loads here and the other way round:
 *
 *     # FFT Data File
 *     # Format: index real imag magnitude phase
 *     # Size: <n>
 *     <i> <re %e> <im %e> <|z| %e> <arg z %e>        one line per element
 *
 * load_complex_array reads the size from the "# Size:" header when present (else it counts the data lines), then the
 * first three fields of every non-comment line; magnitude and phase are redundant and ignored.  0 / -1 like the
 * reference; *data is allocated with allocate_complex_array and owned by the caller.
 */
#ifndef FFT_UTILS_H
#define FFT_UTILS_H

#include "fft_common.h"

#ifdef __cplusplus
extern "C" {
#endif

int save_complex_array(const char* filename, complex_t* data, int n);
int load_complex_array(const char* filename, complex_t** data, int* n);

#ifdef __cplusplus
}
#endif

#endif /* FFT_UTILS_H */
