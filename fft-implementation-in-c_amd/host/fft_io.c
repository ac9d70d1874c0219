/*
 * fft_io.c -- save / load of the reference's text format for complex arrays (see include/fft_utils.h; reference
 * utils/fft_utils.c:77-145).  Plain C, no device involved.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fft_utils.h"

int save_complex_array(const char* filename, complex_t* data, int n) {
    if (!filename || (!data && n > 0) || n < 0) return -1;
    FILE* fp = fopen(filename, "w");
    if (!fp) {
        fprintf(stderr, "Error: Cannot open file %s for writing\n", filename);
        return -1;
    }
    fputs("# FFT Data File\n# Format: index real imag magnitude phase\n", fp);
    fprintf(fp, "# Size: %d\n", n);
    int rc = 0;
    for (int i = 0; i < n && rc == 0; i++) {
        const double re = creal(data[i]), im = cimag(data[i]);
        if (fprintf(fp, "%d %e %e %e %e\n", i, re, im, cabs(data[i]), carg(data[i])) < 0) rc = -1;
    }
    if (fclose(fp) != 0) rc = -1;
    return rc;
}

int load_complex_array(const char* filename, complex_t** data, int* n) {
    if (!filename || !data || !n) return -1;
    FILE* fp = fopen(filename, "r");
    if (!fp) {
        fprintf(stderr, "Error: Cannot open file %s for reading\n", filename);
        return -1;
    }
    char line[512];
    int declared = -1, counted = 0;
    while (fgets(line, sizeof(line), fp)) {
        if (line[0] == '#') {
            const char* s = strstr(line, "Size:");
            if (s && declared < 0) (void)sscanf(s, "Size: %d", &declared);
        } else {
            int i;
            double re, im;
            if (sscanf(line, "%d %lf %lf", &i, &re, &im) == 3) counted++;
        }
    }
    /* the header wins when it is there (as in the reference); a file without one is as long as its data lines */
    const int want = declared >= 0 ? declared : counted;
    if (want <= 0) {
        fclose(fp);
        return -1;
    }
    complex_t* a = allocate_complex_array(want);
    if (!a) {
        fclose(fp);
        return -1;
    }
    rewind(fp);
    int idx = 0;
    while (idx < want && fgets(line, sizeof(line), fp)) {
        if (line[0] == '#') continue;
        int i;
        double re, im;
        if (sscanf(line, "%d %lf %lf", &i, &re, &im) == 3) a[idx++] = re + I * im;
    }
    fclose(fp);
    *data = a; /* elements the file does not provide stay 0 (calloc), as in the reference */
    *n = want;
    return 0;
}
