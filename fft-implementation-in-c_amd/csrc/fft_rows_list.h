// fft_rows_list.h -- the rows-in / rows-out (single-pass) instantiations of tile_fft_kernel.
// They live in their own translation unit (fft_rows_o2.hip) because hipcc -O2 schedules this kernel shape
// measurably better than -O3 (N = 1024 x 65536 fp32: 236 vs 205 Gpoint/s) while the multi-pass shapes prefer -O3.
// X(T, E, FAM)
#define FFT_ROWS_LIST(X)        \
    X(float, 2, FAM_R2)         \
    X(float, 4, FAM_R4)         \
    X(float, 4, FAM_R2)         \
    X(float, 8, FAM_SR16)       \
    X(float, 8, FAM_R4)         \
    X(float, 8, FAM_R2)         \
    X(double, 2, FAM_R2)        \
    X(double, 4, FAM_R4)        \
    X(double, 4, FAM_R2)        \
    X(double, 8, FAM_SR16)      \
    X(double, 8, FAM_R4)        \
    X(double, 8, FAM_R2)
