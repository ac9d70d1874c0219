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
// the same kernel (E = 4, radix-4: AUTO's single-pass choice) with L and C baked in for the full 64 KiB tile, where that measured
// faster (profiles/r2_ab_rows_fixed.txt: n = 128, 256 fp32 +10...12 %; n >= 512 and every fp64 size LOSE 10...30 % -- the unrolled
// stage loop no longer fits the 128-VGPR budget of this kernel)
// X(T, LOG2L, LOG2C)
#define FFT_ROWS_FIXED_LIST(X) X(float, 7, 6) X(float, 8, 5)
// E = 8, radix-8 (AUTO's single-pass choice from n = 512 up), L and C baked in: fp32 only (profiles/r2_ab_rows_fixed.txt: fp32 +12...14 %
// over the generic E = 8 kernel, +14...26 % over E = 4 radix-4; the fp64 ones lose 30 %, generic E = 8 radix-8 is fp64's best).
// X(T, LOG2L, LOG2C)
#define FFT_ROWS_FIXED8_LIST(X) X(float, 9, 4) X(float, 10, 3) X(float, 11, 2) X(float, 12, 1)
