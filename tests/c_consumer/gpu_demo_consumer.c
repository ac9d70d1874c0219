/*
 * gpu_demo_consumer.c -- a compiled-C consumer of the drop-in boundary (test infrastructure).
 *
 * Restates, from scratch, the call sequence of the reference's GPU harness
 * examples/demo_v2_features.c (hardware detection :206-229, automatic selection :50-92, GPU
 * acceleration :95-156, simplified API :159-202) against include/fft_auto.h + include/fft_gpu.h,
 * built with plain `gcc -std=c99` and linked against libfft_mi355x.so -- what a maintainer's C program
 * does, with no Python in between.  Unlike the demo it CHECKS what it computes: every spectrum is
 * compared with the closed-form answer of its input (complex two-tone: X[f] = n, X[g] = n/2, 0 elsewhere).
 * Exit code 0 and a final "c-consumer: OK" line mean success.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fft_auto.h"
#include "fft_gpu.h"

static int failures = 0;

#define EXPECT(cond, ...)                     \
    do {                                      \
        if (!(cond)) {                        \
            fprintf(stderr, "FAIL: " __VA_ARGS__); \
            fprintf(stderr, "\n");            \
            failures++;                       \
        }                                     \
    } while (0)

/* x[j] = exp(2 pi i f j / n) + 0.5 exp(2 pi i g j / n), phases reduced in integers */
static void two_tone(complex_t* x, int n, int f, int g) {
    for (int j = 0; j < n; j++) {
        const double a = TWO_PI * (double)(((long long)f * j) % n) / n;
        const double b = TWO_PI * (double)(((long long)g * j) % n) / n;
        x[j] = (cos(a) + I * sin(a)) + 0.5 * (cos(b) + I * sin(b));
    }
}

/* max |X[k] - expected[k]| / n over all bins */
static double spectrum_error(const complex_t* X, int n, int f, int g) {
    double worst = 0;
    for (int k = 0; k < n; k++) {
        const double want = k == f ? (double)n : (k == g ? 0.5 * n : 0.0);
        const double e = cabs(X[k] - want) / n;
        if (e > worst) worst = e;
    }
    return worst;
}

static void hardware_detection(void) {
    const unsigned caps = fft_get_hardware_capabilities();
    printf("capabilities: 0x%x%s\n", caps, (caps & FFT_HW_GPU_HIP) ? " (HIP GPU)" : "");
    EXPECT(caps & FFT_HW_GPU_HIP, "fft_get_hardware_capabilities() does not report the HIP GPU");
}

static void automatic_selection(void) {
    const int sizes[] = {64, 256, 1024, 4096, 16384, 97, 360, 1000};
    for (int i = 0; i < 8; i++) {
        const int n = sizes[i];
        complex_t* data = fft_alloc_complex((size_t)n);
        CHECK_NULL(data, "fft_alloc_complex");
        fft_plan_t plan = fft_plan_dft_1d(n, data, data, -1, FFT_ESTIMATE);
        EXPECT(plan != NULL, "fft_plan_dft_1d(%d) returned NULL", n);
        if (plan) {
            const int f = 3 % n, g = (n / 3 + 1) % n;
            two_tone(data, n, f, g);
            fft_timer_t t;
            timer_start(&t);
            fft_execute(plan);
            timer_stop(&t);
            const double e = spectrum_error(data, n, f, g);
            printf("  n = %-6d %s  max error %.2e  (%.3f ms cpu)\n", n, is_power_of_two(n) ? "pow2     " : "bluestein", e, t.elapsed_ms);
            EXPECT(e < 1e-9, "n = %d: spectrum error %.3e", n, e);
            fft_destroy_plan(plan);
        }
        fft_free(data);
    }
}

static void gpu_acceleration(void) {
    EXPECT(fft_gpu_available(), "fft_gpu_available() == 0");
    if (!fft_gpu_available()) return;
    EXPECT(fft_gpu_init(FFT_GPU_AUTO) == 0, "fft_gpu_init(FFT_GPU_AUTO) failed");
    printf("GPU device: %s\n", fft_gpu_get_device_name());
    size_t total = 0, avail = 0;
    fft_gpu_get_memory_info(&total, &avail);
    printf("GPU memory: %.1f GB total, %.1f GB available\n", total / 1e9, avail / 1e9);
    EXPECT(total > 0 && avail > 0 && avail <= total, "fft_gpu_get_memory_info: total %zu available %zu", total, avail);

    const int sizes[] = {1024, 4096, 16384, 65536, 262144};
    for (int i = 0; i < 5; i++) {
        const int n = sizes[i];
        complex_t* signal = fft_alloc_complex((size_t)n);
        complex_t* keep = fft_alloc_complex((size_t)n);
        CHECK_NULL(signal, "fft_alloc_complex");
        CHECK_NULL(keep, "fft_alloc_complex");
        const int f = 5, g = n / 3 + 2;
        two_tone(keep, n, f, g);
        /* the demo's two plans: default flags and FFT_PREFER_GPU (both run on the HIP engine here) */
        fft_plan_t plan_default = fft_plan_dft_1d(n, signal, signal, -1, 0);
        fft_plan_t plan_gpu = fft_plan_dft_1d(n, signal, signal, -1, FFT_PREFER_GPU);
        EXPECT(plan_default && plan_gpu, "plans for n = %d", n);
        if (plan_default && plan_gpu) {
            fft_timer_t t;
            memcpy(signal, keep, (size_t)n * sizeof(complex_t));
            timer_start(&t);
            fft_execute(plan_default);
            timer_stop(&t);
            const double e0 = spectrum_error(signal, n, f, g);
            memcpy(signal, keep, (size_t)n * sizeof(complex_t));
            fft_execute(plan_gpu);
            const double e1 = spectrum_error(signal, n, f, g);
            /* out-of-place through fft_execute_dft, then back with an inverse plan */
            complex_t* out = fft_alloc_complex((size_t)n);
            CHECK_NULL(out, "fft_alloc_complex");
            fft_execute_dft(plan_gpu, keep, out);
            const double e2 = spectrum_error(out, n, f, g);
            fft_plan_t inv = fft_plan_dft_1d(n, out, out, +1, FFT_PREFER_GPU);
            double e3 = 1.0;
            if (inv) {
                fft_execute(inv);
                e3 = 0;
                for (int j = 0; j < n; j++) {
                    const double d = cabs(out[j] - keep[j]);
                    if (d > e3) e3 = d;
                }
                fft_destroy_plan(inv);
            }
            printf("  n = %-7d errors %.1e %.1e %.1e  round trip %.1e  (%.2f ms cpu)\n", n, e0, e1, e2, e3, t.elapsed_ms);
            EXPECT(e0 < 1e-9 && e1 < 1e-9 && e2 < 1e-9 && e3 < 1e-9, "n = %d", n);
            fft_free(out);
        }
        fft_destroy_plan(plan_default);
        fft_destroy_plan(plan_gpu);
        fft_free(signal);
        fft_free(keep);
    }

    /* the device-resident route of include/fft_gpu.h: alloc, h2d, batched plan, execute, d2h */
    {
        const int n = 4096, batch = 8;
        complex_t* host = fft_alloc_complex((size_t)n * batch);
        CHECK_NULL(host, "fft_alloc_complex");
        for (int b = 0; b < batch; b++) two_tone(host + (size_t)b * n, n, 1 + 7 * b, n / 3 + 13 * b);
        fft_gpu_memory_t mem = fft_gpu_alloc((size_t)n * batch);
        fft_gpu_plan_t plan = fft_gpu_plan_1d(n, batch, FFT_FORWARD);
        EXPECT(mem && plan, "fft_gpu_alloc / fft_gpu_plan_1d");
        if (mem && plan) {
            fft_gpu_copy_h2d(mem, host, (size_t)n * batch);
            fft_gpu_execute(plan, mem, mem);
            fft_gpu_copy_d2h(host, mem, (size_t)n * batch);
            for (int b = 0; b < batch; b++) {
                const double e = spectrum_error(host + (size_t)b * n, n, 1 + 7 * b, n / 3 + 13 * b);
                EXPECT(e < 1e-9, "batched plan, transform %d: %.3e", b, e);
            }
        }
        fft_gpu_destroy_plan(plan);
        fft_gpu_free(mem);
        fft_free(host);
    }
    fft_gpu_cleanup();
}

static void simplified_api(void) {
    const int n = 1024;
    complex_t* signal = fft_alloc_complex((size_t)n);
    CHECK_NULL(signal, "fft_alloc_complex");
    const double fs = 1024.0;
    generate_sine_wave(signal, n, 50.0, fs); /* a real 50 Hz sine: peaks of n/2 at bins 50 and n - 50 */
    EXPECT(fft_auto(signal, signal, n, -1) == 0, "fft_auto");
    double* mag = compute_magnitude(signal, n);
    double* power = compute_power_spectrum(signal, n);
    double* phase = compute_phase(signal, n);
    int peaks = 0;
    for (int i = 0; i < n; i++)
        if (mag[i] > 100) peaks++;
    printf("fft_auto: |X[50]| = %.3f, |X[%d]| = %.3f, %d peaks\n", mag[50], n - 50, mag[n - 50], peaks);
    EXPECT(peaks == 2 && fabs(mag[50] - n / 2.0) < 1e-6 && fabs(mag[n - 50] - n / 2.0) < 1e-6, "sine spectrum");
    EXPECT(fabs(power[50] - (n / 2.0) * (n / 2.0) / n) < 1e-6, "power spectrum");
    EXPECT(fabs(phase[50] + PI / 2) < 1e-9, "phase of a sine's positive-frequency bin is -pi/2 (got %.6f)", phase[50]);
    free(mag);
    free(power);
    free(phase);

    generate_impulse(signal, n);
    EXPECT(fft_auto(signal, signal, n, -1) == 0, "fft_auto (impulse)");
    double worst = 0;
    for (int k = 0; k < n; k++) {
        const double d = cabs(signal[k] - 1.0);
        if (d > worst) worst = d;
    }
    EXPECT(worst < 1e-12, "impulse spectrum is all ones (%.3e)", worst);
    print_complex_array("first bins", signal, 4);
    fft_free(signal);
}

int main(void) {
    printf("library version: %s\n", fft_version());
    hardware_detection();
    automatic_selection();
    simplified_api();
    gpu_acceleration();
    if (failures) {
        printf("c-consumer: %d check(s) FAILED\n", failures);
        return 1;
    }
    printf("c-consumer: OK\n");
    return 0;
}
