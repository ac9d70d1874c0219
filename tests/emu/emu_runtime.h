// emu_runtime.h -- a workgroup = host threads, __syncthreads() = a barrier.
// TEST INFRASTRUCTURE ONLY (see fft_device.h): lets the build container, which
// has no GPU, execute the unmodified kernel source to check its index algebra.
#pragma once
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <thread>
#include <vector>

#include "fft_device.h"

namespace emu {

// mirror of ffteng::EnginePolicy for the emulation (fft_engine.h is included after this header); the tests steer it
// through the same environment variables the product reads at fft_gpu_init_hip
struct ffteng_policy {
    int team_mode = 1;
    int team_min_batch = 0;
    long long chunk_mb = 0;
    ffteng_policy() {
        if (const char* e = getenv("FFT_HIP_TEAM")) team_mode = atoi(e);
        if (const char* e = getenv("FFT_HIP_TEAM_MIN_BATCH")) team_min_batch = atoi(e);
        if (const char* e = getenv("FFT_HIP_CHUNK_MB")) chunk_mb = atoll(e);
    }
};

struct Runtime {
    void* dmalloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
    void dfree(void* p) { free(p); }
    void h2d(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
    int lds_budget = 160 * 1024;
    int max_lds_bytes() { return lds_budget; }
    int cus = 3;  // few "CUs" so that every workgroup walks several tiles (exercises the persistent loop + prefetch)
    int num_cus() { return cus; }
    void mark(int) {}
    void memset_async(void* p, int v, size_t bytes) { memset(p, v, bytes); }
    void d2d_async(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
    // team kernel geometry for the emulation (0 = no team kernel): bits 0-3 log2(seats per "XCD") + 1, 4-7 "XCDs",
    // 8-19 threads, bit 20: pretend the placement is wrong (one "XCD" gets a workgroup too many) to exercise the fallback
    int team_mode = 0;
    bool team_geometry(int& log2seats, int& n_xcc, int& nthreads) {
        if (!team_mode) return false;
        log2seats = (team_mode & 15) - 1;
        n_xcc = (team_mode >> 4) & 15;
        nthreads = (team_mode >> 8) & 4095;
        return true;
    }
    bool team_default_on(int, int) { return true; }
    bool team_defer(int, int) { return getenv("FFT_EMU_TEAM_PLAIN") == nullptr; }  // the shipped default; FFT_EMU_TEAM_PLAIN: team_fft_kernel
    bool team_asplit(int, int) { return getenv("FFT_EMU_TEAM_ASPLIT") != nullptr; }
    bool team_quad(int, int log2n) { return (log2n >= 10 && log2n <= 12) && getenv("FFT_EMU_TEAM_QUAD") != nullptr; }
    bool wide_rows(int elem_bytes, int log2n) { return elem_bytes == 8 && log2n == 9 && getenv("FFT_EMU_WIDE") != nullptr; }
    bool team_alll2(int, int) { return getenv("FFT_EMU_TEAM_ALLL2") != nullptr; }
    bool team_nodefer(int, int) { return getenv("FFT_EMU_TEAM_NODEFER") != nullptr; }
    bool team_pair(int elem_bytes, int) { return elem_bytes == 8 && getenv("FFT_EMU_TEAM_PAIR") != nullptr; }
    long long team_timeout_ticks() {  // FFT_EMU_TEAM_TIMEOUT_MS: short, for the test with a member that never arrives
        const char* e = getenv("FFT_EMU_TEAM_TIMEOUT_MS");
        return e ? atoll(e) * 100000ll : 60ll * 100000000ll;
    }
    // formation: generous by default (host threads start slowly); FFT_EMU_FORM_TIMEOUT_MS makes it short for the test
    // that delays one workgroup past it
    long long team_form_timeout_ticks() {
        const char* e = getenv("FFT_EMU_FORM_TIMEOUT_MS");
        return e ? atoll(e) * 100000ll : 60ll * 100000000ll;
    }
    ffteng_policy policy;
    int team_grid_skew() { return (team_mode >> 20) & 1; }
    template <class K>
    int max_blocks_per_cu(K, int, size_t) { return 1; }
    long long launches = 0;

    template <class K, class... A>
    void launch(K kernel, long long grid, int block, size_t smem, A... args) {
        launches++;
        run_grid(grid, block, smem, [&]() { kernel(args...); });
    }
    template <class K, class... A>
    void launch_coresident(K kernel, long long grid, int block, size_t smem, A... args) {
        launches++;
        run_grid_coresident(grid, block, smem, [&]() { kernel(args...); });
    }
    static void run_grid(long long grid, int block, size_t smem, const std::function<void()>& body);
    static void run_grid_coresident(long long grid, int block, size_t smem, const std::function<void()>& body);
};

}  // namespace emu
