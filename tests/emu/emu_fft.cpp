// emu_fft.cpp -- compiles the engine + kernels for the CPU emulation and
// exports a tiny C API for pytest (ctypes).  TEST INFRASTRUCTURE ONLY.
#include <time.h>

#include "emu_runtime.h"
#include "fft_engine.h"
#include "fft_plans_ext.h"

namespace emu {
thread_local dim3_ threadIdx_;
thread_local dim3_ blockIdx_;
dim3_ blockDim_;
dim3_ gridDim_;
thread_local unsigned char* smem_ = nullptr;
int xcc_skew = 0;
static thread_local pthread_barrier_t* g_barrier = nullptr;  // the barrier of this thread's workgroup

void sync_threads() { pthread_barrier_wait(g_barrier); }

long long clock_ticks() {  // 100 MHz, like the device's wall clock
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 100000000ll + ts.tv_nsec / 10;
}

static thread_local unsigned* g_xchg = nullptr;  // 1024 words per workgroup

// test hook (FFT_TEST_DROP): workgroup FFT_EMU_DROP_BLOCK leaves right after team formation, as if it hung
bool test_drop() {
    const char* b = getenv("FFT_EMU_DROP_BLOCK");
    return b && (unsigned)atoi(b) == blockIdx_.x;
}

// test hook (FFT_TEST_DELAY in team_form): workgroup FFT_EMU_LATE_BLOCK sleeps FFT_EMU_LATE_MS before it registers
void test_delay() {
    const char* b = getenv("FFT_EMU_LATE_BLOCK");
    const char* ms = getenv("FFT_EMU_LATE_MS");
    if (!b || !ms || (unsigned)atoi(b) != blockIdx_.x) return;
    struct timespec ts = {atoi(ms) / 1000, (long)(atoi(ms) % 1000) * 1000000l};
    nanosleep(&ts, nullptr);
}

unsigned shfl_xor_u32(unsigned v, int mask) {
    unsigned* xchg = g_xchg;
    xchg[threadIdx_.x] = v;
    sync_threads();
    unsigned r = xchg[(threadIdx_.x & ~63u) | ((threadIdx_.x ^ (unsigned)mask) & 63u)];
    sync_threads();
    return r;
}

void Runtime::run_grid(long long grid, int block, size_t smem, const std::function<void()>& body) {
    blockDim_ = {(unsigned)block, 1, 1};
    gridDim_ = {(unsigned)grid, 1, 1};
    std::vector<unsigned char> lds(smem + 64, 0xFF);  // NaN-filled: reads of unwritten LDS poison the result
    unsigned char* lds_base = lds.data() + ((16 - ((uintptr_t)lds.data() & 15)) & 15);
    pthread_barrier_t barrier;
    pthread_barrier_init(&barrier, nullptr, (unsigned)block);
    std::vector<unsigned> xchg(1024);
    std::vector<std::thread> th;
    th.reserve(block);
    for (int t = 0; t < block; t++) {
        th.emplace_back([&, t]() {
            threadIdx_ = {(unsigned)t, 0, 0};
            smem_ = lds_base;
            g_barrier = &barrier;
            g_xchg = xchg.data();
            for (long long b = 0; b < grid; b++) {
                blockIdx_ = {(unsigned)b, 0, 0};
                body();
                pthread_barrier_wait(&barrier);  // next block reuses the LDS image
            }
        });
    }
    for (auto& x : th) x.join();
    pthread_barrier_destroy(&barrier);
}

// Every workgroup of the launch runs at the same time (own LDS image, own barrier): what a kernel with
// inter-workgroup barriers (fft_team.h) needs.  grid * block host threads.
void Runtime::run_grid_coresident(long long grid, int block, size_t smem, const std::function<void()>& body) {
    blockDim_ = {(unsigned)block, 1, 1};
    gridDim_ = {(unsigned)grid, 1, 1};
    std::vector<std::vector<unsigned char>> lds((size_t)grid, std::vector<unsigned char>(smem + 64, 0xFF));
    std::vector<pthread_barrier_t> barriers((size_t)grid);
    std::vector<std::vector<unsigned>> xchg((size_t)grid, std::vector<unsigned>(1024));
    for (auto& b : barriers) pthread_barrier_init(&b, nullptr, (unsigned)block);
    std::vector<std::thread> th;
    th.reserve((size_t)grid * block);
    for (long long b = 0; b < grid; b++) {
        for (int t = 0; t < block; t++) {
            th.emplace_back([&, b, t]() {
                threadIdx_ = {(unsigned)t, 0, 0};
                blockIdx_ = {(unsigned)b, 0, 0};
                unsigned char* base = lds[(size_t)b].data();
                smem_ = base + ((16 - ((uintptr_t)base & 15)) & 15);
                g_barrier = &barriers[(size_t)b];
                g_xchg = xchg[(size_t)b].data();
                body();
            });
        }
    }
    for (auto& x : th) x.join();
    for (auto& b : barriers) pthread_barrier_destroy(&b);
}
}  // namespace emu

template <typename T>
static int run(const void* in, void* out, int n, int batch, int dir, int algo, int lds_budget, int* info, int team = 0) {
    emu::Runtime rt;
    if (lds_budget > 0) rt.lds_budget = lds_budget;
    rt.team_mode = team;
    emu::xcc_skew = rt.team_grid_skew();
    using C = fftk::cpx<T>;
    if (n >= 1 && (n & (n - 1)) == 0) {
        ffteng::Pow2Plan<T, emu::Runtime> plan;
        if (!plan.build(&rt, ffteng::ilog2(n), algo, batch)) return -1;
        if (info) {
            info[0] = (int)plan.passes.size();
            for (size_t i = 0; i < plan.passes.size() && i < 3; i++) {
                info[1 + 2 * i] = plan.passes[i].log2L;
                info[2 + 2 * i] = plan.passes[i].log2C;
            }
            info[7] = plan.chunk;
            if (plan.team.ok) info[0] += 100 * plan.team.NT;  // 100*NT + passes: the team kernel is planned
            info[6] = plan.team.ok ? (plan.team.asplit ? 1 : 0) + (plan.team.defer ? 2 : 0) + (plan.team.pair ? 4 : 0) + (plan.team.quad ? 8 : 0) : 0;
            if (plan.wide.ok) info[6] |= 16;  // wide_row_kernel (fft_wide_row.h) runs the plain executes
        }
        plan.execute((const C*)in, (C*)out, batch, dir > 0);
        // FFT_EMU_PINGPONG: a second execute back into the first one's input before the "sync" (A -> B, then B -> A): after a timeout
        // neither can be repeated -- A was overwritten by the later launch, B was written by a launch that is void
        if (getenv("FFT_EMU_PINGPONG") && in != out) plan.execute((const C*)out, (C*)in, batch, dir > 0);
        if (plan.team.ctl && getenv("FFT_EMU_RECOVER")) {
            // what the HIP backend does when it synchronizes (team_status_of): a timed-out team kernel's executes are
            // replayed on the multi-pass schedule; info[3] = executes that could not be (in place)
            const bool timed_out = plan.team.sticky[fftk::TEAM_STICKY_TIMEOUTS] != 0;
            const int lost = plan.recover_after_timeout(timed_out);
            if (info) info[3] = 1000 + lost + (timed_out ? 100 : 0);
        }
        if (info && plan.team.ctl) {  // what the team kernel reported (the emulation's memory is the host's)
            info[5] = 1 + (int)plan.team.ctl[fftk::TEAM_CTL_STATUS] + 10 * (int)plan.team.sticky[fftk::TEAM_STICKY_FALLBACKS] +
                      100 * (int)plan.team.sticky[fftk::TEAM_STICKY_TIMEOUTS];
        }
    } else {
        ffteng::BluesteinPlan<T, emu::Runtime> plan;
        if (!plan.build(&rt, n, dir, algo, batch)) return -1;
        if (getenv("FFT_EMU_NO_FUSION")) plan.no_fusion = true;
        if (getenv("FFT_EMU_NO_CHAIN")) plan.no_chain = true;
        plan.core.chain_min_log2n = 0;  // the emulated sizes are small: every plan whose tiles agree chains
        if (info) {
            info[0] = 10 + (int)plan.core.passes.size();
            info[4] = (plan.core.hook_capable() && !plan.no_fusion) ? 1 : 0;  // element-wise steps fused into the FFT passes
            if (info[4] && !plan.no_chain && plan.core.chain_capable()) info[4] = 2;  // ... and forward-last + inverse-first as one kernel
            if (info[4] && !plan.no_chain && plan.core.round_capable()) info[4] = 3;  // single pass: FFT -> product -> inverse FFT as ONE kernel
            info[7] = plan.core.chunk;
        }
        plan.execute((const C*)in, (C*)out, batch);
    }
    return 0;
}

extern "C" int emu_fft(const void* in, void* out, int n, int batch, int dir, int prec, int algo, int lds_budget,
                       int* info) {
    return prec == 1 ? run<float>(in, out, n, batch, dir, algo, lds_budget, info)
                     : run<double>(in, out, n, batch, dir, algo, lds_budget, info);
}

// team kernel (fft_team.h) with a small geometry: team_mode = 1 + log2TS | n_teams << 4 | threads << 8 | fail << 20
extern "C" int emu_fft_team(const void* in, void* out, int n, int batch, int dir, int prec, int lds_budget, int team_mode,
                            int* info) {
    return prec == 1 ? run<float>(in, out, n, batch, dir, 0, lds_budget, info, team_mode)
                     : run<double>(in, out, n, batch, dir, 0, lds_budget, info, team_mode);
}

extern "C" int emu_bitrev(const void* in, void* out, int n, int batch, int prec) {
    emu::Runtime rt;
    const long long total = (long long)n * batch;
    const int log2n = ffteng::ilog2(n);
    if (prec == 1)
        rt.launch(fftk::bitrev_kernel<float>, 4, 256, (size_t)0, (const fftk::cpx<float>*)in, (fftk::cpx<float>*)out, log2n, total);
    else
        rt.launch(fftk::bitrev_kernel<double>, 4, 256, (size_t)0, (const fftk::cpx<double>*)in, (fftk::cpx<double>*)out, log2n, total);
    return 0;
}

// ---- plans built on the engine (fft_plans_ext.h)
template <typename T>
static int run2d(const void* in, void* out, int rows, int cols, int nm, int dir, int lds_budget, int* info) {
    emu::Runtime rt;
    if (lds_budget > 0) rt.lds_budget = lds_budget;
    ffteng::Plan2D<T, emu::Runtime> plan;
    if (!plan.build(&rt, rows, cols, dir, nm)) return -1;
    if (info) info[0] = plan.colp ? (plan.colp->passes.size() == 2 ? 3 : 1) : (plan.colt ? 2 : 0);  // 1 direct column pass, 2 transpose path, 3 two strided passes
    plan.execute((const fftk::cpx<T>*)in, (fftk::cpx<T>*)out, nm);
    return 0;
}
extern "C" int emu_fft2d(const void* in, void* out, int rows, int cols, int nm, int dir, int prec, int lds_budget, int* info) {
    return prec == 1 ? run2d<float>(in, out, rows, cols, nm, dir, lds_budget, info) : run2d<double>(in, out, rows, cols, nm, dir, lds_budget, info);
}

template <typename T>
static int run_real(const void* in, void* out, int n, int batch, int r2c) {
    emu::Runtime rt;
    ffteng::RealPlan<T, emu::Runtime> plan;
    if (!plan.build(&rt, n, r2c != 0, batch)) return -1;
    if (r2c) plan.execute_r2c((const T*)in, (fftk::cpx<T>*)out, batch);
    else plan.execute_c2r((const fftk::cpx<T>*)in, (T*)out, batch);
    return 0;
}
extern "C" int emu_real(const void* in, void* out, int n, int batch, int r2c, int prec) {
    return prec == 1 ? run_real<float>(in, out, n, batch, r2c) : run_real<double>(in, out, n, batch, r2c);
}

template <typename T>
static int run_fused(int kind, const void* x, const void* y, const void* h, int nx, int nh, void* out, int batch, int lds_budget, int no_fusion,
                     double fs, int* info) {
    emu::Runtime rt;
    if (lds_budget > 0) rt.lds_budget = lds_budget;
    ffteng::FusedPlan<T, emu::Runtime> plan;
    if (!plan.build(&rt, kind, nx, nh, (const fftk::cpx<T>*)h, batch)) return -1;
    plan.no_fusion = no_fusion != 0;
    if (getenv("FFT_EMU_NO_CHAIN")) plan.no_chain = true;
    plan.core.chain_min_log2n = 0;
    if (info) {
        info[0] = (int)plan.core.passes.size();
        info[1] = plan.fused() ? ((!plan.no_chain && plan.core.chain_capable() && kind != ffteng::FUSED_PSD) ? 2 : 1) : 0;
        if (plan.fused() && !plan.no_chain && plan.core.round_capable() && kind != ffteng::FUSED_PSD && kind != ffteng::FUSED_XCORR) info[1] = 3;
        info[2] = plan.log2m;
        info[3] = plan.core.chunk;
    }
    plan.execute((const fftk::cpx<T>*)x, (const fftk::cpx<T>*)y, out, batch, (T)fs);
    return 0;
}
extern "C" int emu_fused(int kind, const void* x, const void* y, const void* h, int nx, int nh, void* out, int batch, int prec, int lds_budget,
                         int no_fusion, double fs, int* info) {
    return prec == 1 ? run_fused<float>(kind, x, y, h, nx, nh, out, batch, lds_budget, no_fusion, fs, info)
                     : run_fused<double>(kind, x, y, h, nx, nh, out, batch, lds_budget, no_fusion, fs, info);
}
