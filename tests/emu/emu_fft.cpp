// emu_fft.cpp -- compiles the engine + kernels for the CPU emulation and
// exports a tiny C API for pytest (ctypes).  TEST INFRASTRUCTURE ONLY.
#include "emu_runtime.h"
#include "fft_engine.h"

namespace emu {
thread_local dim3_ threadIdx_;
thread_local dim3_ blockIdx_;
dim3_ blockDim_;
dim3_ gridDim_;
unsigned char* smem_ = nullptr;
static pthread_barrier_t g_barrier;

void sync_threads() { pthread_barrier_wait(&g_barrier); }

unsigned shfl_xor_u32(unsigned v, int mask) {
    static unsigned xchg[1024];
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
    smem_ = lds.data() + ((16 - ((uintptr_t)lds.data() & 15)) & 15);
    pthread_barrier_init(&g_barrier, nullptr, (unsigned)block);
    std::vector<std::thread> th;
    th.reserve(block);
    for (int t = 0; t < block; t++) {
        th.emplace_back([&, t]() {
            threadIdx_ = {(unsigned)t, 0, 0};
            for (long long b = 0; b < grid; b++) {
                blockIdx_ = {(unsigned)b, 0, 0};
                body();
                pthread_barrier_wait(&g_barrier);  // next block reuses the LDS image
            }
        });
    }
    for (auto& x : th) x.join();
    pthread_barrier_destroy(&g_barrier);
    smem_ = nullptr;
}
}  // namespace emu

template <typename T>
static int run(const void* in, void* out, int n, int batch, int dir, int algo, int lds_budget, int* info) {
    emu::Runtime rt;
    if (lds_budget > 0) rt.lds_budget = lds_budget;
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
        }
        plan.execute((const C*)in, (C*)out, batch, dir > 0);
    } else {
        ffteng::BluesteinPlan<T, emu::Runtime> plan;
        if (!plan.build(&rt, n, dir, algo, batch)) return -1;
        if (info) info[0] = 10 + (int)plan.core.passes.size();
        plan.execute((const C*)in, (C*)out, batch);
    }
    return 0;
}

extern "C" int emu_fft(const void* in, void* out, int n, int batch, int dir, int prec, int algo, int lds_budget,
                       int* info) {
    return prec == 1 ? run<float>(in, out, n, batch, dir, algo, lds_budget, info)
                     : run<double>(in, out, n, batch, dir, algo, lds_budget, info);
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
