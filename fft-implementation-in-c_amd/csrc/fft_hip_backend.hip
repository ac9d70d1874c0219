// fft_hip_backend.hip -- the C-ABI shim between the plain-C host side
// (host/fft_gpu.c, host/fft_auto.c) and the hand-written HIP kernels.
//
// Exports exactly the symbols declared in include/fft_hip.h: the 13-function
// backend set the reference's dispatcher expects (gpu/fft_gpu.c:32-46, CUDA
// flavour implemented by gpu/fft_cuda.cu:53-252 around cuFFT) plus the additive
// fp32 / batch / stream / multi-device entry points.  No hipFFT / rocFFT.
//
// Error style follows the reference's backend (fft_cuda.cu:34-50): int 0 / -1,
// NULL handles, a diagnostic on stderr; never exit().
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <new>
#include <set>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/fft_hip.h"
#include "fft_kernels.h"
#include "fft_rows_list.h"

namespace fftk {  // instantiated in fft_rows_o2.hip
#define FFT_EXTERN(T, E, FAM) \
    extern template __global__ void tile_fft_kernel<T, E, 1, FAM, LOAD_LCONTIG, STORE_LCONTIG, false, 0>(TileParams<T>);
FFT_ROWS_LIST(FFT_EXTERN)
#undef FFT_EXTERN
#define FFT_EXTERN_FIXED(T, LOG2L, LOG2C) \
    extern template __global__ void tile_fft_kernel<T, 4, 1, FAM_R4, LOAD_LCONTIG, STORE_LCONTIG, false, ((LOG2L) << 8) | (LOG2C)>(TileParams<T>);
FFT_ROWS_FIXED_LIST(FFT_EXTERN_FIXED)
#undef FFT_EXTERN_FIXED
#define FFT_EXTERN_FIXED8(T, LOG2L, LOG2C) \
    extern template __global__ void tile_fft_kernel<T, 8, 1, FAM_SR16, LOAD_LCONTIG, STORE_LCONTIG, false, ((LOG2L) << 8) | (LOG2C)>(TileParams<T>);
FFT_ROWS_FIXED8_LIST(FFT_EXTERN_FIXED8)
#undef FFT_EXTERN_FIXED8
}  // namespace fftk

// the exchange protocol of team_quad_kernel by size (team_quad_slots): 3 = one image per seat + the pair protocol where it measured faster than
// the team counter (profiles/r4_ab_pair_protocol_sizes.txt: n = 2^20 +1.7 %, 2^19 +3.7 %, 2^18 +-0, 2^17 -1.5 % with its first schedule) -- and it keeps the
// window in the L2
#ifndef FFT_QUAD_SLOTS19
#define FFT_QUAD_SLOTS19 3
#endif
#ifndef FFT_QUAD_SLOTS20
#define FFT_QUAD_SLOTS20 3
#endif
#ifndef FFT_QUAD_SLOTS18  // (with the protocol's final schedule: 1.115 - 1.146 against 1.144 - 1.175 ms in four same-box pairs, profiles/r4_ab_pair_protocol_256k_128k.txt;
#define FFT_QUAD_SLOTS18 3  // n = 2^17: +-0.3 %, stays on the team counter)
#endif
#ifndef FFT_QUAD_SLOTS17
#define FFT_QUAD_SLOTS17 1
#endif
#include "fft_engine.h"
#include "fft_plans_ext.h"

#define HIP_TRY(call, fail)                                                                          \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            fprintf(stderr, "HIP error at %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            fail;                                                                                    \
        }                                                                                            \
    } while (0)

#ifndef FFT_TEAM_PAIR_DEFAULT
#define FFT_TEAM_PAIR_DEFAULT 1
#endif

namespace {

// resource counters (fft_gpu_debug_counters_hip): what a "cheap" repeated call must not move
long long g_count_allocs = 0, g_count_streams = 0;

// ---------------------------------------------------------------- runtime policy
struct HipRT {
    hipStream_t stream = nullptr;
    int lds_limit = 64 * 1024;
    int cus = 256;
    bool gfx950 = false;
    ffteng::EnginePolicy policy;
    std::set<const void*> configured;

    void* dmalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
            fprintf(stderr, "fft_hip: hipMalloc(%zu) failed\n", bytes);
            (void)hipGetLastError();
            return nullptr;
        }
        __atomic_fetch_add(&g_count_allocs, 1ll, __ATOMIC_RELAXED);
        return p;
    }
    void dfree(void* p) {
        if (p) (void)hipFree(p);
    }
    void h2d(void* dst, const void* src, size_t bytes) { (void)hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); }
    int max_lds_bytes() { return lds_limit; }
    int num_cus() { return cus; }
    std::map<std::tuple<const void*, int, size_t>, int> occ_cache;  // the answer depends on block size and LDS bytes too
    template <class K>
    int max_blocks_per_cu(K kernel, int threads, size_t smem) {
        const void* key = reinterpret_cast<const void*>(kernel);
        const auto ckey = std::make_tuple(key, threads, smem);
        auto it = occ_cache.find(ckey);
        if (it != occ_cache.end()) return it->second;
        if (smem > 48 * 1024 && !configured.count(key)) {
            (void)hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit);
            (void)hipGetLastError();
            configured.insert(key);
        }
        int n = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, key, threads, smem) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 1;
        }
        if (n > 8) n = 8;
        occ_cache[ckey] = n;
        return n;
    }

    // per-pass HIP-event profiling (fft_gpu_profile_passes_hip): an event after every pass launch
    bool profiling = false;
    std::vector<std::pair<int, hipEvent_t>> marks;
    void mark(int pass_index) {
        if (!profiling) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, stream);
        marks.push_back(std::make_pair(pass_index, e));
    }

    void memset_async(void* p, int v, size_t bytes) { (void)hipMemsetAsync(p, v, bytes, stream); }
    void d2d_async(void* dst, const void* src, size_t bytes) { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream); }
    // Team kernel (fft_team.h): one 512-thread workgroup per CU, 2^log2seats of them on each of n_xcc XCDs.  gfx950 has
    // 32 CUs per XCD in every partition mode (SPX 256 CUs = 8 XCDs, DPX 128 = 4, QPX 64 = 2, CPX 32 = 1), so the XCD
    // count follows from the device's CU count.  The kernel verifies the placement itself (HW_REG_XCC_ID, team_form);
    // this only says which shape to ask for.
    bool team_geometry(int& log2seats, int& n_xcc, int& nthreads) {
        if (!gfx950 || cus < 32 || (cus % 32) != 0 || cus / 32 > 16) return false;
        log2seats = 5;
        n_xcc = cus / 32;
        nthreads = 512;
        return true;
    }
    // sizes where the team kernel measured faster than the multi-pass schedule (tools/team_sweep.py at 4 GiB
    // per execute): fp32 2^16..2^20 (+14, +15, +24, +24, +15..19 %), fp64 2^15..2^19 (+28, +25, +35, +25, +16 %)
    bool team_default_on(int elem_bytes, int log2n) {
        return elem_bytes == 8 ? (log2n >= 15 && log2n <= 20) : (log2n >= 14 && log2n <= 19);
    }
    // the column step on 128-byte row segments (fft_team.h ASPLIT), instantiated for fp32 n = 2^20 where the plain
    // tiles have 64-byte ones.  Measured 125 vs 137 Gpoint/s (the joined halves concentrate twiddles and hand-over in
    // every second tile): an experiment, on only with FFT_HIP_TEAM_ASPLIT=1
    bool team_asplit(int elem_bytes, int log2n) {
        static const int on = FFT_EXP_ENV("FFT_HIP_TEAM_ASPLIT") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_ASPLIT")) : 0;
        return on && elem_bytes == 8 && log2n == 20;
    }
    // team_defer_kernel (fft_team_defer.h: the last row phase fills the turn of the next transform) instead of
    // team_fft_kernel: +2..8 % at every built size (profiles/r1e_team_variant_sweep.txt).  FFT_HIP_TEAM_DEFER=0 runs the
    // plain kernel.
    bool team_defer(int /*elem_bytes*/, int /*log2n*/) {
        static const int on = FFT_EXP_ENV("FFT_HIP_TEAM_DEFER") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_DEFER")) : 1;
        return on != 0;
    }
    // team_defer_kernel PAIR (fft_team_defer.h): 128-byte result segments where the row tiles have only 8 rows
    bool team_pair(int elem_bytes, int log2n) {
        static const int on = FFT_EXP_ENV("FFT_HIP_TEAM_PAIR") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_PAIR")) : FFT_TEAM_PAIR_DEFAULT;
        return on && elem_bytes == 8 && (log2n == 19 || log2n == 20);
    }
    // team_quad_kernel (fft_team_quad.h) instead of the tile-by-tile team kernels: fp32 n = 2^15 .. 2^20.  FFT_HIP_TEAM_QUAD (the
    // experiments build): 0 never, 1 every built size (default), or a bit mask: bit (log2n - 14) = that size only
    bool team_quad(int elem_bytes, int log2n) {
        static const int on = FFT_EXP_ENV("FFT_HIP_TEAM_QUAD") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_QUAD")) : 1;
        if (elem_bytes == 8 ? (log2n < 15 || log2n > 20) : (log2n < 14 || log2n > 16)) return false;
        if (on <= 1) return on == 1;
        return (on >> (log2n - 14)) & 1;
    }
    // the exchange protocol of team_quad_kernel where several are built (n = 2^17 ... 2^20): window slots 1 / 2 with the team's arrival counter,
    // 3 = one image per seat with the pair protocol (the experiments build: FFT_HIP_QUAD_SLOTS=1 / 2 / 3)
    int team_quad_slots(int log2n) {
        static const int v = FFT_EXP_ENV("FFT_HIP_QUAD_SLOTS") ? atoi(FFT_EXP_ENV("FFT_HIP_QUAD_SLOTS")) : 0;
        return v ? v : (log2n == 20 ? FFT_QUAD_SLOTS20 : log2n == 19 ? FFT_QUAD_SLOTS19 : log2n == 18 ? FFT_QUAD_SLOTS18 : log2n == 17 ? FFT_QUAD_SLOTS17 : 1);
    }
    // wide_row_kernel (fft_wide_row.h): single-pass n = 8192 and 16384 fp32.  FFT_HIP_WIDE=0 (the experiments build): the two-pass schedule
    bool wide_rows(int elem_bytes, int log2n) {
        static const int on = FFT_EXP_ENV("FFT_HIP_WIDE") ? atoi(FFT_EXP_ENV("FFT_HIP_WIDE")) : 1;
        return on && gfx950 && ((elem_bytes == 8 && (log2n == 13 || log2n == 14)) || (elem_bytes == 16 && log2n == 13));
    }
    bool team_alll2(int, int) {
        static const int on = FFT_EXP_ENV("FFT_HIP_TEAM_ALLL2") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_ALLL2")) : 0;
        return on != 0;
    }
    // NODEFER (two live windows per team instead of three): measured with the paired kernel (profiles/r2_pmc_variants.txt)
    // n = 2^20: window refetch 3.8 -> 1.2 GB per 512 transforms (L2 hit rate 51 -> 67 %), +1..2.5 % throughput; n = 2^19
    // (teams of 16: the turn costs more than the hits return): -1 %
    bool team_nodefer(int elem_bytes, int log2n) {
        static const int on = FFT_EXP_ENV("FFT_HIP_TEAM_NODEFER") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_NODEFER")) : -1;
        return on >= 0 ? on != 0 : (elem_bytes == 8 && log2n == 20);
    }
    // 100 MHz wall clock.  Formation: 1 ms -- on a device shared with somebody else's kernel the launch gives up at once
    // (nothing touched) and the multi-pass plan queued behind it runs.  Team waits: 2 s, a deadlock breaker only: every
    // member of a formed team is resident and running, so a wait ends when the slowest member gets there.
    long long team_form_timeout_ticks() { return 100000ll; }
    long long team_timeout_ticks() { return 200000000ll; }
    template <class K, class... A>
    void launch_coresident(K kernel, long long grid, int block, size_t smem, A... args) {
        launch(kernel, grid, block, smem, args...);  // LDS footprint > 80 KiB: one workgroup per CU, grid == CUs
    }

    template <class K, class... A>
    void launch(K kernel, long long grid, int block, size_t smem, A... args) {
        const void* key = reinterpret_cast<const void*>(kernel);
        if (smem > 48 * 1024 && !configured.count(key)) {
            (void)hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit);
            (void)hipGetLastError();
            configured.insert(key);
        }
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3((unsigned)block), smem, stream, args...);
    }
};

// ---------------------------------------------------------------- global state
pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
int g_initialized = 0;
int g_device = 0;
char g_device_name[256] = "No GPU";
ffteng::EnginePolicy g_policy;  // read from the environment once, at fft_gpu_init_hip

// Per-device facts the planner needs, read from THAT device's properties the first time a plan or buffer is made on it
// (an 8-GPU process sets each device in turn; nothing is cached from the first device for the others).
struct DeviceInfo {
    bool valid = false;
    int lds_limit = 64 * 1024;
    int cus = 0;
    bool gfx950 = false;
    char name[256];
};
enum { MAX_DEVICES = 64 };
DeviceInfo g_devinfo[MAX_DEVICES];

int probe_device_count() {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

// call with g_lock held or from a context that tolerates a benign double fill (the facts are immutable)
const DeviceInfo* device_info(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return nullptr;
    DeviceInfo& d = g_devinfo[dev];
    if (d.valid) return &d;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        fprintf(stderr, "fft_hip: hipGetDeviceProperties(%d) failed\n", dev);
        (void)hipGetLastError();
        return nullptr;
    }
    DeviceInfo t;
    t.gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    // gfx950: 160 KiB of LDS per workgroup (MI355X_MICROARCH: LDS per CU); other parts: what the runtime reports
    size_t lds = prop.sharedMemPerBlock;
    if (t.gfx950 && lds < 160 * 1024) lds = 160 * 1024;
    if (const char* e = FFT_EXP_ENV("FFT_HIP_LDS_BYTES")) {
        long v = atol(e);
        if (v >= 16384) lds = (size_t)v;
    }
    t.lds_limit = (int)lds;
    t.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    // (prop.name is empty on hosts without the amdgpu.ids marketing-name table: say what is known instead of nothing)
    snprintf(t.name, sizeof(t.name), "%s (%s, %d CUs)", prop.name[0] ? prop.name : (t.gfx950 ? "AMD Instinct MI355X-class GPU" : "AMD GPU"),
             prop.gcnArchName, prop.multiProcessorCount);
    t.valid = true;
    d = t;
    return &d;
}

}  // namespace

// ---------------------------------------------------------------- handles
struct fft_gpu_memory {
    void* device_ptr;
    size_t size;  // bytes
    int device;
};

struct fft_gpu_plan {
    int n = 0, batch = 0, dir = -1, prec = 0, algo = 0, device = 0;
    bool pow2 = true;
    HipRT rt;
    hipStream_t own_stream = nullptr;
    ffteng::Pow2Plan<float, HipRT>* p32 = nullptr;
    ffteng::Pow2Plan<double, HipRT>* p64 = nullptr;
    ffteng::BluesteinPlan<float, HipRT>* b32 = nullptr;
    ffteng::BluesteinPlan<double, HipRT>* b64 = nullptr;
    // plans built on the batched engine (fft_plans_ext.h); kind says which member is live
    int kind = 0;  // 0 complex 1D, 1 complex 2D (n = rows * cols, batch = matrices), 2 r2c, 3 c2r, 4 fused consumer
    int rows = 0, cols = 0;
    ffteng::Plan2D<float, HipRT>* d32 = nullptr;
    ffteng::Plan2D<double, HipRT>* d64 = nullptr;
    ffteng::RealPlan<float, HipRT>* r32 = nullptr;
    ffteng::RealPlan<double, HipRT>* r64 = nullptr;
    ffteng::FusedPlan<float, HipRT>* f32 = nullptr;
    ffteng::FusedPlan<double, HipRT>* f64 = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};
enum { PLAN_C2C = 0, PLAN_2D = 1, PLAN_R2C = 2, PLAN_C2R = 3, PLAN_FUSED = 4 };

namespace {

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            if (hipSetDevice(dev) == hipSuccess) switched = true;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// After a stream sync: what did the last team kernel (fft_team.h) report?  0 done by the team kernel, 1 teams could
// not be formed and the two-pass fallback did the work, 2 a team barrier timed out (results invalid), -1 no team
// kernel in this plan / nothing launched.  Three fallbacks in a row pause the team kernel for the plan's next 64 (then 128 ... 1024)
// executes; the attempt after the pause gives up forming after a tenth of the usual bound.
template <class Core>
int team_status_of(Core* core) {
    if (!core) return -1;
    if (!core->team_pending) return core->team_last_status;  // nothing launched since the last look
    if (!core->team.ctl || !core->team.sticky) return -1;
    unsigned st = 0, sticky[2] = {0, 0};
    if (hipMemcpy(&st, core->team.ctl + fftk::TEAM_CTL_STATUS, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(sticky, core->team.sticky, sizeof(sticky), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    const int launches = core->team_pending;
    core->team_pending = 0;
    // the status word is the LAST launch's; the sticky counters cover every launch since the last look (each execute
    // zeroes the control block, so an earlier execute's TIMEOUT is only visible here)
    if (sticky[fftk::TEAM_STICKY_TIMEOUTS]) st = fftk::TEAM_STATUS_TIMEOUT;
    if (sticky[0] || sticky[1]) (void)hipMemset(core->team.sticky, 0, sizeof(sticky));
    core->team_last_status = (int)st;
    if (st == fftk::TEAM_STATUS_TIMEOUT) {
        // the launch has ended (every wait in the kernel is bounded): replay what it was asked to do on the multi-pass schedule
        const int lost = core->recover_after_timeout(true);
        if (hipStreamSynchronize(core->rt->stream) != hipSuccess) (void)hipGetLastError();
        if (lost)
            fprintf(stderr, "fft_hip: a team kernel barrier timed out; %d execute(s) since the last sync could not be repeated (their input "
                            "is gone, overwritten by a later execute, or not the caller's to keep) and hold invalid data; this plan continues "
                            "with the multi-pass schedule\n", lost);
        else
            fprintf(stderr, "fft_hip: a team kernel barrier timed out; the executes since the last sync were repeated on the multi-pass "
                            "schedule, which this plan keeps from now on\n");
        if (!lost) st = fftk::TEAM_STATUS_NO_TEAMS;  // the caller sees "done by the fallback"
        core->team_last_status = (int)st;
    } else if ((int)sticky[fftk::TEAM_STICKY_FALLBACKS] >= launches && launches > 0) {  // every launch fell back
        core->team_fallbacks += launches;
        if (core->team_fallbacks >= 3) {
            // a shared or partitioned device: stop paying the formation timeout on every execute -- but try again later (round-3 review,
            // weak point 5: the kernel used to be retired for good)
            core->team_suspend = 64 << (core->team_suspensions < 4 ? core->team_suspensions : 4);
            core->team_suspensions++;
            core->team_fallbacks = 0;
            fprintf(stderr, "fft_hip: the team kernel could not form its XCD teams three times in a row; this plan runs its next %d "
                            "executes on the multi-pass schedule before it tries again\n", core->team_suspend);
        }
    } else {
        core->team_fallbacks = 0;
        if (st == fftk::TEAM_STATUS_OK && launches > 0) {
            core->team_seen_ok();  // (in-place executes are no longer staged)
            core->team_suspensions = 0;
        }
    }
    if (st != fftk::TEAM_STATUS_TIMEOUT) {
        core->team_log.clear();
        core->team_log_dropped = 0;
        core->team_stage_used = 0;
    }
    return (int)st;
}

// every power-of-two core of a plan, whatever it is built of (Bluestein, 2D, real and fused plans run their transforms on cores of
// their own: a team kernel's timeout there must reach fft_gpu_plan_sync like any other)
template <class F>
void for_each_core(fft_gpu_plan* p, F&& f) {
    auto any = [&](auto* a) {
        if (!a) return;
        if (a->p2) f(a->p2);
        if (a->bl) f(&a->bl->core);
    };
    if (p->p32) f(p->p32);
    if (p->p64) f(p->p64);
    if (p->b32) f(&p->b32->core);
    if (p->b64) f(&p->b64->core);
    auto two = [&](auto* d) {
        if (!d) return;
        any(&d->rowp);
        if (d->colp) f(d->colp);
        any(d->colt);
    };
    two(p->d32); two(p->d64);
    if (p->r32) any(&p->r32->core);
    if (p->r64) any(&p->r64->core);
    if (p->f32) f(&p->f32->core);
    if (p->f64) f(&p->f64->core);
}

// the worst of the plan's cores: TIMEOUT (2) > NO_TEAMS (1) > OK (0) > no team kernel / nothing launched (-1)
int plan_team_status(fft_gpu_plan* p) {
    if (!p) return -1;
    int worst = -1;
    for_each_core(p, [&](auto* c) {
        const int st = team_status_of(c);
        if (st > worst) worst = st;
    });
    return worst;
}

// bytes one execute reads from its input and writes to its output buffer
void plan_io_bytes(const fft_gpu_plan* p, size_t* in_bytes, size_t* out_bytes) {
    const size_t csz = p->prec == FFT_PREC_F32 ? sizeof(complex32_t) : sizeof(complex_t);
    const size_t rsz = csz / 2;
    const size_t nb = (size_t)p->batch, n = (size_t)p->n;
    switch (p->kind) {
        case PLAN_R2C: *in_bytes = nb * n * rsz; *out_bytes = nb * (n / 2 + 1) * csz; break;
        case PLAN_C2R: *in_bytes = nb * (n / 2 + 1) * csz; *out_bytes = nb * n * rsz; break;
        default: *in_bytes = *out_bytes = nb * n * csz; break;
    }
}

// nb > 0: only the first nb transforms of the plan's batch (1D complex plans; the host-pointer pipeline's last group)
int plan_enqueue(fft_gpu_plan* p, const void* d_in, void* d_out, int nb = 0) {
    if (!p || !d_in || !d_out || nb > p->batch) return -1;
    DeviceGuard guard(p->device);
    const bool inv = p->dir > 0;
    if (nb > 0 && nb < p->batch) {
        if (p->p32) p->p32->execute((const fftk::cpx<float>*)d_in, (fftk::cpx<float>*)d_out, nb, inv);
        else if (p->p64) p->p64->execute((const fftk::cpx<double>*)d_in, (fftk::cpx<double>*)d_out, nb, inv);
        else if (p->b32) p->b32->execute((const fftk::cpx<float>*)d_in, (fftk::cpx<float>*)d_out, nb);
        else if (p->b64) p->b64->execute((const fftk::cpx<double>*)d_in, (fftk::cpx<double>*)d_out, nb);
        else return -1;
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) {
            fprintf(stderr, "fft_hip: kernel launch failed: %s\n", hipGetErrorString(e2));
            return -1;
        }
        return 0;
    }
    if (p->d32) p->d32->execute((const fftk::cpx<float>*)d_in, (fftk::cpx<float>*)d_out, p->batch);
    else if (p->d64) p->d64->execute((const fftk::cpx<double>*)d_in, (fftk::cpx<double>*)d_out, p->batch);
    else if (p->r32 && p->kind == PLAN_R2C) p->r32->execute_r2c((const float*)d_in, (fftk::cpx<float>*)d_out, p->batch);
    else if (p->r64 && p->kind == PLAN_R2C) p->r64->execute_r2c((const double*)d_in, (fftk::cpx<double>*)d_out, p->batch);
    else if (p->r32) p->r32->execute_c2r((const fftk::cpx<float>*)d_in, (float*)d_out, p->batch);
    else if (p->r64) p->r64->execute_c2r((const fftk::cpx<double>*)d_in, (double*)d_out, p->batch);
    else if (p->f32 || p->f64) return -1;  // fused consumers take two inputs: fft_gpu_execute_fused_hip
    else if (p->p32) p->p32->execute((const fftk::cpx<float>*)d_in, (fftk::cpx<float>*)d_out, p->batch, inv);
    else if (p->p64) p->p64->execute((const fftk::cpx<double>*)d_in, (fftk::cpx<double>*)d_out, p->batch, inv);
    else if (p->b32) p->b32->execute((const fftk::cpx<float>*)d_in, (fftk::cpx<float>*)d_out, p->batch);
    else if (p->b64) p->b64->execute((const fftk::cpx<double>*)d_in, (fftk::cpx<double>*)d_out, p->batch);
    else return -1;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "fft_hip: kernel launch failed: %s\n", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

}  // namespace

extern "C" {

// ================================================================ part 1: backend set

int fft_gpu_available_hip(void) { return probe_device_count() > 0 ? 1 : 0; }

int fft_gpu_init_hip(void) {
    pthread_mutex_lock(&g_lock);
    if (g_initialized) {
        pthread_mutex_unlock(&g_lock);
        return 0;
    }
    int count = probe_device_count();
    if (count <= 0) {
        fprintf(stderr, "fft_hip: no HIP device found\n");
        pthread_mutex_unlock(&g_lock);
        return -1;
    }
    // Keep the device the process already selected (one process per GPU under torch.distributed);
    // the reference picks the device with most multiprocessors (fft_cuda.cu:64-79) -- on a
    // homogeneous MI355X node every device ties, so the current one is that choice.
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const DeviceInfo* di = device_info(dev);
    if (!di || hipSetDevice(dev) != hipSuccess) {
        (void)hipGetLastError();
        pthread_mutex_unlock(&g_lock);
        return -1;
    }
    g_device = dev;
    snprintf(g_device_name, sizeof(g_device_name), "%s", di->name);
    // the production policy knobs, read here and nowhere else (no environment access at plan or execute time)
    g_policy = ffteng::EnginePolicy();
    if (const char* e = getenv("FFT_HIP_TEAM")) g_policy.team_mode = atoi(e);
    if (const char* e = getenv("FFT_HIP_TEAM_MIN_BATCH")) g_policy.team_min_batch = atoi(e);
    if (const char* e = getenv("FFT_HIP_CHUNK_MB")) g_policy.chunk_mb = atoll(e);
    g_initialized = 1;
    pthread_mutex_unlock(&g_lock);
    return 0;
}

void fft_gpu_cleanup_hip(void) {
    pthread_mutex_lock(&g_lock);
    if (g_initialized) {
        (void)hipDeviceSynchronize();
        g_initialized = 0;
        snprintf(g_device_name, sizeof(g_device_name), "No GPU");
    }
    pthread_mutex_unlock(&g_lock);
}

fft_gpu_memory_t fft_gpu_alloc_bytes_hip(size_t bytes) {
    if (!g_initialized) return NULL;
    fft_gpu_memory_t mem = (fft_gpu_memory_t)malloc(sizeof(struct fft_gpu_memory));
    if (!mem) return NULL;
    int dev = g_device;
    (void)hipGetDevice(&dev);
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
        fprintf(stderr, "fft_hip: device allocation of %zu bytes failed\n", bytes);
        (void)hipGetLastError();
        free(mem);
        return NULL;
    }
    __atomic_fetch_add(&g_count_allocs, 1ll, __ATOMIC_RELAXED);
    mem->device_ptr = p;
    mem->size = bytes;
    mem->device = dev;
    return mem;
}

fft_gpu_memory_t fft_gpu_alloc_hip(size_t n_complex) { return fft_gpu_alloc_bytes_hip(n_complex * sizeof(complex_t)); }

void fft_gpu_free_hip(fft_gpu_memory_t mem) {
    if (!mem) return;
    if (mem->device_ptr) (void)hipFree(mem->device_ptr);
    free(mem);
}

int fft_gpu_copy_h2d_bytes_hip(fft_gpu_memory_t dst, const void* src, size_t bytes) {
    if (!dst || !src) return -1;
    if (bytes > dst->size) {
        fprintf(stderr, "fft_hip: h2d copy of %zu bytes exceeds the %zu-byte buffer\n", bytes, dst->size);
        return -1;
    }
    if (hipMemcpy(dst->device_ptr, src, bytes, hipMemcpyHostToDevice) == hipSuccess) return 0;
    // The runtime rejects a copy that runs past a page-locked range starting at `src` (another plan registered a SHORTER
    // prefix of this array): go through a bounce buffer of our own instead of failing with stale data on the device.
    (void)hipGetLastError();
    void* tmp = malloc(bytes);
    if (!tmp) return -1;
    memcpy(tmp, src, bytes);
    const hipError_t e = hipMemcpy(dst->device_ptr, tmp, bytes, hipMemcpyHostToDevice);
    free(tmp);
    if (e != hipSuccess) {
        fprintf(stderr, "fft_hip: h2d copy of %zu bytes failed: %s\n", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return -1;
    }
    return 0;
}

int fft_gpu_copy_d2h_bytes_hip(void* dst, fft_gpu_memory_t src, size_t bytes) {
    if (!dst || !src) return -1;
    if (bytes > src->size) {
        fprintf(stderr, "fft_hip: d2h copy of %zu bytes exceeds the %zu-byte buffer\n", bytes, src->size);
        return -1;
    }
    if (hipMemcpy(dst, src->device_ptr, bytes, hipMemcpyDeviceToHost) == hipSuccess) return 0;
    (void)hipGetLastError();  // see fft_gpu_copy_h2d_bytes_hip: a shorter page-locked range at `dst`
    void* tmp = malloc(bytes);
    if (!tmp) return -1;
    const hipError_t e = hipMemcpy(tmp, src->device_ptr, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) memcpy(dst, tmp, bytes);
    free(tmp);
    if (e != hipSuccess) {
        fprintf(stderr, "fft_hip: d2h copy of %zu bytes failed: %s\n", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return -1;
    }
    return 0;
}

void fft_gpu_copy_h2d_hip(fft_gpu_memory_t dst, const complex_t* src, size_t n) {
    (void)fft_gpu_copy_h2d_bytes_hip(dst, src, n * sizeof(complex_t));
}

void fft_gpu_copy_d2h_hip(complex_t* dst, fft_gpu_memory_t src, size_t n) {
    (void)fft_gpu_copy_d2h_bytes_hip(dst, src, n * sizeof(complex_t));
}

void* fft_gpu_memory_ptr_hip(fft_gpu_memory_t mem) { return mem ? mem->device_ptr : NULL; }
size_t fft_gpu_memory_bytes_hip(fft_gpu_memory_t mem) { return mem ? mem->size : 0; }

fft_gpu_plan_t fft_gpu_plan_1d_ex_hip(int n, int batch, fft_direction dir, fft_precision_t prec, fft_gpu_algo_t algo) {
    if (!g_initialized) {
        fprintf(stderr, "fft_hip: plan requested before fft_gpu_init\n");
        return NULL;
    }
    if (n <= 0 || batch <= 0 || (prec != FFT_PREC_F32 && prec != FFT_PREC_F64) || (int)algo < 0 || (int)algo > 6) {
        fprintf(stderr, "fft_hip: invalid plan arguments (n=%d batch=%d prec=%d algo=%d)\n", n, batch, (int)prec, (int)algo);
        return NULL;
    }
    fft_gpu_plan* p = new (std::nothrow) fft_gpu_plan();
    if (!p) return NULL;
    p->n = n;
    p->batch = batch;
    p->dir = ((int)dir < 0) ? -1 : 1;
    p->prec = (int)prec;
    p->algo = (int)algo;
    p->pow2 = (n & (n - 1)) == 0 && algo != FFT_GPU_ALGO_BLUESTEIN;
    if (algo == FFT_GPU_ALGO_BLUESTEIN) algo = FFT_GPU_ALGO_AUTO;
    int dev = g_device;
    (void)hipGetDevice(&dev);
    p->device = dev;
    if (hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        delete p;
        return NULL;
    }
    __atomic_fetch_add(&g_count_streams, 1ll, __ATOMIC_RELAXED);  // (fft_gpu_debug_counters_hip: every stream the backend creates)
    const DeviceInfo* di = device_info(dev);
    if (!di) {
        fft_gpu_destroy_plan_hip(p);
        return NULL;
    }
    p->rt.stream = p->own_stream;
    p->rt.lds_limit = di->lds_limit;
    p->rt.cus = di->cus;
    p->rt.gfx950 = di->gfx950;
    p->rt.policy = g_policy;
    bool ok = false;
    const int log2n = ffteng::ilog2(n);
    if (p->pow2) {
        if (prec == FFT_PREC_F32) {
            p->p32 = new (std::nothrow) ffteng::Pow2Plan<float, HipRT>();
            ok = p->p32 && p->p32->build(&p->rt, log2n, (int)algo, batch);
        } else {
            p->p64 = new (std::nothrow) ffteng::Pow2Plan<double, HipRT>();
            ok = p->p64 && p->p64->build(&p->rt, log2n, (int)algo, batch);
        }
    } else {
        if (n > (1 << 29)) {
            fprintf(stderr, "fft_hip: Bluestein length %d too large\n", n);
        } else if (prec == FFT_PREC_F32) {
            p->b32 = new (std::nothrow) ffteng::BluesteinPlan<float, HipRT>();
            ok = p->b32 && p->b32->build(&p->rt, n, p->dir, (int)algo, batch);
        } else {
            p->b64 = new (std::nothrow) ffteng::BluesteinPlan<double, HipRT>();
            ok = p->b64 && p->b64->build(&p->rt, n, p->dir, (int)algo, batch);
        }
    }
    if (ok && !p->pow2) for_each_core(p, [](auto* c) { c->team_replay = false; });  // (a core of a composite plan transforms the plan's own intermediates)
    if (ok) ok = (hipStreamSynchronize(p->own_stream) == hipSuccess);
    if (!ok) {
        fprintf(stderr, "fft_hip: could not build a plan for n=%d batch=%d\n", n, batch);
        (void)hipGetLastError();
        fft_gpu_destroy_plan_hip(p);
        return NULL;
    }
    return p;
}

fft_gpu_plan_t fft_gpu_plan_1d_hip(int n, int batch, fft_direction dir) {
    return fft_gpu_plan_1d_ex_hip(n, batch, dir, FFT_PREC_F64, FFT_GPU_ALGO_AUTO);
}

void fft_gpu_destroy_plan_hip(fft_gpu_plan_t p) {
    if (!p) return;
    {
        DeviceGuard guard(p->device);
        if (p->rt.stream) (void)hipStreamSynchronize(p->rt.stream);
        delete p->p32;
        delete p->p64;
        delete p->b32;
        delete p->b64;
        delete p->d32;
        delete p->d64;
        delete p->r32;
        delete p->r64;
        delete p->f32;
        delete p->f64;
        if (p->ev0) (void)hipEventDestroy(p->ev0);
        if (p->ev1) (void)hipEventDestroy(p->ev1);
        if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    }
    delete p;
}

// ---- plans built on the batched engine (fft_plans_ext.h): common shell
static fft_gpu_plan* new_plan_shell(int n, int batch, int dir, fft_precision_t prec, int kind) {
    if (!g_initialized) {
        fprintf(stderr, "fft_hip: plan requested before fft_gpu_init\n");
        return NULL;
    }
    if (prec != FFT_PREC_F32 && prec != FFT_PREC_F64) return NULL;
    fft_gpu_plan* p = new (std::nothrow) fft_gpu_plan();
    if (!p) return NULL;
    p->n = n; p->batch = batch; p->dir = dir < 0 ? -1 : 1; p->prec = (int)prec; p->kind = kind; p->pow2 = false;
    int dev = g_device;
    (void)hipGetDevice(&dev);
    p->device = dev;
    const DeviceInfo* di = device_info(dev);
    if (!di || hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        delete p;
        return NULL;
    }
    __atomic_fetch_add(&g_count_streams, 1ll, __ATOMIC_RELAXED);
    p->rt.stream = p->own_stream;
    p->rt.lds_limit = di->lds_limit;
    p->rt.cus = di->cus;
    p->rt.gfx950 = di->gfx950;
    p->rt.policy = g_policy;
    return p;
}
static fft_gpu_plan_t finish_plan(fft_gpu_plan* p, bool ok, const char* what) {
    if (ok) for_each_core(p, [](auto* c) { c->team_replay = false; });  // composite plans: a team timeout in a core is reported, never replayed
    if (ok) ok = (hipStreamSynchronize(p->own_stream) == hipSuccess);
    if (!ok) {
        fprintf(stderr, "fft_hip: could not build a %s plan\n", what);
        (void)hipGetLastError();
        fft_gpu_destroy_plan_hip(p);
        return NULL;
    }
    return p;
}

// 2D complex transform of `n_matrices` row-major rows x cols matrices (replaces the stub gpu/fft_gpu.c:377-385)
fft_gpu_plan_t fft_gpu_plan_2d_ex_hip(int rows, int cols, int n_matrices, fft_direction dir, fft_precision_t prec) {
    if (rows <= 0 || cols <= 0 || n_matrices <= 0 || (long long)rows * cols > (1ll << 30)) {
        fprintf(stderr, "fft_hip: invalid 2D plan arguments (rows=%d cols=%d matrices=%d)\n", rows, cols, n_matrices);
        return NULL;
    }
    fft_gpu_plan* p = new_plan_shell(rows * cols, n_matrices, (int)dir, prec, PLAN_2D);
    if (!p) return NULL;
    p->rows = rows; p->cols = cols;
    bool ok;
    if (prec == FFT_PREC_F32) {
        p->d32 = new (std::nothrow) ffteng::Plan2D<float, HipRT>();
        ok = p->d32 && p->d32->build(&p->rt, rows, cols, p->dir, n_matrices);
    } else {
        p->d64 = new (std::nothrow) ffteng::Plan2D<double, HipRT>();
        ok = p->d64 && p->d64->build(&p->rt, rows, cols, p->dir, n_matrices);
    }
    return finish_plan(p, ok, "2D");
}
fft_gpu_plan_t fft_gpu_plan_2d_hip(int rows, int cols, fft_direction dir) { return fft_gpu_plan_2d_ex_hip(rows, cols, 1, dir, FFT_PREC_F64); }

// real-input forward / real-output inverse 1D transforms, n/2 + 1 bins (reference stubs algorithms/auto/fft_auto.c:391-409)
static fft_gpu_plan_t plan_real(int n, int batch, fft_precision_t prec, bool r2c) {
    if (n <= 0 || batch <= 0) {
        fprintf(stderr, "fft_hip: invalid real-transform plan arguments (n=%d batch=%d)\n", n, batch);
        return NULL;
    }
    fft_gpu_plan* p = new_plan_shell(n, batch, r2c ? -1 : 1, prec, r2c ? PLAN_R2C : PLAN_C2R);
    if (!p) return NULL;
    bool ok;
    if (prec == FFT_PREC_F32) {
        p->r32 = new (std::nothrow) ffteng::RealPlan<float, HipRT>();
        ok = p->r32 && p->r32->build(&p->rt, n, r2c, batch);
    } else {
        p->r64 = new (std::nothrow) ffteng::RealPlan<double, HipRT>();
        ok = p->r64 && p->r64->build(&p->rt, n, r2c, batch);
    }
    return finish_plan(p, ok, r2c ? "r2c" : "c2r");
}
fft_gpu_plan_t fft_gpu_plan_r2c_1d_hip(int n, int batch, fft_precision_t prec) { return plan_real(n, batch, prec, true); }
fft_gpu_plan_t fft_gpu_plan_c2r_1d_hip(int n, int batch, fft_precision_t prec) { return plan_real(n, batch, prec, false); }

// fused consumers (fft_plans_ext.h FusedPlan); h_host: the nh kernel samples of a convolution (host memory, element
// type of `prec`), ignored otherwise
fft_gpu_plan_t fft_gpu_plan_fused_hip(fft_gpu_fused_t kind, int nx, int nh, const void* h_host, int batch, fft_precision_t prec) {
    if (nx <= 0 || batch <= 0 || (int)kind < 0 || (int)kind > 4) {
        fprintf(stderr, "fft_hip: invalid fused plan arguments (kind=%d nx=%d batch=%d)\n", (int)kind, nx, batch);
        return NULL;
    }
    fft_gpu_plan* p = new_plan_shell(nx, batch, -1, prec, PLAN_FUSED);
    if (!p) return NULL;
    bool ok;
    if (prec == FFT_PREC_F32) {
        p->f32 = new (std::nothrow) ffteng::FusedPlan<float, HipRT>();
        ok = p->f32 && p->f32->build(&p->rt, (int)kind, nx, nh, (const fftk::cpx<float>*)h_host, batch);
    } else {
        p->f64 = new (std::nothrow) ffteng::FusedPlan<double, HipRT>();
        ok = p->f64 && p->f64->build(&p->rt, (int)kind, nx, nh, (const fftk::cpx<double>*)h_host, batch);
    }
    return finish_plan(p, ok, "fused");
}
// elements per output row of a fused plan (complex values; FFT_GPU_FUSED_PSD: real values)
int fft_gpu_fused_out_len_hip(fft_gpu_plan_t p) {
    if (!p) return -1;
    if (p->f32) return p->f32->out_len();
    if (p->f64) return p->f64->out_len();
    return -1;
}
// async on the plan's stream.  d_x: [batch][nx]; d_y: second signal (XCORR) or NULL; d_out: [batch][out_len]
int fft_gpu_execute_fused_hip(fft_gpu_plan_t p, const void* d_x, const void* d_y, void* d_out, double sample_rate) {
    if (!p || !d_x || !d_out || (!p->f32 && !p->f64)) return -1;
    DeviceGuard guard(p->device);
    if (p->f32) {
        if (p->f32->kind == ffteng::FUSED_XCORR && !d_y) return -1;
        p->f32->execute((const fftk::cpx<float>*)d_x, (const fftk::cpx<float>*)d_y, d_out, p->batch, (float)sample_rate);
    } else {
        if (p->f64->kind == ffteng::FUSED_XCORR && !d_y) return -1;
        p->f64->execute((const fftk::cpx<double>*)d_x, (const fftk::cpx<double>*)d_y, d_out, p->batch, sample_rate);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fprintf(stderr, "fft_hip: kernel launch failed: %s\n", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

int fft_gpu_plan_set_stream_hip(fft_gpu_plan_t p, void* hip_stream) {
    if (!p) return -1;
    (void)hipStreamSynchronize(p->rt.stream);
    p->rt.stream = hip_stream ? (hipStream_t)hip_stream : p->own_stream;
    return 0;
}

int fft_gpu_execute_ptr_hip(fft_gpu_plan_t p, const void* d_in, void* d_out) { return plan_enqueue(p, d_in, d_out); }

int fft_gpu_plan_sync_hip(fft_gpu_plan_t p) {
    if (!p) return -1;
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->rt.stream), return -1);
    return plan_team_status(p) == (int)fftk::TEAM_STATUS_TIMEOUT ? -1 : 0;
}

// Profiling: every workgroup of the team kernel logs its 100 MHz wall clock at its first `events` timeline events
// into d_trace[workgroup * events + i] (device memory of 256 * events * 8 bytes, owned by the caller).  NULL = off.
int fft_gpu_plan_team_trace_hip(fft_gpu_plan_t p, void* d_trace, int events) {
    if (!p) return -1;
    auto set = [&](auto* core) {
        if (!core || !core->team.ok) return -1;
        core->team.trace = (long long*)d_trace;
        core->team.trace_events = d_trace ? events : 0;
        return 0;
    };
    if (p->p32) return set(p->p32);
    if (p->p64) return set(p->p64);
    return -1;
}

// Process-wide planner policy for plans created AFTER the call (a negative argument keeps the current value):
// team_mode 0 never the team kernel / 1 where it measured faster / 2 every built size; team_min_batch 0 = the measured
// crossover; chunk_mb 0 = default launch-group size.  fft_gpu_init_hip seeds the same three from the environment
// (FFT_HIP_TEAM, FFT_HIP_TEAM_MIN_BATCH, FFT_HIP_CHUNK_MB), once.
int fft_gpu_set_policy_hip(int team_mode, int team_min_batch, int chunk_mb) {
    pthread_mutex_lock(&g_lock);
    if (team_mode >= 0) g_policy.team_mode = team_mode;
    if (team_min_batch >= 0) g_policy.team_min_batch = team_min_batch;
    if (chunk_mb >= 0) g_policy.chunk_mb = chunk_mb;
    pthread_mutex_unlock(&g_lock);
    return 0;
}

int fft_gpu_host_register_hip(void* host_ptr, size_t bytes) {
    if (!host_ptr || !bytes || !g_initialized) return -1;
    if (hipHostRegister(host_ptr, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();  // already registered, or not registrable: copies simply stay pageable
        return -1;
    }
    return 0;
}
int fft_gpu_host_unregister_hip(void* host_ptr) {
    if (!host_ptr) return -1;
    if (hipHostUnregister(host_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return 0;
}
// 1: the address is page-locked host memory known to the runtime (hipHostRegister / hipHostMalloc), 0: it is not
int fft_gpu_host_is_registered_hip(const void* host_ptr) {
    if (!host_ptr) return 0;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host_ptr) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return attr.type == hipMemoryTypeHost ? 1 : 0;
}
// Device copy benchmark on the current device: the best of a few launch shapes of copy16_kernel over `bytes` bytes
// (read + written bytes per second, GB/s); what bench.py quotes as the box's practical ceiling.  -1 on failure.
// mode 0 copy (read + written bytes per second), 1 read only, 2 write only: the best of several launch shapes (1 / 4 / 8
// accesses in flight per thread, 8 or 16 workgroups per CU, with and without the non-temporal hint; copy: also the tile copy in
// the engine's own shape, LDS-DMA in and nt stores out), GB/s; -1 on failure
double fft_gpu_stream_bench_hip(size_t bytes, int iters, int mode) {
    if (!g_initialized || bytes < (1u << 20) || iters <= 0 || mode < 0 || mode > 2) return -1.0;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (a) (void)hipFree(a);
        return -1.0;
    }
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const long long n16 = (long long)(bytes / 16);
    const DeviceInfo* di = device_info(g_device);
    const int cus = di ? di->cus : 256;
    const fftk::vec16<float>* in = (const fftk::vec16<float>*)a;
    fftk::vec16<float>* out = (fftk::vec16<float>*)b;
    double best = -1.0;
    // shapes 0 - 11: grid-stride (1 / 4 / 8 accesses in flight x 8 / 16 workgroups per CU x plain / nt); 12 (copy only): the LDS-DMA tile
    // copy; 13 - 30: tile-wise streams in the single-pass FFT kernels' shape: tiles of 16 / 32 / 64 KiB (4 / 8 / 16 accesses in flight per
    // thread) x 2 / 4 / 8 workgroups per CU x plain / nt
    const int n_shapes = 31;
    static const bool verbose = FFT_EXP_ENV("FFT_HIP_STREAM_VERBOSE") != nullptr;  // (the experiments build: every shape's rate on stderr)
    for (int shape = 0; shape < n_shapes; shape++) {
        if (shape == 12 && mode != 0) continue;
        const int ts = shape - 13;  // tile shapes: (ts % 3) -> workgroups per CU, (ts / 3) % 3 -> tile size, ts / 9 -> nt
        const int tile_u = shape >= 13 ? (4 << ((ts / 3) % 3)) : 8;
        const int per_cu = shape >= 13 ? (2 << (ts % 3)) : (shape / 3) % 2 ? 16 : 8;  // resident 256-thread workgroups per CU
        const bool nt = shape >= 13 ? ts >= 9 : shape >= 6;
        const unsigned grid = (unsigned)(cus * per_cu);
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0, nullptr);
            for (int it = 0; it < iters; it++) {
                if (shape >= 13) {
                    const long long n_tiles = (long long)(bytes / (4096 * (size_t)tile_u));
#define FFT_TILE_LAUNCH2(M, N, U) hipLaunchKernelGGL((fftk::stream_tile_kernel<M, N, U>), dim3(grid), dim3(256), 0, nullptr, in, out, n_tiles)
#define FFT_TILE_LAUNCH1(M, N) do { if (tile_u == 4) FFT_TILE_LAUNCH2(M, N, 4); else if (tile_u == 8) FFT_TILE_LAUNCH2(M, N, 8); else FFT_TILE_LAUNCH2(M, N, 16); } while (0)
#define FFT_TILE_LAUNCH(M) do { if (nt) FFT_TILE_LAUNCH1(M, 1); else FFT_TILE_LAUNCH1(M, 0); } while (0)
                    if (mode == 0) FFT_TILE_LAUNCH(0); else if (mode == 1) FFT_TILE_LAUNCH(1); else FFT_TILE_LAUNCH(2);
#undef FFT_TILE_LAUNCH
#undef FFT_TILE_LAUNCH1
#undef FFT_TILE_LAUNCH2
                    continue;
                }
                if (shape == 12) {
                    const void* key = reinterpret_cast<const void*>(fftk::copy_dma_kernel);
                    (void)hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                    hipLaunchKernelGGL(fftk::copy_dma_kernel, dim3((unsigned)cus), dim3(512), 131072, nullptr, in, out, (long long)(bytes / 65536));
                    continue;
                }
#define FFT_STREAM_LAUNCH(U, M, N) hipLaunchKernelGGL((fftk::stream16_kernel<U, M, N>), dim3(grid), dim3(256), 0, nullptr, in, out, n16)
#define FFT_STREAM_MODE(M)                                                                                        \
    switch (shape % 3) {                                                                                          \
        case 0: if (nt) FFT_STREAM_LAUNCH(1, M, 1); else FFT_STREAM_LAUNCH(1, M, 0); break;                       \
        case 1: if (nt) FFT_STREAM_LAUNCH(4, M, 1); else FFT_STREAM_LAUNCH(4, M, 0); break;                       \
        default: if (nt) FFT_STREAM_LAUNCH(8, M, 1); else FFT_STREAM_LAUNCH(8, M, 0); break;                      \
    }
                if (mode == 0) { FFT_STREAM_MODE(0) } else if (mode == 1) { FFT_STREAM_MODE(1) } else { FFT_STREAM_MODE(2) }
#undef FFT_STREAM_MODE
#undef FFT_STREAM_LAUNCH
            }
            (void)hipEventRecord(e1, nullptr);
            if (hipEventSynchronize(e1) != hipSuccess) break;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1 && ms > 0.f) {
                const double gbs = (mode == 0 ? 2.0 : 1.0) * (double)bytes * iters / (ms * 1e-3) / 1e9;
                if (verbose) fprintf(stderr, "stream bench mode %d shape %2d: %7.1f GB/s\n", mode, shape, gbs);
                if (gbs > best) best = gbs;
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    (void)hipGetLastError();
    return best;
}
double fft_gpu_copy_bench_hip(size_t bytes, int iters) { return fft_gpu_stream_bench_hip(bytes, iters, 0); }
void fft_gpu_debug_counters_hip(long long* device_allocations, long long* streams_created) {
    if (device_allocations) *device_allocations = __atomic_load_n(&g_count_allocs, __ATOMIC_RELAXED);
    if (streams_created) *streams_created = __atomic_load_n(&g_count_streams, __ATOMIC_RELAXED);
}

// FFT_MEASURE at the device level (reference TODO algorithms/auto/fft_auto.c:232-235): instead of the static table of
// sizes and batch crossover (HipRT::team_default_on, TeamDesc::min_batch) time THIS plan's two schedules -- the team
// kernel and the multi-pass plan -- on scratch buffers of its own size and keep the faster.
int fft_gpu_plan_measure_hip(fft_gpu_plan_t p, int iters) {
    if (!p || iters <= 0) return -1;
    DeviceGuard guard(p->device);
    auto measure = [&](auto* core) -> int {
        if (!core || !core->team.tables) return 0;  // one schedule only: nothing to choose
        size_t in_b = 0, out_b = 0;
        plan_io_bytes(p, &in_b, &out_b);
        void* buf = p->rt.dmalloc(in_b);
        if (!buf) return -1;
        (void)hipMemsetAsync(buf, 0, in_b, p->rt.stream);
        core->team_suspend = 0;  // (measure both schedules now, whatever happened before)
        const int saved_min = core->team.min_batch;
        const bool saved_ok = core->team.ok;  // false: switched off before (an option, a timeout)
        float ms[2] = {0.f, 0.f};
        int rc = 0;
        for (int which = 0; which < 2 && rc == 0; which++) {  // 0 multi-pass, 1 team kernel
            if (which == 1 && core->team_disabled) { ms[1] = 1e30f; break; }  // a timeout retired the team kernel for good
            core->team.ok = which == 1;
            core->team.min_batch = 1;
            float warm = 0.f;
            if (fft_gpu_execute_timed_hip(p, buf, buf, 1, &warm) != 0 || fft_gpu_execute_timed_hip(p, buf, buf, iters, &ms[which]) != 0) rc = -1;
            if (which == 1 && plan_team_status(p) != (int)fftk::TEAM_STATUS_OK) ms[1] = 1e30f;  // fell back or timed out: not a candidate
        }
        core->team.min_batch = saved_min;
        p->rt.dfree(buf);
        if (rc != 0) {
            core->team.ok = saved_ok && !core->team_disabled;  // (never re-enable what was off before the call)
            return -1;
        }
        const bool team_wins = ms[1] < ms[0] && !core->team_disabled;
        core->team.ok = team_wins;
        if (team_wins) core->team.min_batch = 1;  // measured for this plan's batch: the static crossover no longer applies
        return team_wins ? 1 : 0;
    };
    if (p->p32) return measure(p->p32);
    if (p->p64) return measure(p->p64);
    return 0;
}

int fft_gpu_plan_set_option_hip(fft_gpu_plan_t p, fft_gpu_plan_option_t option, int value) {
    if (!p) return -1;
    auto each_core = [&](auto&& f) { for_each_core(p, f); };
    switch (option) {
        case FFT_GPU_OPT_TEAM_NO_REPLAY:
            // only plain 1D complex plans ever replay (the cores of composite plans transform the plan's own intermediates)
            if (p->p32) p->p32->team_replay = value == 0;
            if (p->p64) p->p64->team_replay = value == 0;
            return 0;
        case FFT_GPU_OPT_TEAM_FORCE_FALLBACK:
            each_core([&](auto* c) { c->team_force_fallback = value != 0; });
            return 0;
        case FFT_GPU_OPT_TEAM_ENABLE:
            each_core([&](auto* c) {
                c->team.ok = value != 0 && c->team.tables != nullptr && !c->team_disabled;
                c->team_suspend = 0;
            });
            return 0;
        case FFT_GPU_OPT_NO_FUSION:
            if (p->b32) p->b32->no_fusion = value != 0;
            if (p->b64) p->b64->no_fusion = value != 0;
            if (p->f32) p->f32->no_fusion = value != 0;
            if (p->f64) p->f64->no_fusion = value != 0;
            return (p->b32 || p->b64 || p->f32 || p->f64) ? 0 : -1;
        case FFT_GPU_OPT_NO_CHAIN: {  // 0 the planner's rule, 1 never, 2 wherever the tiles agree (tools: the size rule is a measured one)
            auto set = [&](auto* q) {
                if (!q) return;
                q->no_chain = value == 1;
                q->core.chain_min_log2n = value == 2 ? 0 : std::remove_reference_t<decltype(q->core)>::kChainMinLog2n;
            };
            set(p->b32); set(p->b64); set(p->f32); set(p->f64);
            return (p->b32 || p->b64 || p->f32 || p->f64) ? 0 : -1;
        }
        default:
            return -1;
    }
}

int fft_gpu_plan_team_status_hip(fft_gpu_plan_t p) {
    if (!p) return -1;
    DeviceGuard guard(p->device);
    HIP_TRY(hipStreamSynchronize(p->rt.stream), return -1);
    return plan_team_status(p);
}

void fft_gpu_execute_hip(fft_gpu_plan_t p, fft_gpu_memory_t in, fft_gpu_memory_t out, fft_direction /*ignored*/) {
    if (!p || !in || !out) return;
    size_t need_in = 0, need_out = 0;
    plan_io_bytes(p, &need_in, &need_out);
    if (in->size < need_in || out->size < need_out) {
        fprintf(stderr, "fft_hip: execute needs %zu-byte input and %zu-byte output buffers (got in=%zu out=%zu)\n", need_in, need_out,
                in->size, out->size);
        return;
    }
    if (plan_enqueue(p, in->device_ptr, out->device_ptr) != 0) return;
    (void)fft_gpu_plan_sync_hip(p);  // the reference blocks here (cudaDeviceSynchronize, fft_cuda.cu:184)
}

int fft_gpu_execute_timed_hip(fft_gpu_plan_t p, const void* d_in, void* d_out, int iters, float* elapsed_ms) {
    if (!p || iters <= 0 || !elapsed_ms) return -1;
    DeviceGuard guard(p->device);
    if (!p->ev0) {
        HIP_TRY(hipEventCreate(&p->ev0), return -1);
        HIP_TRY(hipEventCreate(&p->ev1), return -1);
    }
    HIP_TRY(hipEventRecord(p->ev0, p->rt.stream), return -1);
    for (int i = 0; i < iters; i++)
        if (plan_enqueue(p, d_in, d_out) != 0) return -1;
    HIP_TRY(hipEventRecord(p->ev1, p->rt.stream), return -1);
    HIP_TRY(hipEventSynchronize(p->ev1), return -1);
    HIP_TRY(hipEventElapsedTime(elapsed_ms, p->ev0, p->ev1), return -1);
    if (plan_team_status(p) == (int)fftk::TEAM_STATUS_TIMEOUT) return -1;
    return 0;
}

// One execute with a HIP event after every pass launch: ms[i] = summed device time of pass i's launches,
// launches[i] = how many launches that was.  Events are recorded on the plan's stream.
int fft_gpu_profile_passes_hip(fft_gpu_plan_t p, const void* d_in, void* d_out, float* ms, int* launches, int max_passes) {
    if (!p || !ms || !launches || max_passes <= 0) return -1;
    DeviceGuard guard(p->device);
    for (int i = 0; i < max_passes; i++) { ms[i] = 0.f; launches[i] = 0; }
    hipEvent_t start;
    HIP_TRY(hipEventCreate(&start), return -1);
    p->rt.marks.clear();
    HIP_TRY(hipEventRecord(start, p->rt.stream), return -1);
    p->rt.profiling = true;
    int rc = plan_enqueue(p, d_in, d_out);
    p->rt.profiling = false;
    HIP_TRY(hipStreamSynchronize(p->rt.stream), rc = -1);
    hipEvent_t prev = start;
    for (auto& m : p->rt.marks) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, prev, m.second) == hipSuccess && m.first < max_passes) {
            ms[m.first] += t;
            launches[m.first] += 1;
        }
        prev = m.second;
    }
    for (auto& m : p->rt.marks) (void)hipEventDestroy(m.second);
    p->rt.marks.clear();
    (void)hipEventDestroy(start);
    return rc;
}

int fft_gpu_plan_info_hip(fft_gpu_plan_t p, fft_gpu_plan_info_t* info) {
    if (!p || !info) return -1;
    memset(info, 0, sizeof(*info));
    info->n = p->n;
    info->batch = p->batch;
    info->direction = p->dir;
    info->precision = p->prec;
    info->device = p->device;
    auto fill = [&](auto* core) {
        info->algo = core->algo;
        info->chunk_batch = core->chunk;
        info->team_tiles = core->team.ok ? core->team.NT : 0;
        info->team_kernel = !core->team.ok ? 0 : core->team.quad ? 3 : core->team.defer ? 2 : 1;
        if (core->team.ok) info->workspace_bytes += core->team.scratch_bytes;
        info->workspace_bytes += core->scratch_bytes;
        if (core->algo == ffteng::ALGO_RADIX2_SHFL) {
            info->n_passes = 1;
            info->factors[0] = 1 << core->log2n;
        } else if (core->algo == ffteng::ALGO_RADIX2_GLOBAL) {
            info->n_passes = core->log2n + 1;
        } else {
            info->n_passes = core->wide.ok ? 1 : (int)core->passes.size();
            for (size_t i = 0; i < core->passes.size() && i < 4; i++) info->factors[i] = 1 << core->passes[i].log2L;
        }
    };
    if (p->p32) fill(p->p32);
    if (p->p64) fill(p->p64);
    if (p->b32) {
        fill(&p->b32->core);
        info->bluestein_m = 1 << p->b32->log2m;
        info->fused = (p->b32->core.hook_capable() && !p->b32->no_fusion) ? ((!p->b32->no_chain && p->b32->core.round_capable()) ? 3 : (!p->b32->no_chain && p->b32->core.chain_capable()) ? 2 : 1) : 0;
        info->workspace_bytes += (size_t)p->batch * ((size_t)1 << p->b32->log2m) * sizeof(complex32_t);
    }
    if (p->b64) {
        fill(&p->b64->core);
        info->bluestein_m = 1 << p->b64->log2m;
        info->fused = (p->b64->core.hook_capable() && !p->b64->no_fusion) ? ((!p->b64->no_chain && p->b64->core.round_capable()) ? 3 : (!p->b64->no_chain && p->b64->core.chain_capable()) ? 2 : 1) : 0;
        info->workspace_bytes += (size_t)p->batch * ((size_t)1 << p->b64->log2m) * sizeof(complex_t);
    }
    auto fill_fused = [&](auto* f) {  // fused consumers: the padded transform behind them
        fill(&f->core);
        const bool pair = !f->no_chain && f->kind != ffteng::FUSED_PSD;
        info->fused = f->fused() ? ((pair && f->kind != ffteng::FUSED_XCORR && f->core.round_capable()) ? 3 : (pair && f->core.chain_capable()) ? 2 : 1) : 0;
    };
    if (p->f32) fill_fused(p->f32);
    if (p->f64) fill_fused(p->f64);
    return 0;
}

const char* fft_gpu_get_device_name_hip(void) { return g_device_name; }

void fft_gpu_get_memory_info_hip(size_t* total, size_t* available) {
    if (!total || !available) return;
    *total = 0;
    *available = 0;
    if (!g_initialized) return;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
        *total = tot;
        *available = fr;
    } else {
        (void)hipGetLastError();
    }
}

int fft_gpu_device_count_hip(void) { return probe_device_count(); }

int fft_gpu_set_device_hip(int device) {
    int count = probe_device_count();
    if (device < 0 || device >= count) return -1;
    HIP_TRY(hipSetDevice(device), return -1);
    pthread_mutex_lock(&g_lock);
    g_device = device;
    if (const DeviceInfo* di = device_info(device)) snprintf(g_device_name, sizeof(g_device_name), "%s", di->name);
    pthread_mutex_unlock(&g_lock);
    return 0;
}

int fft_gpu_get_device_hip(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return dev;
}

// host-pointer batched transform: one batched plan, 64-bit offsets (the reference loops
// single transforms with `in + i * n` in int arithmetic, gpu/fft_gpu.c:366-374)
// Host arrays that are page-locked (fft_malloc / fft_gpu_host_register_hip / hipHostMalloc) go through the device in groups of
// ~128 MiB on three streams: while group g is transformed, group g + 1 is on its way in and group g - 1 on its way out (PCIe
// is full duplex): 36-37 GB/s each way against 26-27 GB/s of the plain copy-all / transform / copy-all path at 2-3 GB.
// Pageable arrays cannot be copied asynchronously and take the plain path.  Returns 1 when the pipeline does not apply.
static int dft_batch_pipelined(const void* in, void* out, int n, int batch, fft_direction dir, fft_precision_t prec) {
    const size_t esz = prec == FFT_PREC_F32 ? sizeof(complex32_t) : sizeof(complex_t);
    const size_t per = (size_t)n * esz;
    if (!fft_gpu_host_is_registered_hip(in) || !fft_gpu_host_is_registered_hip(out)) return 1;
    long long group = (long long)((128ull << 20) / per);
    if (group < 1) group = 1;
    // measured (tools/host_batch_time.py): 37 against 27 GB/s each way from 2 GB up, no gain at 0.4 GB (the pipeline's own set-up)
    if (group * 8 > batch) return 1;  // less than 1 GiB: the plain path
    fft_gpu_plan_t plan = fft_gpu_plan_1d_ex_hip(n, (int)group, dir, prec, FFT_GPU_ALGO_AUTO);
    if (!plan) return -1;
    DeviceGuard guard(plan->device);
    void* buf[2] = {nullptr, nullptr};
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_fft[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    int rc = -1;
    bool ok = hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 2 && ok; k++) {
        buf[k] = plan->rt.dmalloc((size_t)group * per);
        ok = buf[k] && hipEventCreateWithFlags(&ev_in[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ev_fft[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_out[k], hipEventDisableTiming) == hipSuccess;
    }
    if (ok) {
        const char* src = (const char*)in;
        char* dst = (char*)out;
        int g = 0;
        for (long long b0 = 0; b0 < batch && ok; b0 += group, g++) {
            const int k = g & 1;
            const int nb = (int)((batch - b0) < group ? (batch - b0) : group);
            if (g >= 2) ok = hipStreamWaitEvent(s_in, ev_out[k], 0) == hipSuccess;  // the group that used this buffer has left it
            ok = ok && hipMemcpyAsync(buf[k], src + (size_t)b0 * per, (size_t)nb * per, hipMemcpyHostToDevice, s_in) == hipSuccess &&
                 hipEventRecord(ev_in[k], s_in) == hipSuccess && hipStreamWaitEvent(plan->rt.stream, ev_in[k], 0) == hipSuccess &&
                 plan_enqueue(plan, buf[k], buf[k], nb) == 0 && hipEventRecord(ev_fft[k], plan->rt.stream) == hipSuccess &&
                 hipStreamWaitEvent(s_out, ev_fft[k], 0) == hipSuccess &&
                 hipMemcpyAsync(dst + (size_t)b0 * per, buf[k], (size_t)nb * per, hipMemcpyDeviceToHost, s_out) == hipSuccess &&
                 hipEventRecord(ev_out[k], s_out) == hipSuccess;
        }
        const bool drained = hipStreamSynchronize(s_in) == hipSuccess && hipStreamSynchronize(s_out) == hipSuccess;
        if (ok && drained && fft_gpu_plan_sync_hip(plan) == 0) rc = 0;
    }
    if (rc != 0) (void)hipGetLastError();
    for (int k = 0; k < 2; k++) {
        if (buf[k]) plan->rt.dfree(buf[k]);
        if (ev_in[k]) (void)hipEventDestroy(ev_in[k]);
        if (ev_fft[k]) (void)hipEventDestroy(ev_fft[k]);
        if (ev_out[k]) (void)hipEventDestroy(ev_out[k]);
    }
    if (s_in) (void)hipStreamDestroy(s_in);
    if (s_out) (void)hipStreamDestroy(s_out);
    fft_gpu_destroy_plan_hip(plan);
    return rc;
}

int fft_gpu_dft_1d_batch_hip(const void* in, void* out, int n, int batch, fft_direction dir, fft_precision_t prec) {
    if (!in || !out || n <= 0 || batch <= 0) return -1;
    if (!g_initialized && fft_gpu_init_hip() != 0) return -1;
    const size_t esz = prec == FFT_PREC_F32 ? sizeof(complex32_t) : sizeof(complex_t);
    const size_t bytes = (size_t)n * (size_t)batch * esz;
    {
        const int piped = dft_batch_pipelined(in, out, n, batch, dir, prec);
        if (piped <= 0) return piped;
    }
    fft_gpu_plan_t plan = fft_gpu_plan_1d_ex_hip(n, batch, dir, prec, FFT_GPU_ALGO_AUTO);
    if (!plan) return -1;
    fft_gpu_memory_t buf = fft_gpu_alloc_bytes_hip(bytes);
    int rc = -1;
    if (buf && fft_gpu_copy_h2d_bytes_hip(buf, in, bytes) == 0 && plan_enqueue(plan, buf->device_ptr, buf->device_ptr) == 0 &&
        fft_gpu_plan_sync_hip(plan) == 0 && fft_gpu_copy_d2h_bytes_hip(out, buf, bytes) == 0)
        rc = 0;
    fft_gpu_free_hip(buf);
    fft_gpu_destroy_plan_hip(plan);
    return rc;
}

int fft_gpu_dft_1d_hip(complex_t* in, complex_t* out, int n, fft_direction dir) {
    return fft_gpu_dft_1d_batch_hip(in, out, n, 1, dir, FFT_PREC_F64);
}

int fft_gpu_bit_reverse_hip(const void* d_in, void* d_out, int n, int batch, fft_precision_t prec, void* hip_stream) {
    if (!d_in || !d_out || n <= 0 || (n & (n - 1)) != 0 || batch <= 0) return -1;
    const long long total = (long long)n * batch;
    const int log2n = ffteng::ilog2(n);
    long long grid = (total + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipStream_t s = (hipStream_t)hip_stream;
    if (prec == FFT_PREC_F32)
        hipLaunchKernelGGL(fftk::bitrev_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s,
                           (const fftk::cpx<float>*)d_in, (fftk::cpx<float>*)d_out, log2n, total);
    else
        hipLaunchKernelGGL(fftk::bitrev_kernel<double>, dim3((unsigned)grid), dim3(256), 0, s,
                           (const fftk::cpx<double>*)d_in, (fftk::cpx<double>*)d_out, log2n, total);
    HIP_TRY(hipGetLastError(), return -1);
    HIP_TRY(hipStreamSynchronize(s), return -1);
    return 0;
}

}  // extern "C"
