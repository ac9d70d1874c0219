// fft_engine.h -- host-side planner and launcher of the power-of-two engine and
// of Bluestein on top of it.
//
// Templated on a runtime policy RT so that the same planning / launch code is
// driven by the HIP runtime in the product (fft_hip_backend.hip) and by the
// thread-based kernel emulation in tests/emu (no GPU in the build container).
//
// RT must provide:
//   void* dmalloc(size_t bytes);  void dfree(void*);
//   void  h2d(void* dst, const void* src, size_t bytes);          // blocking
//   template <class K, class... A> void launch(K kernel, long long grid, int block, size_t smem, A... args);
//   int   max_lds_bytes();   int num_cus();
//   template <class K> int max_blocks_per_cu(K kernel, int threads, size_t smem);
//   void  mark(int pass_index);   // profiling hook, called after each pass launch (no-op unless enabled)
//   void  memset_async(void*, int, size_t);
//   bool  team_geometry(int& log2seats, int& n_xcc, int& nthreads);   // false: no team kernel on this device
//   long long team_timeout_ticks();        // bound of a team wait (deadlock breaker)
//   long long team_form_timeout_ticks();   // bound of team formation (shared device -> multi-pass fallback)
//   EnginePolicy policy;                   // the few production knobs, read once when the runtime is set up
//   template <class K, class... A> void launch_coresident(K kernel, long long grid, int block, size_t smem, A... args);
//
// Scheme (reference optimizations/parallel_fft.c:213-272 is the
// CPU statement of the same four-step idea):
//   n <= one LDS tile        1 pass : rows in, rows out
//   n = L1*L2                2 passes: A = L2-strided column FFTs of length L1 + twiddle W_n^(k1 n2), written
//                                      in place-shaped into scratch; B = contiguous row FFTs of length L2,
//                                      transposed (c-contiguous) store => natural order, no separate transpose
//   n = L1*L2*L3             3 passes: A on n, then A on the rows of length L2*L3, then B
// The batch is processed in chunks so that the scratch image (chunk*n elements)
// can stay resident in the 256 MiB Infinity Cache between the passes.
#pragma once

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "fft_kernels.h"
#include "fft_kernels_chain.h"
#include "fft_team_list.h"
#include "fft_team_defer.h"
#if defined(FFT_EMU)
#include "fft_team_quad.h"
#include "fft_wide_row.h"
#else
#include "fft_team_quad_decl.h"
#define FFT_WIDE_DECL_ONLY
#include "fft_wide_row.h"
namespace fftk {
extern template __global__ void wide_row_kernel<float, 13>(WideParams<float>);
extern template __global__ void wide_row_kernel<float, 14>(WideParams<float>);
extern template __global__ void wide_row_kernel<double, 13, 16>(WideParams<double>);
}
#endif

namespace ffteng {

using fftk::cpx;

// Production policy knobs (everything else that steers the planner is an experiment switch: FFT_EXP_ENV, fft_device.h).
struct EnginePolicy {
    int team_mode = 1;         // FFT_HIP_TEAM: 0 never the team kernel, 1 where it measured faster, 2 every built size
    int team_min_batch = 0;    // FFT_HIP_TEAM_MIN_BATCH: > 0 overrides the measured batch crossover
    long long chunk_mb = 0;    // FFT_HIP_CHUNK_MB: > 0 overrides the multi-pass launch-group size
};

// Element-wise work fused into the first pass's load and the last pass's store of one execute (fftk::TileHooks).
// Pitches are in elements; 0 = the plan's n.  Tables live in device memory; pre_tab must be readable up to the next
// multiple of 16 bytes past n_in (the planner pads its own tables).
template <typename T>
struct ExecHooks {
    const fftk::cpx<T>* pre_tab = nullptr;
    int pre_mode = fftk::HOOK_NONE;
    long long n_in = 0;        // valid input samples per transform (0: n); the rest reads as zero
    long long in_pitch = 0;
    const fftk::cpx<T>* post_tab = nullptr;
    long long post_tab_b = 0;  // per-transform pitch of post_tab (0: shared)
    int post_mode = fftk::HOOK_NONE;
    long long n_out = 0;       // outputs stored per transform (0: n)
    long long out_pitch = 0;
    // single-pass plans only (execute_round): the spectral product between the forward and the inverse transform of ONE kernel
    const fftk::cpx<T>* mid_tab = nullptr;
    int mid_mode = fftk::HOOK_NONE;
};

enum Algo { ALGO_AUTO = 0, ALGO_RADIX2 = 1, ALGO_RADIX4 = 2, ALGO_SPLIT_RADIX = 3, ALGO_RADIX2_GLOBAL = 4, ALGO_BLUESTEIN = 5, ALGO_RADIX2_SHFL = 6 };

struct PassDesc {
    int log2L = 0, log2C = 0, E = 16;
    int log2H = 0;  // column groups per tile (kernel template H = 1 << log2H)
    int tw_levels = 2;  // inter-pass twiddle: 2- or 3-level table product
    int fam = 0;        // butterfly family of this pass (fftk::FAM_*)
    int sa_bits = 0, t0_bits = 0, t1_bits = 0, t2_bits = 0;
    int o_sb = 0, o_t0 = 0, o_t1 = 0, o_t2 = 0, tables_elems = 0, off_tables = 0, group_bytes = 0;
    int loadm = 0, storem = 0, twiddle = 0;
    int n_ct = 1, n_o = 1;
    long long n_b_per_transform = 1;  // tiles along "b" contributed by ONE transform of the batch
    long long in_b = 0, in_o = 0, in_c = 0, in_l = 0;
    long long out_b = 0, out_o = 0, out_c = 0, out_k = 0;
    int in_blk_bits = 30;
    long long in_blk_stride = 0;
    int n_cols = 0;  // valid columns along the tiled dimension; -1: the batch (single-pass row kernel)
    int log2Ntw = 0;
    int nthreads = 0;
    int tw_o = 0;  // fftk::TileParams::tw_o
    long long col_stride = 1;  // fftk::TileParams::col_stride
    int smem_bytes = 0;
    int seg_bytes = 0;  // contiguous bytes per row segment on the c-contiguous side
};

inline int tile_E(long long L, int kind = 0);
inline int tile_E(long long L, int kind) {  // kind: 0 single-pass rows, 1 column pass, 2 row pass with transposed store
    // elements per thread of a tile kernel: 8 (32 data VGPRs + 2 x 32 prefetch VGPRs fit the 2-waves-per-SIMD
    // budget without spills; the radix-16 variant needs 64 + 64 and spilled -- see DESIGN.md "what was tried")
    static int pref[3] = {0, 0, 0};
    static bool init = false;
    if (!init) {
        init = true;
        pref[0] = 4; pref[1] = 8; pref[2] = 8;  // measured: rows-in/rows-out +8 % with 4 elements per thread (16 waves per CU); the multi-pass kernels lose 15-20 %
        if (const char* e = FFT_EXP_ENV("FFT_HIP_E")) sscanf(e, "%d,%d,%d", &pref[0], &pref[1], &pref[2]);
    }
    int e = (pref[kind] == 4) ? 4 : 8;
    while (e > L) e >>= 1;
    return e;
}

inline int ilog2(long long v) {
    int l = 0;
    while (v > 1) { v >>= 1; l++; }
    return l;
}

template <typename T>
static void make_twiddle_table(std::vector<cpx<T>>& t, long long period, long long count, long long step) {
    // t[i] = exp(-2 pi i * (i*step) / period), evaluated in long double, rounded once
    t.resize((size_t)count);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (long long i = 0; i < count; i++) {
        long long m = (i * step) % period;
        // exact symmetry reduction to the first octant keeps cos/sin arguments small
        long double ang = two_pi * (long double)m / (long double)period;
        t[(size_t)i].re = (T)cosl(ang);
        t[(size_t)i].im = (T)(-sinl(ang));
        if (m == 0) { t[(size_t)i].re = (T)1; t[(size_t)i].im = (T)0; }
        else if (4 * m == period) { t[(size_t)i].re = (T)0; t[(size_t)i].im = (T)-1; }
        else if (2 * m == period) { t[(size_t)i].re = (T)-1; t[(size_t)i].im = (T)0; }
        else if (4 * m == 3 * period) { t[(size_t)i].re = (T)0; t[(size_t)i].im = (T)1; }
    }
}

// Plan of the team kernel (fft_team.h): one transform per XCD, one HBM round trip.
template <typename T>
struct TeamDesc {
    bool ok = false;
    int log2L1 = 0, log2L2 = 0, log2CA = 0, log2CB = 0, log2TS = 0, NT = 0, nthreads = 0;
    int n_xcc = 0, log2seats = 0, n_teams = 0;  // XCDs x seats per XCD; n_teams = n_xcc << (log2seats - log2TS)
    bool defer = false;   // team_defer_kernel: the last row phase of a transform runs after the next one's column step
    bool asplit = false;  // column step on half-height, double-width tiles (128-byte row segments), fft_team.h ASPLIT
    bool alll2 = false;   // team_defer_kernel ALLL2 (with NODEFER): every phase handed over during the column step, two arrivals
    bool nodefer = false; // team_defer_kernel NODEFER (with PAIR): no deferred phase, two live windows per team
    bool pair = false;    // team_defer_kernel PAIR: adjacent row tiles in phases (0,1) / (2,3), 2 CB-row result segments (fp32)
    bool quad = false;    // team_quad_kernel (fft_team_quad.h): whole-line segments, both steps decimated by 4, exchange in four rounds
    int E = 0;  // elements per thread = stage radix (fp32: 16 -> 512 threads, 8 -> 1024 threads; fp64: 8)
    int data_bytes = 0, tables_elems = 0, smem_bytes = 0;
    int o_sb1 = 0, o_sa2 = 0, o_sb2 = 0, o_t0 = 0, o_t1 = 0, sa1_bits = 0, sa2_bits = 0, t0_bits = 0;
    int min_batch = 1;
    cpx<T>* tables = nullptr;
    unsigned char* scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned* ctl = nullptr;
    unsigned* sticky = nullptr;  // the allocation: [ TEAM_STICKY_WORDS | control block ]; ctl points behind the sticky words
    long long* trace = nullptr;  // profiling only (fft_gpu_plan_team_trace_hip); caller-owned device memory
    int trace_events = 0;
};

template <typename T, typename RT>
class Pow2Plan {
  public:
    static constexpr int V = 16 / (int)sizeof(cpx<T>);
    static constexpr int SZ = (int)sizeof(cpx<T>);

    RT* rt = nullptr;
    int log2n = 0;
    int algo = ALGO_AUTO;  // resolved (never AUTO after build)
    int fam = fftk::FAM_SR16;
    int max_batch = 1;
    int chunk = 1;  // transforms per launch group
    std::vector<PassDesc> passes;
    cpx<T>* scratch = nullptr;
    size_t scratch_bytes = 0;
    cpx<T>* scratch2 = nullptr;  // execute_chain: the inverse transform's image (scratch_bytes; allocated in build() for prefer_chain plans)
    std::vector<cpx<T>*> pass_tables;  // one device blob per pass: [sa | sb | t0 | t1 | t2]
    cpx<T>* tw_half = nullptr;  // W_n^k, k < n/2 (RADIX2_GLOBAL)
    TeamDesc<T> team;           // team.ok: execute() runs the team kernel, with the two-pass plan queued behind it as fallback
    const unsigned* run_if = nullptr;  // handed to the tile launches of the current execute (fallback mode)
    int team_pending = 0;        // team launches since the host last read the status word
    // what those launches were asked to do: replayed on the multi-pass schedule if one of them ends in TEAM_STATUS_TIMEOUT
    // (recover_after_timeout) -- where that is provably safe; see there.  `in` is what the replay reads: the caller's input, or the
    // plan's staged copy of it for an in-place execute of a plan whose team kernel has not yet been seen to work (team_proven).
    struct TeamExec { const cpx<T>* in; cpx<T>* out; int nb; bool inverse; bool scale_inverse; bool staged; };
    std::vector<TeamExec> team_log;
    int team_log_dropped = 0;    // executes beyond the log's capacity since the last sync: lost if a timeout voids them
    bool team_replay = true;     // the caller keeps every buffer of an execute alive and unmodified until the next sync (FFT_GPU_OPT_TEAM_NO_REPLAY clears it)
    bool team_proven = false;    // a sync has seen a team launch of this plan end with status OK: in-place executes are no longer staged
    cpx<T>* team_stage = nullptr;  // staged input of in-place executes until then (freed by the sync that proves the kernel)
    size_t team_stage_bytes = 0, team_stage_used = 0;
    int team_suspend = 0;        // executes that still skip the team kernel after it fell back three times in a row (a shared or
    int team_suspensions = 0;    // partitioned device): then it is tried again, with a short formation timeout; the pause doubles (64 ... 1024)
    bool team_disabled = false;  // a team wait timed out once: the team kernel is never enabled again for this plan
    int team_unrecoverable = 0;  // executes a timeout invalidated and recover_after_timeout could not repeat (since the last one)
    int team_fallbacks = 0;      // consecutive executes that ended in the two-pass fallback
    int team_last_status = -1;   // what the host last read from the status word (-1: never launched)
    bool team_force_fallback = false;  // test hook: the team kernel pretends its placement check failed
    bool ok = false;

    ~Pow2Plan() { destroy(); }

    void destroy() {
        if (!rt) return;
        for (auto* t : pass_tables) rt->dfree(t);
        pass_tables.clear();
        if (tw_half) rt->dfree(tw_half);
        if (scratch) rt->dfree(scratch);
        if (scratch2) rt->dfree(scratch2);
        scratch2 = nullptr;
        delete mirror;
        mirror = nullptr;
        if (wide.tables) rt->dfree(wide.tables);
        wide = WideDesc();
        if (team.tables) rt->dfree(team.tables);
        if (team.scratch) rt->dfree(team.scratch);
        if (team.sticky) rt->dfree(team.sticky);
        if (team_stage) rt->dfree(team_stage);
        team_stage = nullptr;
        team_stage_bytes = team_stage_used = 0;
        team = TeamDesc<T>();
        tw_half = nullptr;
        scratch = nullptr;
    }

    // ---- team_quad_kernel (fft_team_quad.h): n = L x L with L = 4 E^2, teams of TS = n / (4 tiles) seats.  Built for fp32
    // n = 2^20 on the device (E = 16, a whole XCD per transform) and for n = 2^12 in the emulation (E = 4, 64-thread
    // workgroups, teams of 4).  Tables [W_n^x, x < L | W_L^y, y < L]; window: 2 slots of TS images per team.
    // ---- wide_row_kernel (fft_wide_row.h): single-pass n = 8192 fp32 (the emulation: n = 512, 32 threads)
    struct WideDesc {
        bool ok = false;
        int nthreads = 0, smem_bytes = 0, tables_elems = 0, o_sb = 0, sa_bits = 0;
        cpx<T>* tables = nullptr;
    } wide;
    void build_wide() {
        if (!rt->wide_rows(SZ, log2n)) return;
        const long long L = 1ll << log2n;
        WideDesc d;
        d.nthreads = (int)(L / 16);
        d.sa_bits = fftk::team_stage_table_bits(SZ, log2n);
        d.o_sb = 1 << d.sa_bits;
        int ne = d.o_sb + (1 << (log2n - d.sa_bits));
        ne = (ne * SZ + 15) / 16 * 16 / SZ;
        d.tables_elems = ne;
        d.smem_bytes = (int)((2 * L * SZ > 140 * 1024 ? 1 : 2) * L * SZ) + ne * SZ;  // n = 16384: one image, run in place
        if (d.smem_bytes > rt->max_lds_bytes()) return;
        std::vector<cpx<T>> blob((size_t)ne), part;
        for (auto& z : blob) { z.re = (T)1; z.im = (T)0; }
        make_twiddle_table<T>(part, L, 1ll << d.sa_bits, 1);
        std::copy(part.begin(), part.end(), blob.begin());
        make_twiddle_table<T>(part, L, 1ll << (log2n - d.sa_bits), 1ll << d.sa_bits);
        std::copy(part.begin(), part.end(), blob.begin() + d.o_sb);
        d.tables = (cpx<T>*)rt->dmalloc(blob.size() * SZ);
        if (!d.tables) return;
        rt->h2d(d.tables, blob.data(), blob.size() * SZ);
        d.ok = true;
        wide = d;
    }
    void launch_wide(const cpx<T>* in, cpx<T>* out, int nb, bool inverse, T scale) {
        if constexpr (SZ == 16) {
#if !defined(FFT_EMU)
            fftk::WideParams<T> wp;
            memset(&wp, 0, sizeof(wp));
            wp.in = in; wp.out = out; wp.tables = wide.tables; wp.tables_bytes = wide.tables_elems * SZ;
            wp.o_sb = wide.o_sb; wp.sa_bits = wide.sa_bits; wp.nb = nb; wp.inverse = inverse ? 1 : 0; wp.nt = 3; wp.scale = scale;
            rt->launch(fftk::wide_row_kernel<T, 13, 16>, std::min<long long>(nb, rt->num_cus()), wide.nthreads, (size_t)wide.smem_bytes, wp);
#endif
        }
        if constexpr (SZ == 8) {
            fftk::WideParams<T> wp;
            memset(&wp, 0, sizeof(wp));
            wp.in = in; wp.out = out; wp.tables = wide.tables; wp.tables_bytes = wide.tables_elems * SZ;
            wp.o_sb = wide.o_sb; wp.sa_bits = wide.sa_bits; wp.nb = nb; wp.inverse = inverse ? 1 : 0; wp.nt = 3; wp.scale = scale;
            const long long grid = std::min<long long>(nb, rt->num_cus());
#if defined(FFT_EMU)
            rt->launch(fftk::wide_row_kernel<T, 9>, grid, wide.nthreads, (size_t)wide.smem_bytes, wp);
#else
            if (log2n == 14) rt->launch(fftk::wide_row_kernel<T, 14>, grid, wide.nthreads, (size_t)wide.smem_bytes, wp);
            else rt->launch(fftk::wide_row_kernel<T, 13>, grid, wide.nthreads, (size_t)wide.smem_bytes, wp);
#endif
        }
    }

    // ---- team_quad_kernel (fft_team_quad.h): n = L1 x L2 (L1 >= L2), L1 = 4 E RA, L2 = 4 E RB (E values per thread and chunk, a
    // radix-E and a radix-RA / RB stage), teams of TS = L2 RA / threads seats.  Device (512 threads, E = 16): n = 2^20 (RA RB 16 16,
    // teams of 32), 2^19 (16 8, 16), 2^18 (8 8, 8), 2^17 (8 4, 4), 2^16 (4 4, 2); emulation (E = 4): n = 2^12 (4 4), 2^11 (4 2)
    // and 2^10 (2 2).  Tables [W_n^x, x < L2/2 | W_L1^y, y < L1 | W_L2^y, y < L2 | W_n^(L2/2)]; window: 2 slots of TS images per team.
    // fp64 (one value per 16-byte access, 8 values per thread and chunk): n = 2^14 = 256 x 64 on one CU, 2^15 = 256 x 128 on teams of 2,
    // 2^16 = 256 x 256 on teams of 4 (RA <= E: no larger size has two-stage sub-transforms).
    static constexpr int quad_E(int log2n_) {
        return SZ == 8 ? ((log2n_ >= 15 && log2n_ <= 20) ? 16 : (log2n_ >= 10 && log2n_ <= 12) ? 4 : 0)
                       : ((log2n_ >= 14 && log2n_ <= 16) ? 8 : (log2n_ >= 10 && log2n_ <= 12) ? 4 : 0);
    }
    void build_team_quad(int batch) {
        const int mode = rt->policy.team_mode;
        if (mode <= 0) return;
        const int E = quad_E(log2n);
        if (!E || !rt->team_quad(SZ, log2n)) return;
        TeamDesc<T> d;
        if (!rt->team_geometry(d.log2seats, d.n_xcc, d.nthreads)) return;
        // (fp64 2^14: 256 x 64, not 128 x 128 -- a wave of the row step must lie in ONE block of MA rows, and MA = 32 there)
        const int log2L1 = (SZ == 16 && log2n == 14) ? 8 : (log2n + 1) / 2, log2L2 = log2n - log2L1;
        const long long L1 = 1ll << log2L1, L2 = 1ll << log2L2;
        const long long RA = L1 / 4 / E, RB = L2 / 4 / E;
        if (RB < 2 || RA > E || RA * E * 4 != L1 || RB * E * 4 != L2) return;
        const int log2TE = log2L1 - 2 + ilog2(d.nthreads) - ilog2(RA);  // values of a chunk image: MA rows x (threads / RA) columns
        d.log2TS = log2n - 2 - log2TE;
        if (d.log2TS < 0 || d.log2TS > d.log2seats) return;
        const long long NC = L2 >> d.log2TS, NR = L1 >> d.log2TS;
        if (NC * RA != d.nthreads || NR * RB != d.nthreads || NC < 8) return;
#if !defined(FFT_EMU)
        if (d.log2TS != log2n - (SZ == 8 ? 15 : 14)) return;  // the device instantiations
#else
        if (!((log2n == 12 && d.log2TS == 2) || ((log2n == 11 || log2n == 10) && (d.log2TS == 1 || d.log2TS == 2)) || (log2n == 10 && d.log2TS == 0))) return;  // the emulation's
#endif
        if (mode == 1 && !rt->team_default_on(SZ, log2n)) return;
        d.quad = true;
        d.E = E;
        d.NT = 4;
        d.log2L1 = log2L1;
        d.log2L2 = log2L2;
        d.n_teams = d.n_xcc << (d.log2seats - d.log2TS);
        if (d.n_teams > fftk::TEAM_CTL_MAX_TEAMS) return;
        d.data_bytes = 2 * (SZ << log2TE);
        d.tables_elems = (int)(L2 / 2 + L1 + L2);  // what the kernel keeps in LDS; W_n^(L2/2) follows in the blob
        d.smem_bytes = d.data_bytes + d.tables_elems * SZ + 32;  // + sh[8]: team slot, XCD, formed, timed out, next transform
        if (d.smem_bytes > rt->max_lds_bytes()) return;
        std::vector<cpx<T>> blob, part;
        make_twiddle_table<T>(blob, L1 * L2, L2 / 2, 1);
        make_twiddle_table<T>(part, L1, L1, 1);
        blob.insert(blob.end(), part.begin(), part.end());
        make_twiddle_table<T>(part, L2, L2, 1);
        blob.insert(blob.end(), part.begin(), part.end());
        make_twiddle_table<T>(part, L1 * L2, 2, L2 / 2);
        blob.push_back(part[1]);
        blob.push_back(part[0]);  // (pad to 16 bytes)
        d.scratch_bytes = ((size_t)SZ << (log2TE + d.log2TS)) * 2 * (size_t)d.n_teams;  // two window slots per team (one used where the kernel is built with one)
        d.tables = (cpx<T>*)rt->dmalloc(blob.size() * SZ);
        d.scratch = (unsigned char*)rt->dmalloc(d.scratch_bytes);
        d.sticky = (unsigned*)rt->dmalloc((fftk::TEAM_STICKY_WORDS + fftk::TEAM_CTL_WORDS) * sizeof(unsigned));
        d.ctl = d.sticky ? d.sticky + fftk::TEAM_STICKY_WORDS : nullptr;
        if (!d.tables || !d.scratch || !d.sticky) {
            rt->dfree(d.tables); rt->dfree(d.scratch); rt->dfree(d.sticky);
            return;
        }
        rt->memset_async(d.sticky, 0, fftk::TEAM_STICKY_WORDS * sizeof(unsigned));
        rt->h2d(d.tables, blob.data(), blob.size() * SZ);
        // batch crossover against the multi-pass schedule (tools/batch_crossover.py; profiles/r3_batch_crossover.txt, the last table --
        // measured after team formation had shrunk from 40 to 15 us): n = 2^20 and 2^19 from 128 MiB per execute (145 vs 111, 151 vs 123
        // Gpoint/s), 2^18 ... 2^15 from 256 MiB (178 vs 164, 189 vs 169, 188 vs 174, 188 vs 179); and at least 2 transforms per team
        {
            const long long mib = d.log2TS >= 4 ? 128 : 256;
            d.min_batch = mode == 1 ? (int)std::max<long long>(2ll * d.n_teams, (mib << 20) / ((long long)SZ << log2n)) : d.n_teams;
        }
        if (rt->policy.team_min_batch > 0) d.min_batch = rt->policy.team_min_batch;
        (void)batch;
        d.ok = true;
        team = d;
    }

    void launch_team_quad(const fftk::TeamParams<T>& tp) {
        const long long grid = (long long)team.n_xcc << team.log2seats;
#define FFT_QUAD_GO(...) rt->launch_coresident(fftk::team_quad_kernel<T, __VA_ARGS__>, grid, team.nthreads, (size_t)team.smem_bytes, tp)
#if defined(FFT_EMU)
        const bool pair = getenv("FFT_EMU_QUAD_SLOTS") && atoi(getenv("FFT_EMU_QUAD_SLOTS")) == 3 && team.log2TS >= 1;
        if (pair && log2n == 12) FFT_QUAD_GO(4, 2, 2, 6, 6, 2, 3);  // the pair protocol (the device's n = 2^20) on teams of 4 and of 2
        else if (pair && log2n == 11 && team.log2TS == 1) FFT_QUAD_GO(4, 2, 1, 6, 5, 1, 3);
        else if (pair && log2n == 11) FFT_QUAD_GO(4, 2, 1, 6, 5, 2, 3);
        else if (pair && team.log2TS == 1) FFT_QUAD_GO(4, 1, 1, 5, 5, 1, 3);
        else if (pair) FFT_QUAD_GO(4, 1, 1, 5, 5, 2, 3);
        else if (log2n == 12) FFT_QUAD_GO(4, 2, 2, 6, 6, 2, 2);
        else if (log2n == 11 && team.log2TS == 1) FFT_QUAD_GO(4, 2, 1, 6, 5, 1, 2);
        else if (log2n == 11) FFT_QUAD_GO(4, 2, 1, 6, 5, 2, 1);
        else if (team.log2TS == 0) FFT_QUAD_GO(4, 1, 1, 5, 5, 0, 1);
        else if (team.log2TS == 1) FFT_QUAD_GO(4, 1, 1, 5, 5, 1, 1);
        else FFT_QUAD_GO(4, 1, 1, 5, 5, 2, 1);
#else
        if constexpr (SZ == 8) {
            const int slots = rt->team_quad_slots(log2n);
            if (log2n == 20 && slots == 3) FFT_QUAD_GO(16, 4, 4, 10, 10, 5, 3);
            else if (log2n == 19 && slots == 3) FFT_QUAD_GO(16, 4, 3, 10, 9, 4, 3);
            else if (log2n == 18 && slots == 3) FFT_QUAD_GO(16, 3, 3, 9, 9, 3, 3);
            else if (log2n == 17 && slots == 3) FFT_QUAD_GO(16, 3, 2, 9, 8, 2, 3);
            else if (log2n == 20 && slots == 1) FFT_QUAD_GO(16, 4, 4, 10, 10, 5, 1);
            else if (log2n == 20) FFT_QUAD_GO(16, 4, 4, 10, 10, 5, 2);
            else if (log2n == 19 && slots == 1) FFT_QUAD_GO(16, 4, 3, 10, 9, 4, 1);
            else if (log2n == 19) FFT_QUAD_GO(16, 4, 3, 10, 9, 4, 2);
            else if (log2n == 18 && slots == 2) FFT_QUAD_GO(16, 3, 3, 9, 9, 3, 2);
            else if (log2n == 18) FFT_QUAD_GO(16, 3, 3, 9, 9, 3, 1);
            else if (log2n == 17) FFT_QUAD_GO(16, 3, 2, 9, 8, 2, 1);
            else if (log2n == 15) FFT_QUAD_GO(16, 2, 1, 8, 7, 0, 1);
            else FFT_QUAD_GO(16, 2, 2, 8, 8, 1, 1);
        } else {
            if (log2n == 14) FFT_QUAD_GO(8, 3, 1, 8, 6, 0, 1);
            else if (log2n == 15) FFT_QUAD_GO(8, 3, 2, 8, 7, 1, 1);
            else FFT_QUAD_GO(8, 3, 3, 8, 8, 2, 1);
        }
#endif
#undef FFT_QUAD_GO
        (void)grid;
    }

    // ---- team kernel: geometry, tables, the L2-resident transposition windows, the control block
    void build_team(int batch) {
        // FFT_HIP_TEAM: 0 never; 1 (default) where it measured faster than the multi-pass schedule on MI355X
        // (RT::team_default_on); 2 every size the kernel is built for (fft_team_list.h)
        const int mode = rt->policy.team_mode;
        if (mode <= 0) return;
        TeamDesc<T> d;
        if (!rt->team_geometry(d.log2seats, d.n_xcc, d.nthreads)) return;
        const int log2V = ilog2(V);
        const int log2TE = ilog2(d.nthreads) + 3 + log2V;  // elements of a tile: threads * 8 lane accesses of V elements
        d.E = 8 * V;  // elements per thread = the radix of the stages (16 fp32, 8 fp64)
        // Tiles per workgroup and step (NT) x team size (TS) x tile = n.  Four tiles per workgroup keep the hand-over
        // pipeline of fft_team.h full (two phases written during the column step, two from registers), so the team
        // shrinks with n: TS = n / (4 tiles): 32 CUs (a whole XCD) for n = 2^20 fp32, 16, 8, 4, 2 below.  n too large
        // for NT = 4 on a whole XCD does not fit the team's registers; NT = 2 / 1 on a whole XCD exist in the kernel
        // (and are emulated) but lose to the multi-pass schedule.
        int log2NT = 2;
        if (const char* e = FFT_EXP_ENV("FFT_HIP_TEAM_TILES")) log2NT = ilog2(atoi(e));  // tests (emulation): 1 or 2 tiles
        d.log2TS = log2n - log2NT - log2TE;
        if (d.log2TS > d.log2seats) return;
        if (d.log2TS < 1) return;  // a "team" of one CU exchanges nothing: single-CU sizes belong to the multi-pass plan
        d.NT = 1 << log2NT;
        d.n_teams = d.n_xcc << (d.log2seats - d.log2TS);
        if (mode == 1 && !rt->team_default_on(SZ, log2n)) return;
        d.log2L1 = log2n / 2;
        if (const char* e = FFT_EXP_ENV("FFT_HIP_TEAM_L1")) d.log2L1 = atoi(e);  // experiments / tests: force the split
        d.log2L2 = log2n - d.log2L1;
        d.log2CA = log2TE - d.log2L1;
        d.log2CB = log2TE - d.log2L2;
        // a thread holds 8 * V elements of one column; fp32 pairs the lanes of adjacent rows for its 16-byte stores
        const int log2E = ilog2(d.E);
        if (d.log2L1 < log2E + log2V || d.log2L2 < log2E || d.log2CA < log2V || d.log2CB < log2V || d.log2CA > 5) return;
        if (d.n_teams > fftk::TEAM_CTL_MAX_TEAMS) return;
        // ASPLIT: built for four tiles per workgroup; on the device only where it is instantiated (RT::team_asplit)
        d.asplit = d.NT == 4 && d.log2CA + 1 <= 5 && d.log2L1 - 1 >= log2E + log2V && rt->team_asplit(SZ, log2n);
        const long long L1 = (1ll << d.log2L1) >> (d.asplit ? 1 : 0), L2 = 1ll << d.log2L2;  // L1: the column stage table's length
        d.data_bytes = 2 * (SZ << log2TE);  // LDS-DMA landing image + work image, one tile each
        // tables: [sa1 | sb1 | sa2 | sb2 | t0 | t1]
        const int log2L1tab = d.log2L1 - (d.asplit ? 1 : 0);
        d.sa1_bits = fftk::team_stage_table_bits(SZ, log2L1tab);
        d.sa2_bits = fftk::team_stage_table_bits(SZ, d.log2L2);
        d.t0_bits = (log2n + 1) / 2;
        int ne = 1 << d.sa1_bits;
        d.o_sb1 = ne; ne += 1 << (log2L1tab - d.sa1_bits);
        if (d.log2L2 == log2L1tab) { d.o_sa2 = 0; d.o_sb2 = d.o_sb1; }
        else { d.o_sa2 = ne; ne += 1 << d.sa2_bits; d.o_sb2 = ne; ne += 1 << (d.log2L2 - d.sa2_bits); }
        d.o_t0 = ne; ne += 1 << d.t0_bits;
        d.o_t1 = ne; ne += 1 << (log2n - d.t0_bits);
        ne = (ne * SZ + 15) / 16 * 16 / SZ;
        d.tables_elems = ne;
        d.smem_bytes = d.data_bytes + ne * SZ + 16;
        if (d.smem_bytes > rt->max_lds_bytes()) return;
        std::vector<cpx<T>> blob((size_t)ne), part;
        for (auto& z : blob) { z.re = (T)1; z.im = (T)0; }
        make_twiddle_table<T>(part, L1, 1ll << d.sa1_bits, 1);
        std::copy(part.begin(), part.end(), blob.begin());
        make_twiddle_table<T>(part, L1, 1ll << (log2L1tab - d.sa1_bits), 1ll << d.sa1_bits);
        std::copy(part.begin(), part.end(), blob.begin() + d.o_sb1);
        if (d.log2L2 != log2L1tab) {
            make_twiddle_table<T>(part, L2, 1ll << d.sa2_bits, 1);
            std::copy(part.begin(), part.end(), blob.begin() + d.o_sa2);
            make_twiddle_table<T>(part, L2, 1ll << (d.log2L2 - d.sa2_bits), 1ll << d.sa2_bits);
            std::copy(part.begin(), part.end(), blob.begin() + d.o_sb2);
        }
        const long long n = 1ll << log2n;
        make_twiddle_table<T>(part, n, 1ll << d.t0_bits, 1);
        std::copy(part.begin(), part.end(), blob.begin() + d.o_t0);
        make_twiddle_table<T>(part, n, 1ll << (log2n - d.t0_bits), 1ll << d.t0_bits);
        std::copy(part.begin(), part.end(), blob.begin() + d.o_t1);
        d.defer = d.NT == 4 && !d.asplit && rt->team_defer(SZ, log2n);
        d.pair = d.defer && V == 2 && d.log2CB >= 1 && rt->team_pair(SZ, log2n);
        d.nodefer = d.pair && rt->team_nodefer(SZ, log2n);
        d.alll2 = d.nodefer && rt->team_alll2(SZ, log2n);
        d.scratch_bytes = ((size_t)SZ << (log2TE + d.log2TS)) * (d.defer ? 4 : 2) * (size_t)d.n_teams;  // 2 (deferred kernel: up to 4) windows of TS tiles per team
        d.tables = (cpx<T>*)rt->dmalloc(blob.size() * SZ);
        d.scratch = (unsigned char*)rt->dmalloc(d.scratch_bytes);
        // [ sticky words | control block ]: only the control block is zeroed per launch
        d.sticky = (unsigned*)rt->dmalloc((fftk::TEAM_STICKY_WORDS + fftk::TEAM_CTL_WORDS) * sizeof(unsigned));
        d.ctl = d.sticky ? d.sticky + fftk::TEAM_STICKY_WORDS : nullptr;
        if (!d.tables || !d.scratch || !d.sticky) {
            rt->dfree(d.tables); rt->dfree(d.scratch); rt->dfree(d.sticky);
            return;
        }
        rt->memset_async(d.sticky, 0, fftk::TEAM_STICKY_WORDS * sizeof(unsigned));
        rt->h2d(d.tables, blob.data(), blob.size() * SZ);
        // Small executes keep the multi-pass schedule: the launch's fixed costs (team formation, pipeline fill, the last
        // transforms of uneven teams) and an intermediate that still fits the Infinity Cache favour it.  Measured crossover
        // (tools/batch_crossover.py, profiles/r2_batch_crossover.txt, against the multi-pass schedule as it stands at the end
        // of round 2 -- radix-8 column passes, nt streams, measured splits): fp32 n = 2^20 from 256 MiB per execute (131 vs
        // 116 Gpoint/s at 32 transforms), 2^19 from 512 MiB, 2^18 from 1 GiB, 2^17 only from 4 GiB (185 vs 182), 2^16 from
        // 2 GiB (193 vs 187); fp64 2^19 / 2^17 / 2^15 from 512 MiB; and at least 4 transforms per team.
        // fft_gpu_plan_measure_hip replaces the table by a measurement of the plan at hand.
        {
            long long mib = 512;
            if (SZ == 8) mib = d.log2TS >= 5 ? 256 : d.log2TS == 4 ? 512 : d.log2TS == 3 ? 1024 : d.log2TS == 2 ? 4096 : 2048;
            d.min_batch = mode == 1 ? (int)std::max<long long>(4ll * d.n_teams, (mib << 20) / ((long long)SZ << log2n)) : d.n_teams;
        }
        if (rt->policy.team_min_batch > 0) d.min_batch = rt->policy.team_min_batch;
        (void)batch;
        d.ok = true;
        team = d;
    }

    template <int LOG2N>
    void launch_team_n(const fftk::TeamParams<T>& tp) {
        const long long grid = (long long)team.n_xcc << team.log2seats;
        constexpr int GEO = fftk::TeamGeo<T, LOG2N>::value;
        if (GEO == 0) return;
        if (team.defer && team.pair && team.nodefer && team.alll2 && fftk::TeamPairBuilt<T, LOG2N>::value)
            rt->launch_coresident(fftk::team_defer_kernel<T, 8 * V, (GEO ? GEO : 1), fftk::TeamPairBuilt<T, LOG2N>::value, fftk::TeamPairBuilt<T, LOG2N>::value,
                                                          fftk::TeamPairBuilt<T, LOG2N>::value>,
                                  grid, team.nthreads, (size_t)team.smem_bytes, tp);
        else if (team.defer && team.pair && team.nodefer && fftk::TeamPairBuilt<T, LOG2N>::value)
            rt->launch_coresident(fftk::team_defer_kernel<T, 8 * V, (GEO ? GEO : 1), fftk::TeamPairBuilt<T, LOG2N>::value, fftk::TeamPairBuilt<T, LOG2N>::value>,
                                  grid, team.nthreads, (size_t)team.smem_bytes, tp);
        else if (team.defer && team.pair && fftk::TeamPairBuilt<T, LOG2N>::value)
            rt->launch_coresident(fftk::team_defer_kernel<T, 8 * V, (GEO ? GEO : 1), fftk::TeamPairBuilt<T, LOG2N>::value>, grid, team.nthreads,
                                  (size_t)team.smem_bytes, tp);
        else if (team.defer)
            rt->launch_coresident(fftk::team_defer_kernel<T, 8 * V, (GEO ? GEO : 1)>, grid, team.nthreads, (size_t)team.smem_bytes, tp);
        else if (fftk::TeamAsplitBuilt<T, LOG2N>::value && team.asplit)
            rt->launch_coresident(fftk::team_fft_kernel<T, 4, 8 * V, (GEO ? GEO : 1), fftk::TeamAsplitBuilt<T, LOG2N>::value>, grid,
                                  team.nthreads, (size_t)team.smem_bytes, tp);
        else
            rt->launch_coresident(fftk::team_fft_kernel<T, 4, 8 * V, (GEO ? GEO : 1)>, grid, team.nthreads, (size_t)team.smem_bytes, tp);
    }

    template <int NT>
    void launch_team_emu(const fftk::TeamParams<T>& tp) {
        const long long grid = (long long)team.n_xcc << team.log2seats;
        if (NT == 4 && team.defer && team.pair && team.nodefer && team.alll2)
            rt->launch_coresident(fftk::team_defer_kernel<T, 8 * V, 0, (V == 2), true, (V == 2)>, grid, team.nthreads, (size_t)team.smem_bytes, tp);
        else if (NT == 4 && team.defer && team.pair && team.nodefer)
            rt->launch_coresident(fftk::team_defer_kernel<T, 8 * V, 0, (V == 2), true>, grid, team.nthreads, (size_t)team.smem_bytes, tp);
        else if (NT == 4 && team.defer && team.pair)
            rt->launch_coresident(fftk::team_defer_kernel<T, 8 * V, 0, (V == 2)>, grid, team.nthreads, (size_t)team.smem_bytes, tp);
        else if (NT == 4 && team.defer)
            rt->launch_coresident(fftk::team_defer_kernel<T, 8 * V, 0>, grid, team.nthreads, (size_t)team.smem_bytes, tp);
        else if (NT == 4 && team.asplit)
            rt->launch_coresident(fftk::team_fft_kernel<T, 4, 8 * V, 0, true>, grid, team.nthreads, (size_t)team.smem_bytes, tp);
        else
            rt->launch_coresident(fftk::team_fft_kernel<T, NT, 8 * V, 0>, grid, team.nthreads, (size_t)team.smem_bytes, tp);
    }

    int built_geo() const {
        switch (log2n) {
            case 15: return fftk::TeamGeo<T, 15>::value;
            case 16: return fftk::TeamGeo<T, 16>::value;
            case 17: return fftk::TeamGeo<T, 17>::value;
            case 18: return fftk::TeamGeo<T, 18>::value;
            case 19: return fftk::TeamGeo<T, 19>::value;
            case 20: return fftk::TeamGeo<T, 20>::value;
            default: return 0;
        }
    }

    bool team_geometry_is_built() const {
        if (team.quad) return true;  // build_team_quad only plans the instantiated shapes
#if defined(FFT_EMU)
        return true;
#else
        return team.NT == 4 && built_geo() == FFT_TEAM_GEO(team.log2L1, team.log2L2, team.log2CA, team.log2CB, team.log2TS);  // (team.pair without a PAIR instantiation runs the plain deferred kernel)
#endif
    }

    void launch_team(const cpx<T>* in, cpx<T>* out, int nb, bool inverse, T scale) {
        fftk::TeamParams<T> tp;
        memset(&tp, 0, sizeof(tp));
        tp.in = in; tp.out = out;
        tp.tables = team.tables;
        tp.scratch = team.scratch;
        tp.ctl = team.ctl;
        tp.tables_bytes = team.tables_elems * SZ;
        tp.data_bytes = team.data_bytes;
        tp.log2L1 = team.log2L1; tp.log2L2 = team.log2L2; tp.log2CA = team.log2CA; tp.log2CB = team.log2CB; tp.log2TS = team.log2TS;
        tp.n_xcc = team.n_xcc;
        tp.log2seats = team.log2seats;
        tp.nb = nb;
        tp.inverse = inverse ? 1 : 0;
        tp.o_sb1 = team.o_sb1; tp.o_sa2 = team.o_sa2; tp.o_sb2 = team.o_sb2; tp.o_t0 = team.o_t0; tp.o_t1 = team.o_t1;
        tp.sa1_bits = team.sa1_bits; tp.sa2_bits = team.sa2_bits; tp.t0_bits = team.t0_bits;
        tp.timeout_ticks = rt->team_timeout_ticks();
        // (after a fallback the next attempts give up ten times sooner: a device that is shared NOW mostly still is)
        tp.form_timeout_ticks = (team_fallbacks > 0 || team_suspensions > 0) ? std::max<long long>(1, rt->team_form_timeout_ticks() / 10) : rt->team_form_timeout_ticks();
        tp.sticky = team.sticky;
        static const int ablate = FFT_EXP_ENV("FFT_HIP_TEAM_ABLATE") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_ABLATE")) : 0;  // profiling only
        tp.ablate = ablate;
        static const int dma_split = FFT_EXP_ENV("FFT_HIP_TEAM_DMA_SPLIT") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_DMA_SPLIT")) : 4;
        tp.dma_split = dma_split;
        static const int dma_split2 = FFT_EXP_ENV("FFT_HIP_TEAM_DMA_SPLIT2") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_DMA_SPLIT2")) : 8;
        tp.dma_split2 = dma_split2;
        static const int seat_rot = FFT_EXP_ENV("FFT_HIP_TEAM_SEAT_ROT") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_SEAT_ROT")) : 0;
        tp.seat_rot = seat_rot;
        // team_quad_kernel: transforms claimed team by team from a device-wide counter (the XCDs run 2 - 4 % apart; FFT_HIP_TEAM_DYNAMIC=0: static split)
        const int dynamic = FFT_EXP_ENV("FFT_HIP_TEAM_DYNAMIC") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_DYNAMIC")) : 1;
        tp.dynamic = team.quad ? dynamic : 0;
        // measured (profiles/r2_ab_pair.txt): 4 for the plain kernels, 2 where the row tiles are paired
        static const int tile_rot = FFT_EXP_ENV("FFT_HIP_TEAM_TILE_ROT") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_TILE_ROT")) : -1;
        tp.tile_rot = tile_rot >= 0 ? tile_rot : (team.pair ? 2 : 4);
        // column-tile DMA with the non-temporal bit (read once: the first lines to leave the L2, which keeps more of the
        // hand-over windows there): +1..3 % at every size; nt result stores and nt window loads measured even or worse
        // With paired row tiles (PAIR) a result store instruction writes whole 128-byte lines, so the result stream can be
        // non-temporal too (as 64-byte halves it could not: the halves left the L2 one by one, WRITE 8.6 -> 10 GB): +3-5 %
        static const int nt_mask = FFT_EXP_ENV("FFT_HIP_TEAM_NT") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_NT")) : -1;
        tp.nt_mask = nt_mask >= 0 ? nt_mask : (team.quad ? 7 : team.nodefer ? 7 : team.pair ? 3 : 1);
        static const int tune = FFT_EXP_ENV("FFT_HIP_TEAM_TUNE") ? atoi(FFT_EXP_ENV("FFT_HIP_TEAM_TUNE")) : 0;
        tp.tune = tune;
        tp.force_no_teams = team_force_fallback ? 1 : 0;  // test hook (fft_gpu_plan_set_option_hip): exercise the fallback on a healthy device
        tp.trace = team.trace;
        tp.trace_events = team.trace_events;
        tp.scale = scale;
        rt->memset_async(team.ctl, 0, fftk::TEAM_CTL_WORDS * sizeof(unsigned));
        if (team.quad) {
            launch_team_quad(tp);
            return;
        }
#if defined(FFT_EMU)
        switch (team.NT) {
            case 1: launch_team_emu<1>(tp); break;
            case 2: launch_team_emu<2>(tp); break;
            default: launch_team_emu<4>(tp); break;
        }
#else
        switch (log2n) {  // the device instantiations have their geometry baked in (fft_team_list.h)
            case 15: launch_team_n<15>(tp); break;
            case 16: launch_team_n<16>(tp); break;
            case 17: launch_team_n<17>(tp); break;
            case 18: launch_team_n<18>(tp); break;
            case 19: launch_team_n<19>(tp); break;
            default: launch_team_n<20>(tp); break;
        }
#endif
    }

    // ---- twiddle-table layout of a pass: fills the bit splits and element offsets, returns the element count
    static int layout_tables(PassDesc& p) {
        const long long L = 1ll << p.log2L;
        p.sa_bits = (L * SZ > 8192) ? (p.log2L + 1) / 2 : p.log2L;  // two-level stage table above 8 KiB
        int n = 1 << p.sa_bits;
        p.o_sb = n;
        n += 1 << (p.log2L - p.sa_bits);
        p.o_t0 = p.o_t1 = p.o_t2 = n;
        p.t0_bits = p.t1_bits = p.t2_bits = 0;
        if (p.twiddle) {
            if (p.tw_levels == 2) {  // W^m = t0[lo] * t1[hi]: one complex multiply per element
                p.t0_bits = (p.log2Ntw + 1) / 2;
                p.t1_bits = p.log2Ntw - p.t0_bits;
                p.t2_bits = 0;
            } else {                 // three short tables when the two-level pair does not fit in LDS
                p.t0_bits = (p.log2Ntw + 2) / 3;
                p.t1_bits = (p.log2Ntw - p.t0_bits + 1) / 2;
                p.t2_bits = p.log2Ntw - p.t0_bits - p.t1_bits;
            }
            p.o_t0 = n; n += 1 << p.t0_bits;
            p.o_t1 = n; n += 1 << p.t1_bits;
            p.o_t2 = n; n += 1 << p.t2_bits;
        }
        n = (n * SZ + 15) / 16 * 16 / SZ;
        p.tables_elems = n;
        return n;
    }

    // ---- LDS footprint of one pass for a candidate group width CG = 2^log2CG
    static int lds_bytes(PassDesc& p, int log2CG, int log2H = 0) {
        const long long L = 1ll << p.log2L, CG = 1ll << log2CG;
        long long data = CG * L * SZ;
        if (p.loadm == fftk::LOAD_LCONTIG || p.storem == fftk::STORE_LCONTIG) data = CG * (L * SZ + 16);
        data = (data + 15) & ~15ll;
        p.group_bytes = (int)data;
        long long off = data << log2H;
        p.off_tables = (int)off;
        off += (long long)layout_tables(p) * SZ;
        return off > 0x7fffffff ? 0x7fffffff : (int)off;
    }

    // Choose the tile: group width CG, H groups per tile (C = H * CG columns), thread count.
    //  - threads = (CG/V) * (L/E) <= 512 (the kernels are built for 2 waves per SIMD = 8 waves per CU);
    //  - H = 2 groups (each with its own LDS region, processed interleaved by the same threads) whenever that
    //    fits LDS: doubles the row segment seen by HBM, halves the barriers per byte, doubles the bytes in flight;
    //  - stop widening once the row segment reaches 512 bytes (no measurable gain beyond, tools/membench).
    static bool choose_tile(PassDesc& p, long long extent_cols, int budget) {
        const int log2E = ilog2(p.E);
        const int log2V = ilog2(V);
        // two interleaved groups per tile (H = 2) were measured: the column pass gains 7 %, every row pass loses
        // 20 %, and the pair is slower than H = 1 at every size tried -- kept behind FFT_HIP_GROUPS=1 for experiments
        static const int no_groups = FFT_EXP_ENV("FFT_HIP_GROUPS") ? 0 : 1;
        int best = -1, best_levels = 2, best_h = 0;
        long long best_c = 0;
        for (int levels = 2; levels <= 3; levels++) {
            p.tw_levels = levels;
            // two interleaved groups pay off for the column pass only (measured: row passes get slower)
            static const int groups_mask = FFT_EXP_ENV("FFT_HIP_GROUPS") ? atoi(FFT_EXP_ENV("FFT_HIP_GROUPS")) : 0;  // bit0 column pass, bit1 row passes
            const int want = (p.loadm == fftk::LOAD_CCONTIG) ? (groups_mask & 1) : (groups_mask & 2);
            for (int lh = (p.E == 8 && !no_groups && want) ? 1 : 0; lh >= 0; lh--) {
                int cand = -1;
                for (int lc = log2V; (1ll << (lc + lh)) <= extent_cols || (lc == log2V && lh == 0); lc++) {
                    const long long threads = (1ll << (lc - log2V)) << (p.log2L - log2E);
                    if (threads > (p.E == 4 ? 1024 : 512)) break;
                    if (lds_bytes(p, lc, lh) > budget) break;
                    cand = lc;
                    if ((1ll << (lc + lh)) * SZ >= 512 && threads >= 256) break;
                }
                if (cand >= 0 && (1ll << (cand + lh)) > best_c) {
                    best = cand; best_h = lh; best_levels = levels; best_c = 1ll << (cand + lh);
                }
            }
            if (!p.twiddle) break;
        }
        if (best < 0) return false;
        p.tw_levels = best_levels;
        p.log2H = best_h;
        p.log2C = best + best_h;
        p.nthreads = (int)((1ll << (best - log2V)) << (p.log2L - log2E));
        p.smem_bytes = lds_bytes(p, best, best_h);
        long long seg = (1ll << p.log2C) * SZ;
        p.seg_bytes = (int)(seg > 1 << 20 ? 1 << 20 : seg);
        return true;
    }

    // Relative time of a pass vs a pass with >= 512-byte row segments, measured on MI355X:
    // strided READS of narrow segments cost more than strided writes; 32-byte stores are ruinous.
    static double seg_cost(int seg_bytes) {  // column pass: strided read + strided (or tile-major) write
        if (seg_bytes >= 512) return 1.0;
        if (seg_bytes >= 256) return 1.07;
        if (seg_bytes >= 128) return 1.27;
        if (seg_bytes >= 64) return 1.57;
        if (seg_bytes >= 32) return 2.1;
        return 4.0;
    }
    static double store_cost(int seg_bytes) {  // row pass: contiguous read, strided store
        if (seg_bytes >= 64) return 1.0;
        if (seg_bytes >= 32) return 4.3;
        return 8.0;
    }

    // Plans that run forward + inverse back to back (Bluestein, the fused consumers) set this before build(): a split whose
    // first and last pass share a tile lets the middle two passes of the pair run as ONE kernel (execute_chain), which is
    // worth more than the wider row segments of the split the cost model would pick for a single transform
    // (tools/ab_chain.py, profiles/r2_ab_chain.txt).  The bonus is in the cost model's unit (1 = one pass).
    bool prefer_chain = false;
    // Two-pass plans with unequal factors (odd powers of two): the INVERSE of a forward / inverse pair runs on the mirrored split
    // L2 x L1, whose first pass has the tile of this plan's last pass -- so the pair chains too (execute_chain).  `mirror` is a
    // plan of its own (tables only: no scratch, no team kernel) built with the split forced.
    Pow2Plan* mirror = nullptr;
    int force_l1 = 0;        // > 0: two-pass split with this log2 L1 only (the mirror plan)
    bool tables_only = false;  // no scratch image, no team kernel (the mirror plan: its passes run on the owner's images)
    bool wants_hooks = false;  // the plan's owner fuses element-wise work into the first load / last store (execute_hooked)
    static bool ends_chainable(const PassDesc& a, const PassDesc& b) {
        return a.log2L == b.log2L && a.log2C == b.log2C && a.E == 8 && b.E == 8 && a.nthreads == b.nthreads && a.nthreads <= 512 &&
               a.log2H == 0 && b.log2H == 0;
    }
    double chain_bonus(const PassDesc& a, const PassDesc& b) const {
        static const double bonus = FFT_EXP_ENV("FFT_HIP_CHAIN_BONUS") ? atof(FFT_EXP_ENV("FFT_HIP_CHAIN_BONUS")) : 0.4;
        return (prefer_chain && ends_chainable(a, b)) ? bonus : 0.0;
    }

    bool build(RT* runtime, int log2n_, int algo_, int batch) {
        rt = runtime;
        log2n = log2n_;
        max_batch = batch;
        algo = algo_ == ALGO_AUTO ? ALGO_SPLIT_RADIX : algo_;
        if (algo == ALGO_RADIX2_SHFL && (log2n < 7 || log2n > 10)) algo = ALGO_RADIX2;  // one-wave kernel covers 128..1024
        fam = algo == ALGO_RADIX2 ? fftk::FAM_R2 : algo == ALGO_RADIX4 ? fftk::FAM_R4 : fftk::FAM_SR16;
        const long long n = 1ll << log2n;
        if (algo == ALGO_RADIX2_GLOBAL || algo == ALGO_RADIX2_SHFL) {
            if (log2n >= 1) {
                std::vector<cpx<T>> t;
                make_twiddle_table<T>(t, n, n / 2, 1);
                tw_half = (cpx<T>*)rt->dmalloc(t.size() * SZ);
                if (!tw_half) return false;
                rt->h2d(tw_half, t.data(), t.size() * SZ);
            }
            chunk = batch;
            ok = true;
            return true;
        }
        if (log2n == 0) { ok = true; chunk = batch; return true; }

        const int budget = rt->max_lds_bytes();
        std::vector<PassDesc> best;
        double best_cost = 1e30;
        // Cost model = relative HBM time: every pass reads and writes the data once (0.5 + 0.5); a
        // c-contiguous side whose row segments are shorter than 128 B pays seg_cost(); fewer passes win.
        int force[3] = {0, 0, 0};
        int n_force = 0;
        if (const char* e = FFT_EXP_ENV("FFT_HIP_FORCE_SPLIT")) n_force = sscanf(e, "%d,%d,%d", &force[0], &force[1], &force[2]);
        if (n_force > 0 && force[0] + force[1] + force[2] != log2n) n_force = 0;
        if (force_l1 > 0 && force_l1 < log2n) { n_force = 2; force[0] = force_l1; force[1] = log2n - force_l1; force[2] = 0; }

        // ---- candidate: single pass (rows in, rows out)
        if (n_force <= 1) {
            PassDesc p;
            p.log2L = log2n;
            p.E = tile_E(n);
            // from n = 512 up eight elements per thread and radix-8 stages (two exchanges instead of four or five) beat the
            // 1024-thread radix-4 kernel: fp32 +14...26 % with the shape baked in, fp64 +8...17 % (profiles/r2_ab_rows_fixed.txt).
            // Plans that fuse element-wise work into their ends keep the E = 4 kernel: that is the hooked instantiation.
            static const bool e_forced = FFT_EXP_ENV("FFT_HIP_E") != nullptr;
            if (!e_forced && !wants_hooks && algo_ == ALGO_AUTO && log2n >= 9) p.E = 8;
            p.loadm = fftk::LOAD_LCONTIG;
            p.storem = fftk::STORE_LCONTIG;
            p.in_c = n; p.in_l = 1; p.out_c = n; p.out_k = 1;
            p.n_cols = -1;
            // fp64, n >= 2048: no LDS staging.  One complex128 is one 16-byte access, so the eight elements l = r + (n/8) e a
            // thread owns can be loaded and stored directly; the lanes of one row cover 512-byte (n = 2048) / 1 KiB (4096) runs,
            // the "columns" of the tile are whole transforms n apart (TileParams::col_stride).  Measured (profiles/r2_ab_rows_fixed.txt):
            // n = 2048: 130 -> 148, 4096: 125 -> 145 Gpoint/s; n = 1024 (256-byte runs): 132 -> 118, stays staged
            static const int direct64 = FFT_EXP_ENV("FFT_HIP_DIRECT64") ? atoi(FFT_EXP_ENV("FFT_HIP_DIRECT64")) : 1;
            if (direct64 && SZ == 16 && p.E == 8 && log2n >= 11) {
                p.loadm = fftk::LOAD_CCONTIG;
                p.storem = fftk::STORE_CCONTIG;
                p.col_stride = n;
            }
            long long ext = 1;
            while (ext < batch) ext <<= 1;
            if (ext < V) ext = V;
            if (choose_tile(p, ext, budget)) {
                best.assign(1, p);
                best_cost = 1.0;
            }
        }
        // ---- candidates: two passes n = L1 * L2
        for (int l1 = 4; l1 <= log2n - 4; l1++) {
            const int l2 = log2n - l1;
            if (n_force >= 1 && !(n_force == 2 && force[0] == l1)) continue;
            PassDesc a, b;
            // W_n^(k1 n2) is applied by the column pass (A) to its results; applying it in the row pass (B) at load was
            // measured slower for BOTH kernels (0.78 + 0.59 vs 0.69 + 0.51 ms) -- kept behind FFT_HIP_TWIDDLE_IN_B
            static const int tw_in_b = FFT_EXP_ENV("FFT_HIP_TWIDDLE_IN_B") ? 1 : 0;
            a.log2L = l1; a.E = tile_E(1ll << l1, 1); a.loadm = fftk::LOAD_CCONTIG; a.storem = fftk::STORE_CCONTIG; a.twiddle = tw_in_b ? 0 : 1; a.log2Ntw = log2n;
            b.twiddle = tw_in_b ? 1 : 0; b.log2Ntw = log2n;
            a.in_b = n; a.in_c = 1; a.in_l = 1ll << l2; a.out_b = n; a.out_c = 1; a.out_k = 1ll << l2;
            a.n_cols = 1 << l2;
            if (!choose_tile(a, 1ll << l2, budget)) continue;
            a.n_ct = (1 << l2) >> a.log2C;
            b.log2L = l2; b.E = tile_E(1ll << l2, 2); b.loadm = fftk::LOAD_LCONTIG; b.storem = fftk::STORE_CCONTIG;
            b.in_b = n; b.in_c = 1ll << l2; b.in_l = 1; b.out_b = n; b.out_c = 1; b.out_k = 1ll << l1;
            b.n_cols = 1 << l1;
            if (!choose_tile(b, 1ll << l1, budget)) continue;
            b.n_ct = (1 << l1) >> b.log2C;
            if (!FFT_EXP_ENV("FFT_HIP_NO_TILE_MAJOR")) {
                // Tile-major scratch: pass A stores each tile (L1 rows x C_A columns) as ONE contiguous block, pass B
                // gathers its rows from the n2/C_A blocks in chunks of C_B*C_A contiguous elements.  The scratch
                // layout is ours to choose; this turns pass A's strided write into a linear one.
                const long long CA = 1ll << a.log2C;
                a.out_k = CA;              // row k1 of the tile
                a.out_c = 1ll << l1;       // tile base = ct*C_A * L1  (tile_coord multiplies c0 = ct*C_A by out_c)
                b.in_c = CA;               // row k1 inside a block
                b.in_blk_bits = a.log2C;   // n2 -> block n2 / C_A, column n2 % C_A
                b.in_blk_stride = CA << l1;
            }
            // where tools/split_sweep.py (profiles/r2_split_sweep.txt) measured another two-pass split 3...8 % faster than the model's
            // pick on MI355X, the measurement wins: 2^17 = 256 x 512, 2^18 = 512 x 512, 2^19 = 1024 x 512 (both precisions)
            const double measured = ((log2n == 17 && l1 == 8) || (log2n == 18 && l1 == 9) || (log2n == 19 && l1 == 10)) ? 0.4 : 0.0;
            const double cost = seg_cost(a.seg_bytes) + 0.5 + 0.5 * store_cost(b.seg_bytes) + 0.01 * abs(l1 - l2) - chain_bonus(a, b) - measured;
            if (cost < best_cost) {
                best_cost = cost;
                best.clear();
                best.push_back(a);
                best.push_back(b);
            }
        }
        // ---- candidates: three passes n = L1 * L2 * L3
        if (log2n >= 12) {
            for (int l1 = 4; l1 <= log2n - 8; l1++)
                for (int l2 = 4; l2 <= log2n - l1 - 4; l2++) {
                    const int l3 = log2n - l1 - l2;
                    if (l1 > 12 || l2 > 12 || l3 > 12) continue;
                    if (n_force >= 1 && !(n_force == 3 && force[0] == l1 && force[1] == l2)) continue;
                    const long long M = 1ll << (l2 + l3);
                    PassDesc a, m, b;
                    a.log2L = l1; a.E = tile_E(1ll << l1, 1); a.loadm = fftk::LOAD_CCONTIG; a.storem = fftk::STORE_CCONTIG; a.twiddle = 1; a.log2Ntw = log2n;
                    a.in_b = n; a.in_c = 1; a.in_l = M; a.out_b = n; a.out_c = 1; a.out_k = M;
                    a.n_cols = (int)M;
                    if (!choose_tile(a, M, budget)) continue;
                    a.n_ct = (int)(M >> a.log2C);
                    m.log2L = l2; m.E = tile_E(1ll << l2, 1); m.loadm = fftk::LOAD_CCONTIG; m.storem = fftk::STORE_CCONTIG; m.twiddle = 1; m.log2Ntw = l2 + l3;
                    m.n_b_per_transform = 1ll << l1;
                    m.in_b = M; m.in_c = 1; m.in_l = 1ll << l3; m.out_b = M; m.out_c = 1; m.out_k = 1ll << l3;
                    m.n_cols = 1 << l3;
                    if (!choose_tile(m, 1ll << l3, budget)) continue;
                    m.n_ct = (1 << l3) >> m.log2C;
                    b.log2L = l3; b.E = tile_E(1ll << l3, 2); b.loadm = fftk::LOAD_LCONTIG; b.storem = fftk::STORE_CCONTIG;
                    b.n_o = 1 << l2;
                    b.in_b = n; b.in_o = 1ll << l3; b.in_c = M; b.in_l = 1;
                    b.out_b = n; b.out_o = 1ll << l1; b.out_c = 1; b.out_k = 1ll << (l1 + l2);
                    b.n_cols = 1 << l1;
                    if (!choose_tile(b, 1ll << l1, budget)) continue;
                    b.n_ct = (1 << l1) >> b.log2C;
                    const double cost = seg_cost(a.seg_bytes) + seg_cost(m.seg_bytes) + 0.5 + 0.5 * store_cost(b.seg_bytes) +
                                        0.01 * (abs(l1 - l2) + abs(l2 - l3)) - chain_bonus(a, b);
                    if (cost < best_cost) {
                        best_cost = cost;
                        best.clear();
                        best.push_back(a);
                        best.push_back(m);
                        best.push_back(b);
                    }
                }
        }
        if (best.empty()) return false;
        passes = best;
        // Butterfly family per pass.  An explicit algo applies to every pass.  AUTO picks what measured fastest on
        // MI355X per pass shape: rows-in/rows-out and column passes run radix-4 stages, the
        // row pass with transposed store runs radix-8 (split-radix codelet) stages.
        {
            // column passes: radix-8 since the instantiations with L and C baked in exist for it too (profiles/r2_ab_column_family.txt:
            // fp32 +2...12 % at every multi-pass size, fp64 from 2^18 up; fp64 below: radix-4 stays 2 % ahead)
            const int col_fam = (SZ == 8 || log2n >= 18) ? fftk::FAM_SR16 : fftk::FAM_R4;
            int auto_fams[3] = {fftk::FAM_R4, col_fam, fftk::FAM_SR16};  // single-pass, column pass, row pass
            if (const char* e = FFT_EXP_ENV("FFT_HIP_AUTO_FAMS")) sscanf(e, "%d,%d,%d", &auto_fams[0], &auto_fams[1], &auto_fams[2]);
            for (auto& p : passes) {
                if (algo_ != ALGO_AUTO) p.fam = fam;
                else if (passes.size() == 1 && p.E == 8 && !FFT_EXP_ENV("FFT_HIP_AUTO_FAMS")) p.fam = fftk::FAM_SR16;  // single pass, E = 8 (with or without staging)
                else if (p.loadm == fftk::LOAD_CCONTIG) p.fam = auto_fams[1];
                else if (p.storem == fftk::STORE_CCONTIG) p.fam = auto_fams[2];
                else p.fam = (p.E == 8 && !FFT_EXP_ENV("FFT_HIP_AUTO_FAMS")) ? fftk::FAM_SR16 : auto_fams[0];
            }
        }

        if (!make_pass_tables()) return false;

        // ---- Infinity-Cache blocking of the batch (multi-pass only)
        chunk = batch;
        if (passes.size() > 1) {
            long long target = 1024ll << 20;  // scratch bytes per launch group (FFT_HIP_CHUNK_MB); see DESIGN.md on the Infinity Cache
            if (rt->policy.chunk_mb > 0) target = rt->policy.chunk_mb << 20;
            long long per = n * SZ;
            long long c = target / per;
            if (c < 1) c = 1;
            if (c > batch) c = batch;
            chunk = (int)c;
            scratch_bytes = (size_t)chunk * (size_t)per;
            if (!tables_only) {
                scratch = (cpx<T>*)rt->dmalloc(scratch_bytes);
                if (!scratch) return false;
            }
            // forward + inverse plans whose middle passes chain: the second image is allocated with the plan, not in the first
            // execute (a failed allocation only means the unchained path runs: execute_chain tries once more, then reports false)
            if (prefer_chain && algo_ == ALGO_AUTO && passes.size() == 2 && !ends_chainable(passes.front(), passes.back()) &&
                passes[0].log2L != passes[1].log2L && !tables_only) {
                mirror = new Pow2Plan<T, RT>();
                mirror->force_l1 = passes[1].log2L;
                mirror->tables_only = true;
                mirror->wants_hooks = true;
                if (!mirror->build(rt, log2n, ALGO_AUTO, batch) || mirror->passes.size() != 2 || mirror->chunk != chunk ||
                    !ends_chainable(mirror->passes.front(), passes.back())) {
                    delete mirror;
                    mirror = nullptr;
                }
            }
            if (prefer_chain && algo_ == ALGO_AUTO && !tables_only && (mirror || ends_chainable(passes.front(), passes.back()))) scratch2 = (cpx<T>*)rt->dmalloc(scratch_bytes);
        }
        if (algo_ == ALGO_AUTO && !wants_hooks && !tables_only) build_wide();
        // (plans whose owner fuses element-wise work into the ends -- Bluestein, the fused consumers -- never launch a team kernel:
        // execute_hooked / _chain / _round run the multi-pass kernels; they do not hold its tables and windows either)
        if (passes.size() > 1 && algo_ == ALGO_AUTO && !tables_only && !wants_hooks) {
            build_team_quad(batch);
            if (!team.ok) build_team(batch);
            if (team.ok && !team_geometry_is_built()) {
                rt->dfree(team.tables); rt->dfree(team.scratch); rt->dfree(team.sticky);
                team = TeamDesc<T>();
            }
        }
        ok = true;
        return true;
    }

    // The COLUMN transforms of a row-major rows x cols matrix (2D transforms; reference model: the column loop of
    // applications/image_fft.c:43-60, there an extract / transform / put-back per column): one in-place column pass of the
    // four-step engine -- cols-strided sub-transforms of length rows, `cols` of them per matrix, no inter-pass twiddle.
    // Needs rows = 2^log2rows to fit one LDS tile and cols to be a multiple of the 16-byte lane access; false otherwise
    // (the caller then goes through a transpose).  execute(in, out, n_matrices, inverse) transforms whole matrices.
    // one table blob per pass, laid out exactly as the kernel keeps it in LDS: stage tables [sa | sb], then the two- / three-level
    // inter-pass twiddle tables of W_(2^log2Ntw) where the pass applies one
    bool make_pass_tables() {
        for (auto& p : passes) {
            layout_tables(p);
            std::vector<cpx<T>> blob((size_t)p.tables_elems), part;
            for (auto& z : blob) { z.re = (T)1; z.im = (T)0; }
            const long long L = 1ll << p.log2L;
            make_twiddle_table<T>(part, L, 1ll << p.sa_bits, 1);
            std::copy(part.begin(), part.end(), blob.begin());
            make_twiddle_table<T>(part, L, 1ll << (p.log2L - p.sa_bits), 1ll << p.sa_bits);
            std::copy(part.begin(), part.end(), blob.begin() + p.o_sb);
            if (p.twiddle) {
                const long long Ntw = 1ll << p.log2Ntw;
                make_twiddle_table<T>(part, Ntw, 1ll << p.t0_bits, 1);
                std::copy(part.begin(), part.end(), blob.begin() + p.o_t0);
                make_twiddle_table<T>(part, Ntw, 1ll << p.t1_bits, 1ll << p.t0_bits);
                std::copy(part.begin(), part.end(), blob.begin() + p.o_t1);
                make_twiddle_table<T>(part, Ntw, 1ll << p.t2_bits, 1ll << (p.t0_bits + p.t1_bits));
                std::copy(part.begin(), part.end(), blob.begin() + p.o_t2);
            }
            cpx<T>* d = (cpx<T>*)rt->dmalloc(blob.size() * SZ);
            if (!d) return false;
            rt->h2d(d, blob.data(), blob.size() * SZ);
            pass_tables.push_back(d);
        }
        return true;
    }

    bool build_columns(RT* runtime, int log2rows, int cols, int n_matrices) {
        rt = runtime;
        log2n = log2rows;
        max_batch = n_matrices;
        algo = ALGO_SPLIT_RADIX;
        if (log2rows < 1 || cols < V || (cols % V) != 0) return false;
        PassDesc a;
        a.log2L = log2rows; a.E = tile_E(1ll << log2rows, 1); a.loadm = fftk::LOAD_CCONTIG; a.storem = fftk::STORE_CCONTIG;
        a.twiddle = 0; a.log2Ntw = 0;
        const long long per_matrix = (long long)cols << log2rows;
        a.in_b = per_matrix; a.in_c = 1; a.in_l = cols; a.out_b = per_matrix; a.out_c = 1; a.out_k = cols;
        a.n_cols = cols;
        long long ext = 1;
        while (ext < cols) ext <<= 1;
        if (!choose_tile(a, ext, rt->max_lds_bytes())) return false;
        a.n_ct = (int)((cols + (1ll << a.log2C) - 1) >> a.log2C);
        a.fam = fftk::FAM_R4;
        passes.assign(1, a);
        if (!make_pass_tables()) return false;
        chunk = n_matrices;
        ok = true;
        return true;
    }

    // The same column transforms in TWO passes with wide tiles, for row counts whose single column pass has narrow segments
    // (rows x C columns must fit one LDS tile: C = 2 at 4096 rows fp32) or does not fit at all: the four-step split
    // rows = L1 * L2 applied to strided columns.  With r = r1 L2 + r2 and k = k1 + L1 k2:
    //   pass A  for every r2 (the tile's o index): length-L1 transforms over the rows r1 L2 + r2, times W_rows^(k1 r2) (the twiddle
    //           index is o, TileParams::tw_o), stored in place of their inputs (row k1 L2 + r2);
    //   pass B  for every k1 (o): length-L2 transforms over the CONTIGUOUS rows k1 L2 + r2, stored to row k1 + L1 k2.
    // Both passes move cols-wide contiguous row segments; pass B changes rows, so the pair is never in place: execute() goes
    // in -> scratch -> out like every two-pass plan (in == out allowed).
    bool build_columns2(RT* runtime, int log2rows, int cols, int n_matrices) {
        rt = runtime;
        log2n = log2rows;
        max_batch = n_matrices;
        algo = ALGO_SPLIT_RADIX;
        if (log2rows < 6 || cols < V || (cols % V) != 0) return false;
        const int l1 = (log2rows + 1) / 2, l2 = log2rows - l1;
        const long long per_matrix = (long long)cols << log2rows;
        long long ext = 1;
        while (ext < cols) ext <<= 1;
        PassDesc a, b;
        a.log2L = l1; a.E = tile_E(1ll << l1, 1); a.loadm = fftk::LOAD_CCONTIG; a.storem = fftk::STORE_CCONTIG;
        a.twiddle = 1; a.log2Ntw = log2rows; a.tw_o = 1;
        a.in_b = per_matrix; a.out_b = per_matrix; a.in_c = 1; a.out_c = 1;
        a.n_o = 1 << l2; a.in_o = cols; a.out_o = cols;
        a.in_l = (long long)cols << l2; a.out_k = (long long)cols << l2;
        a.n_cols = cols;
        if (!choose_tile(a, ext, rt->max_lds_bytes())) return false;
        a.n_ct = (int)((cols + (1ll << a.log2C) - 1) >> a.log2C);
        a.fam = fftk::FAM_R4;
        b.log2L = l2; b.E = tile_E(1ll << l2, 1); b.loadm = fftk::LOAD_CCONTIG; b.storem = fftk::STORE_CCONTIG;
        b.twiddle = 0; b.log2Ntw = 0;
        b.in_b = per_matrix; b.out_b = per_matrix; b.in_c = 1; b.out_c = 1;
        b.n_o = 1 << l1; b.in_o = (long long)cols << l2; b.out_o = cols;
        b.in_l = cols; b.out_k = (long long)cols << l1;
        b.n_cols = cols;
        if (!choose_tile(b, ext, rt->max_lds_bytes())) return false;
        b.n_ct = (int)((cols + (1ll << b.log2C) - 1) >> b.log2C);
        b.fam = fftk::FAM_R4;
        passes.clear();
        passes.push_back(a);
        passes.push_back(b);
        if (!make_pass_tables()) return false;
        chunk = n_matrices;  // execute() steps through the batch in units of 2^log2n elements: one launch group = every matrix
        scratch_bytes = (size_t)n_matrices * (size_t)per_matrix * SZ;
        scratch = (cpx<T>*)rt->dmalloc(scratch_bytes);
        if (!scratch) return false;
        ok = true;
        return true;
    }

    template <int E, int FAM, int LM, int SM, bool TW>
    void launch_one(const fftk::TileParams<T>& tp, long long grid, const PassDesc& p) {
        if (E == 8 && p.log2H == 1) launch_one_h<E, (E == 8 ? 2 : 1), FAM, LM, SM, TW>(tp, grid, p);
        else launch_one_h<E, 1, FAM, LM, SM, TW>(tp, grid, p);
    }

    template <int E, int H, int FAM, int LM, int SM, bool TW>
    void launch_one_h(const fftk::TileParams<T>& tp, long long grid, const PassDesc& p) {
        // Hot shapes get an instantiation with L and C baked in (every shift / mask / LDS offset becomes an
        // immediate: fewer integer VALU ops and address VGPRs; measured +9 % at N = 2^20).  Shapes: the full
        // 64 KiB tile of L in {128, 256, 512, 1024} in AUTO's families, column pass and transposing row pass.
        constexpr bool HAS_FIX = (E == 8 && H == 1) &&
                                 ((LM == fftk::LOAD_CCONTIG && (FAM == fftk::FAM_R4 || FAM == fftk::FAM_SR16) && TW) ||
                                  (LM == fftk::LOAD_LCONTIG && SM == fftk::STORE_CCONTIG && FAM == fftk::FAM_SR16 && !TW));
        static const int use_fixed = FFT_EXP_ENV("FFT_HIP_FIXED") ? atoi(FFT_EXP_ENV("FFT_HIP_FIXED")) : 1;
        if (HAS_FIX && use_fixed) {
            const int full_c = 13 - ilog2(SZ / 8) - p.log2L;  // log2 of (8192 or 4096 elements) / L
            if (p.log2C == full_c) {
                switch (p.log2L) {
                    case 7: launch_fixed<E, H, FAM, LM, SM, TW, HAS_FIX, 7>(tp, grid, p); return;
                    case 8: launch_fixed<E, H, FAM, LM, SM, TW, HAS_FIX, 8>(tp, grid, p); return;
                    case 9: launch_fixed<E, H, FAM, LM, SM, TW, HAS_FIX, 9>(tp, grid, p); return;
                    case 10: launch_fixed<E, H, FAM, LM, SM, TW, HAS_FIX, 10>(tp, grid, p); return;
                    default: break;
                }
            }
        }
        // ... and the single-pass rows kernel (E = 4, radix-4) for n = 128, 256 fp32 (fft_rows_list.h says why only those)
        constexpr bool ROWS_FIX = E == 4 && H == 1 && LM == fftk::LOAD_LCONTIG && SM == fftk::STORE_LCONTIG && FAM == fftk::FAM_R4 && !TW && SZ == 8;
        if (ROWS_FIX && use_fixed && p.log2C == 13 - p.log2L) {
            switch (p.log2L) {
                case 7: launch_fixed<E, H, FAM, LM, SM, TW, ROWS_FIX, 7>(tp, grid, p); return;
                case 8: launch_fixed<E, H, FAM, LM, SM, TW, ROWS_FIX, 8>(tp, grid, p); return;
                default: break;
            }
        }
        // ... and its E = 8 radix-8 form, n = 512 ... 4096, fp32
        constexpr bool ROWS_FIX8 = E == 8 && H == 1 && LM == fftk::LOAD_LCONTIG && SM == fftk::STORE_LCONTIG && FAM == fftk::FAM_SR16 && !TW && SZ == 8;
        if (ROWS_FIX8 && use_fixed && p.log2C == 13 - p.log2L) {
            switch (p.log2L) {
                case 9: launch_fixed<E, H, FAM, LM, SM, TW, ROWS_FIX8, 9>(tp, grid, p); return;
                case 10: launch_fixed<E, H, FAM, LM, SM, TW, ROWS_FIX8, 10>(tp, grid, p); return;
                case 11: launch_fixed<E, H, FAM, LM, SM, TW, ROWS_FIX8, 11>(tp, grid, p); return;
                case 12: launch_fixed<E, H, FAM, LM, SM, TW, ROWS_FIX8, 12>(tp, grid, p); return;
                default: break;
            }
        }
        launch_kernel(fftk::tile_fft_kernel<T, E, H, FAM, LM, SM, TW, 0>, tp, grid, p);
    }

    template <int E, int H, int FAM, int LM, int SM, bool TW, bool HAS_FIX, int LOG2L>
    void launch_fixed(const fftk::TileParams<T>& tp, long long grid, const PassDesc& p) {
        constexpr int FIX = HAS_FIX ? ((LOG2L << 8) | (13 - (SZ == 16 ? 1 : 0) - LOG2L)) : 0;
        launch_kernel(fftk::tile_fft_kernel<T, E, H, FAM, LM, SM, TW, FIX>, tp, grid, p);
    }

    template <class K>
    void launch_kernel(K kernel, const fftk::TileParams<T>& tp, long long grid, const PassDesc& p) {
        if (grid < 0) {
            // persistent grid: exactly the workgroups that are resident at once (occupancy query: VGPRs, LDS, waves)
            int per_cu = rt->max_blocks_per_cu(kernel, p.nthreads, (size_t)p.smem_bytes);
            static const int force_per_cu = FFT_EXP_ENV("FFT_HIP_WG_PER_CU") ? atoi(FFT_EXP_ENV("FFT_HIP_WG_PER_CU")) : 0;  // experiments
            if (force_per_cu > 0) per_cu = force_per_cu;
            if (per_cu < 1) per_cu = 1;
            grid = (long long)rt->num_cus() * per_cu;
            if (grid > tp.n_tiles) grid = tp.n_tiles;
        }
        rt->launch(kernel, grid, p.nthreads, (size_t)p.smem_bytes, tp);
    }

    template <int FAM, int LM, int SM, bool TW>
    void launch_mode(const fftk::TileParams<T>& tp, long long grid, const PassDesc& p) {
        using namespace fftk;
        switch (p.E) {
            case 2: launch_one<2, FAM_R2, LM, SM, TW>(tp, grid, p); break;
            case 4: launch_one<4, (FAM == FAM_SR16 ? FAM_R4 : FAM), LM, SM, TW>(tp, grid, p); break;
            default: launch_one<8, FAM, LM, SM, TW>(tp, grid, p); break;
        }
    }

    template <int FAM, int H>
    void launch_fam(const fftk::TileParams<T>& tp, long long grid, const PassDesc& p) {
        using namespace fftk;
        if (p.loadm == LOAD_CCONTIG) {
            if (p.twiddle) launch_mode<FAM, LOAD_CCONTIG, STORE_CCONTIG, true>(tp, grid, p);
            else launch_mode<FAM, LOAD_CCONTIG, STORE_CCONTIG, false>(tp, grid, p);
        } else if (p.storem == STORE_CCONTIG) {
            if (p.twiddle) launch_mode<FAM, LOAD_LCONTIG, STORE_CCONTIG, true>(tp, grid, p);
            else launch_mode<FAM, LOAD_LCONTIG, STORE_CCONTIG, false>(tp, grid, p);
        } else {
            launch_mode<FAM, LOAD_LCONTIG, STORE_LCONTIG, false>(tp, grid, p);
        }
    }

    fftk::TileParams<T> pass_params(size_t ipass, const cpx<T>* in, cpx<T>* out, int nb, bool inverse, T scale) {
        const PassDesc& p = passes[ipass];
        fftk::TileParams<T> tp;
        memset(&tp, 0, sizeof(tp));
        tp.in = in;
        tp.out = out;
        tp.tables = pass_tables[ipass];
        tp.tables_bytes = p.tables_elems * SZ;
        tp.group_bytes = p.group_bytes;
        tp.off_tables = p.off_tables;
        tp.o_sb = p.o_sb; tp.o_t0 = p.o_t0; tp.o_t1 = p.o_t1; tp.o_t2 = p.o_t2;
        tp.sa_bits = p.sa_bits; tp.t0_bits = p.t0_bits; tp.t1_bits = p.t1_bits; tp.t2_bits = p.t2_bits;
        tp.log2L = p.log2L;
        tp.log2C = p.log2C;
        tp.n_ct = p.n_ct;
        tp.n_o = p.n_o;
        tp.in_b = p.in_b; tp.in_o = p.in_o; tp.in_c = p.in_c; tp.in_l = p.in_l;
        tp.out_b = p.out_b; tp.out_o = p.out_o; tp.out_c = p.out_c; tp.out_k = p.out_k;
        tp.in_blk_bits = p.in_blk_bits; tp.in_blk_stride = p.in_blk_stride;
        tp.inverse = inverse ? 1 : 0;
        tp.scale = scale;
        tp.tw_o = p.tw_o;
        tp.col_stride = p.col_stride;
        tp.run_if = run_if;
        static const int ablate = FFT_EXP_ENV("FFT_HIP_ABLATE") ? atoi(FFT_EXP_ENV("FFT_HIP_ABLATE")) : 0;  // profiling only
        tp.ablate = ablate & ~48;
        if (passes.size() == 2 && ipass == 0) tp.ablate |= (ablate & 16);  // experiment: pass A writes a wrapped (cache-sized) scratch
        if (passes.size() == 2 && ipass == 1) tp.ablate |= (ablate & 32);  // experiment: pass B reads it
        static const int pair16 = FFT_EXP_ENV("FFT_HIP_PAIR16") ? atoi(FFT_EXP_ENV("FFT_HIP_PAIR16")) : 0;
        // column pass with narrow (64-byte) row segments: walk the launch transform-fastest (measured 0.64 -> 0.57 ms
        // per 128 transforms at L = 1024; wider-segment shapes and row passes prefer the natural order)
        static const int order_a = FFT_EXP_ENV("FFT_HIP_ORDER_A") ? atoi(FFT_EXP_ENV("FFT_HIP_ORDER_A")) : -1;  // column pass; -1 = auto
        static const int order_b = FFT_EXP_ENV("FFT_HIP_ORDER_B") ? atoi(FFT_EXP_ENV("FFT_HIP_ORDER_B")) : 0;  // row pass
        tp.order_g = 0;
        tp.tiles_per_b = 1;
        if (p.n_cols >= 0 && p.n_b_per_transform == 1) {
            int g = p.loadm == fftk::LOAD_CCONTIG ? order_a : order_b;
            if (g < 0) {
                g = 0;
                if (p.seg_bytes < 128) { g = 1; while (g * 2 <= nb) g *= 2; }
            }
            while (g > 1 && (nb % g) != 0) g >>= 1;  // must divide the transforms of this launch
            tp.order_g = g;
            tp.tiles_per_b = p.n_o * p.n_ct;
        }
        tp.pair16 = (pair16 && ((1ll << p.log2C) * SZ < 128)) ? 1 : 0;
        {   // non-temporal hint on the HBM streams of plans whose every strided side moves whole 128-byte lines (TileParams::nt;
            // profiles/r2_ab_tile_nt.txt: +5 % at n = 64, +10 % on the two-pass 2^16, +2...5 % on 2^21, 2^22; a plan with 64-byte
            // row segments -- L = 1024 column passes -- loses 6...20 % with the hint on any of its passes and gets none)
            static const int nt_force = FFT_EXP_ENV("FFT_HIP_TILE_NT") ? atoi(FFT_EXP_ENV("FFT_HIP_TILE_NT")) : -1;
            bool wide = true;
            for (const PassDesc& q : passes)
                if (q.n_cols >= 0 && (1ll << q.log2C) * SZ < 128) wide = false;
            tp.nt = nt_force >= 0 ? nt_force : (wide ? 3 : 0);
        }
        if (p.n_cols < 0) {  // single-pass row kernel: columns are the transforms of the batch
            tp.n_cols = nb;
            tp.n_ct = (int)((nb + (1ll << p.log2C) - 1) >> p.log2C);
            tp.n_tiles = tp.n_ct;
        } else {
            tp.n_cols = p.n_cols;
            tp.n_tiles = (long long)nb * p.n_b_per_transform * p.n_o * p.n_ct;
        }
        return tp;
    }

    void launch_pass(size_t ipass, const cpx<T>* in, cpx<T>* out, int nb, bool inverse, T scale) {
        const PassDesc& p = passes[ipass];
        const fftk::TileParams<T> tp = pass_params(ipass, in, out, nb, inverse, scale);
        static const int nonpersistent = FFT_EXP_ENV("FFT_HIP_NONPERSISTENT") ? 1 : 0;
        long long grid = nonpersistent ? tp.n_tiles : -1;  // -1: launch_one sizes the persistent grid from the occupancy query
        switch (p.fam) {
            case fftk::FAM_R2: launch_fam<fftk::FAM_R2, 1>(tp, grid, p); break;
            case fftk::FAM_R4: launch_fam<fftk::FAM_R4, 1>(tp, grid, p); break;
            default: launch_fam<fftk::FAM_SR16, 1>(tp, grid, p); break;
        }
    }

    // ---- fused element-wise work (fftk::TileHooks): three HOOK instantiations cover AUTO's plans -- the single-pass
    // rows kernel, the first (column) pass and the last (transposing row) pass
    static int hook_kind(const PassDesc& p) {
        using namespace fftk;
        if (p.log2H != 0) return 0;
        if (p.loadm == LOAD_LCONTIG && p.storem == STORE_LCONTIG && p.E == 4 && p.fam == FAM_R4 && !p.twiddle) return 1;
        if (p.loadm == LOAD_CCONTIG && p.storem == STORE_CCONTIG && p.E == 8 && p.fam == FAM_R4 && p.twiddle) return 2;
        if (p.loadm == LOAD_LCONTIG && p.storem == STORE_CCONTIG && p.E == 8 && p.fam == FAM_SR16 && !p.twiddle) return 3;
        if (p.loadm == LOAD_CCONTIG && p.storem == STORE_CCONTIG && p.E == 8 && p.fam == FAM_SR16 && p.twiddle) return 4;
        return 0;
    }
    bool hook_capable() const {
        if (!ok || passes.empty() || log2n == 0) return false;
        if (algo == ALGO_RADIX2_GLOBAL || algo == ALGO_RADIX2_SHFL) return false;
        if (passes.size() == 1) return hook_kind(passes[0]) == 1;
        const int first = hook_kind(passes.front());
        return (first == 2 || first == 4) && hook_kind(passes.back()) == 3;
    }

    // the hooked column / transposing row pass with L and C baked in for the full 64 KiB tile of L = 128 ... 1024 (as launch_one_h)
    template <int FAM, int LM, int SM, bool TW, int HOOK>
    void launch_hooked_fixed(const fftk::TileParams<T>& tp, const PassDesc& p) {
        using namespace fftk;
        static const int use_fixed = FFT_EXP_ENV("FFT_HIP_FIXED") ? atoi(FFT_EXP_ENV("FFT_HIP_FIXED")) : 1;
        constexpr int D = SZ == 16 ? 1 : 0;
        // (the fp64 store-hooked row pass spills 23...42 VGPRs with its shape baked in: generic there)
        constexpr bool FIX_OK = !(SZ == 16 && (HOOK & 2));
        if constexpr (FIX_OK) {
          if (use_fixed && p.log2C == 13 - D - p.log2L) {
            switch (p.log2L) {
                case 7: launch_kernel(tile_fft_kernel<T, 8, 1, FAM, LM, SM, TW, (7 << 8) | (13 - D - 7), HOOK>, tp, -1, p); return;
                case 8: launch_kernel(tile_fft_kernel<T, 8, 1, FAM, LM, SM, TW, (8 << 8) | (13 - D - 8), HOOK>, tp, -1, p); return;
                case 9: launch_kernel(tile_fft_kernel<T, 8, 1, FAM, LM, SM, TW, (9 << 8) | (13 - D - 9), HOOK>, tp, -1, p); return;
                case 10: launch_kernel(tile_fft_kernel<T, 8, 1, FAM, LM, SM, TW, (10 << 8) | (13 - D - 10), HOOK>, tp, -1, p); return;
                default: break;
            }
          }
        }
        launch_kernel(tile_fft_kernel<T, 8, 1, FAM, LM, SM, TW, 0, HOOK>, tp, -1, p);
    }

    // side: bit 0 this launch carries the load side of `h`, bit 1 the store side
    void launch_pass_hooked(size_t ipass, const cpx<T>* in, cpx<T>* out, int nb, bool inverse, T scale, const ExecHooks<T>& h, int side,
                            long long tab_off) {
        using namespace fftk;
        const PassDesc& p = passes[ipass];
        TileParams<T> tp = pass_params(ipass, in, out, nb, inverse, scale);
        const long long n = 1ll << log2n;
        TileHooks<T>& k = tp.hk;
        k.n_in = 0x7fffffff; k.n_out = 0x7fffffff; k.in_vec_ok = 1; k.out_vec_ok = 1;
        if (side & 1) {
            const long long pitch = h.in_pitch ? h.in_pitch : n;
            k.pre_tab = h.pre_tab; k.pre_mode = h.pre_tab ? h.pre_mode : HOOK_NONE;
            k.n_in = (int)(h.n_in ? h.n_in : n);
            k.in_vec_ok = (pitch % V) == 0 ? 1 : 0;
            if (p.n_cols < 0) tp.in_c = pitch; else tp.in_b = pitch;
        }
        if (side & 2) {
            const long long pitch = h.out_pitch ? h.out_pitch : n;
            k.post_tab = h.post_tab ? h.post_tab + tab_off : nullptr;
            k.post_tab_b = h.post_tab_b;
            k.post_mode = (h.post_tab || h.post_mode == HOOK_ABS2) ? h.post_mode : HOOK_NONE;
            k.n_out = (int)(h.n_out ? h.n_out : n);
            k.out_vec_ok = (pitch % V) == 0 ? 1 : 0;
            if (p.n_cols < 0) tp.out_c = pitch; else tp.out_b = pitch;
        }
        if (side & 4) {  // FFT -> mid product -> inverse FFT in this one launch (single-pass plans: execute_round)
            k.mid_tab = h.mid_tab;
            k.mid_mode = (h.mid_tab || h.mid_mode == HOOK_ABS2) ? h.mid_mode : HOOK_NONE;
            if (hook_kind(p) == 1) launch_kernel(tile_fft_kernel<T, 4, 1, FAM_R4, LOAD_LCONTIG, STORE_LCONTIG, false, 0, 3 | 8>, tp, -1, p);
            return;
        }
        switch (hook_kind(p)) {
            // HOOK bits: 1 load side, 2 store side, 4 table values prefetched with the data, 8 round trip (fft_kernels.h)
            case 1: launch_kernel(tile_fft_kernel<T, 4, 1, FAM_R4, LOAD_LCONTIG, STORE_LCONTIG, false, 0, 3>, tp, -1, p); break;
            case 2: launch_kernel(tile_fft_kernel<T, 8, 1, FAM_R4, LOAD_CCONTIG, STORE_CCONTIG, true, 0, 1 | 4>, tp, -1, p); break;
            case 3: launch_hooked_fixed<FAM_SR16, LOAD_LCONTIG, STORE_CCONTIG, false, 2>(tp, p); break;
            case 4: launch_hooked_fixed<FAM_SR16, LOAD_CCONTIG, STORE_CCONTIG, true, 1 | 4>(tp, p); break;
            default: break;
        }
    }

    // ---- single-pass sizes: out = IFFT( mid( FFT( pre(in) ) ) ) * post as ONE kernel (TileHooks::mid_tab): the transform pair
    // of a small Bluestein or convolution costs one HBM round trip of the user's data, nothing else.  The mid table is shared
    // by the batch (h.mid_tab) or absent (HOOK_ABS2); per-transform products (cross-correlation) take two kernels.
    bool round_capable() const { return hook_capable() && passes.size() == 1 && hook_kind(passes[0]) == 1; }
    void execute_round(const cpx<T>* in, cpx<T>* out, int nb, const ExecHooks<T>& h, T extra_scale = (T)1) {
        const long long n = 1ll << log2n;
        const T scale = (T)((1.0L / (long double)n) * (long double)extra_scale);
        run_if = nullptr;
        launch_pass_hooked(0, in, out, nb, false, scale, h, 3 | 4, 0);
        rt->mark(0);
    }

    // ---- forward transform -> spectral product -> inverse transform with the forward's LAST pass and the inverse's FIRST
    // pass as one kernel (fft_kernels_chain.h): possible when both have the same tile
    const Pow2Plan* inv_plan() const { return mirror ? mirror : this; }  // whose passes run the inverse half of execute_chain
    Pow2Plan* inv_plan() { return mirror ? mirror : this; }
    int chain_smem() const {
        const PassDesc &a = inv_plan()->passes.front(), &b = passes.back();
        const int data = (std::max(a.group_bytes, b.group_bytes) + 15) & ~15;
        return data + (a.tables_elems + b.tables_elems) * SZ + 16;
    }
    static constexpr int kChainMinLog2n = 0;  // measured (tools/ab_chain.py, profiles/r2_ab_chain.txt): chaining wins at every size it applies to (2^14 ... 2^21: +7...27 %)
    int chain_min_log2n = kChainMinLog2n;
    bool chain_capable() const {
        if (!hook_capable() || passes.size() < 2 || log2n < chain_min_log2n) return false;
        if (mirror && !mirror->hook_capable()) return false;
        const PassDesc &a = inv_plan()->passes.front(), &b = passes.back();
        return a.log2L == b.log2L && a.log2C == b.log2C && a.nthreads == b.nthreads && a.nthreads <= 512 && a.log2H == 0 && b.log2H == 0 && chain_smem() <= rt->max_lds_bytes();
    }
    void launch_chain(const cpx<T>* in, cpx<T>* out, int nb, const ExecHooks<T>& h, long long tab_off) {
        using namespace fftk;
        const size_t last = passes.size() - 1;
        const PassDesc &a = inv_plan()->passes.front(), &b = passes.back();
        ChainParams<T> q;
        q.b = pass_params(last, in, nullptr, nb, false, (T)1);
        q.a = inv_plan()->pass_params(0, nullptr, out, nb, true, (T)1);
        const int data = (std::max(a.group_bytes, b.group_bytes) + 15) & ~15;
        q.b.off_tables = data;
        q.off_tables_a = data + b.tables_elems * SZ;
        TileHooks<T>& k = q.b.hk;
        k.n_in = 0x7fffffff; k.n_out = 0x7fffffff; k.in_vec_ok = 1; k.out_vec_ok = 1;
        k.post_tab = h.post_tab ? h.post_tab + tab_off : nullptr;
        k.post_tab_b = h.post_tab_b;
        k.post_mode = (h.post_tab || h.post_mode == HOOK_ABS2) ? h.post_mode : HOOK_NONE;
        // the full 64 KiB tile of L in {128 ... 1024} gets an instantiation with L and C baked in (as launch_one_h does)
        const int full_c = 13 - ilog2(SZ / 8) - a.log2L;
        static const int use_fixed = FFT_EXP_ENV("FFT_HIP_FIXED") ? atoi(FFT_EXP_ENV("FFT_HIP_FIXED")) : 1;
        if (use_fixed && a.log2C == full_c) {
            switch (a.log2L) {
                case 7: launch_chain_kernel(tile_fft_ba_kernel<T, ((7 << 8) | (13 - (SZ == 16 ? 1 : 0) - 7))>, q, a.nthreads); return;
                case 8: launch_chain_kernel(tile_fft_ba_kernel<T, ((8 << 8) | (13 - (SZ == 16 ? 1 : 0) - 8))>, q, a.nthreads); return;
                case 9: launch_chain_kernel(tile_fft_ba_kernel<T, ((9 << 8) | (13 - (SZ == 16 ? 1 : 0) - 9))>, q, a.nthreads); return;
                case 10: launch_chain_kernel(tile_fft_ba_kernel<T, ((10 << 8) | (13 - (SZ == 16 ? 1 : 0) - 10))>, q, a.nthreads); return;
                default: break;
            }
        }
        launch_chain_kernel(tile_fft_ba_kernel<T, 0>, q, a.nthreads);
    }
    template <class K>
    void launch_chain_kernel(K kernel, const fftk::ChainParams<T>& q, int nthreads) {
        const int smem = chain_smem();
        int per_cu = rt->max_blocks_per_cu(kernel, nthreads, (size_t)smem);
        if (per_cu < 1) per_cu = 1;
        long long grid = (long long)rt->num_cus() * per_cu;
        if (grid > q.b.n_tiles) grid = q.b.n_tiles;
        rt->launch(kernel, grid, nthreads, (size_t)smem, q);
    }
    // out = IFFT( post_f( FFT( pre_f(in) ) ) ) with the ends of `hf` (load side + spectral product) and `hi` (store side of
    // the inverse); requires chain_capable().  One HBM round trip less than execute_hooked twice.
    bool execute_chain(const cpx<T>* in, cpx<T>* out, int nb, const ExecHooks<T>& hf, const ExecHooks<T>& hi, T extra_scale = (T)1) {
        const long long n = 1ll << log2n;
        if (!scratch2) {
            scratch2 = (cpx<T>*)rt->dmalloc(scratch_bytes);
            if (!scratch2) return false;
        }
        const T scale = (T)((1.0L / (long double)n) * (long double)extra_scale);
        const long long ip = hf.in_pitch ? hf.in_pitch : n, op = hi.out_pitch ? hi.out_pitch : n;
        const size_t last = passes.size() - 1;
        run_if = nullptr;
        for (int b0 = 0; b0 < nb; b0 += chunk) {
            const int cb = (nb - b0) < chunk ? (nb - b0) : chunk;
            launch_pass_hooked(0, in + (size_t)b0 * (size_t)ip, scratch, cb, false, (T)1, hf, 1, 0);
            rt->mark(0);
            if (passes.size() == 3) launch_pass(1, scratch, scratch, cb, false, (T)1);
            launch_chain(scratch, scratch2, cb, hf, (long long)b0 * hf.post_tab_b);
            rt->mark(1);
            if (passes.size() == 3) launch_pass(1, scratch2, scratch2, cb, true, (T)1);
            inv_plan()->run_if = nullptr;
            inv_plan()->launch_pass_hooked(last, scratch2, out + (size_t)b0 * (size_t)op, cb, true, scale, hi, 2, (long long)b0 * hi.post_tab_b);
            rt->mark(2);
        }
        return true;
    }

    // execute() with fused element-wise ends; requires hook_capable().  Never the team kernel (it has no hooks).
    void execute_hooked(const cpx<T>* in, cpx<T>* out, int nb, bool inverse, const ExecHooks<T>& h, T extra_scale = (T)1) {
        const long long n = 1ll << log2n;
        const T scale = (T)((inverse ? 1.0L / (long double)n : 1.0L) * (long double)extra_scale);
        const long long ip = h.in_pitch ? h.in_pitch : n, op = h.out_pitch ? h.out_pitch : n;
        run_if = nullptr;
        if (passes.size() == 1) {
            launch_pass_hooked(0, in, out, nb, inverse, scale, h, 3, 0);
            rt->mark(0);
            return;
        }
        const size_t last = passes.size() - 1;
        for (int b0 = 0; b0 < nb; b0 += chunk) {
            const int cb = (nb - b0) < chunk ? (nb - b0) : chunk;
            const cpx<T>* src = in + (size_t)b0 * (size_t)ip;
            cpx<T>* dst = out + (size_t)b0 * (size_t)op;
            launch_pass_hooked(0, src, scratch, cb, inverse, (T)1, h, 1, 0);
            rt->mark(0);
            if (passes.size() == 3) {
                launch_pass(1, scratch, scratch, cb, inverse, (T)1);
                rt->mark(1);
            }
            launch_pass_hooked(last, scratch, dst, cb, inverse, scale, h, 2, (long long)b0 * h.post_tab_b);
            rt->mark((int)last);
        }
    }

    static long long grid_for(long long total, int block) {
        long long g = (total + block - 1) / block;
        if (g > 16384) g = 16384;
        if (g < 1) g = 1;
        return g;
    }

    // Transform `nb` contiguous transforms (nb <= max_batch); in == out allowed.
    // The host has synchronized and read the team kernel's status (HIP: team_status_of; the emulation: its test driver).
    // ok: forget the log.  TIMEOUT: a formed team stopped making progress (the kernel's waits are bounded, the launch has
    // ended) -- its results are invalid.  The team kernel is switched off for this plan and every logged OUT-OF-PLACE execute
    // since the last sync is replayed on the multi-pass schedule (the inputs are intact); returns the number of executes that
    // could NOT be repaired (in place: part of the input is gone).  The caller synchronizes again afterwards.
    int recover_after_timeout(bool timed_out) {
        int lost = 0;
        if (timed_out) {
            team.ok = false;  // (the buffers stay allocated until the plan is destroyed)
            team_disabled = true;
            std::vector<TeamExec> log;
            log.swap(team_log);
            // A logged execute is repeated only where that is PROVABLY right (round-3 ADVICE): its input still holds what the caller put
            // there.  Every launch since the last sync is suspect -- a broken team has written part of its output, and a launch fed by a
            // suspect buffer has written garbage everywhere -- so an input is intact only if NO other logged execute wrote into it (a
            // later one: the ping-pong A -> B, B -> A overwrites A before the replay reads it) unless that writer came EARLIER and has
            // itself been repeated.  In place: only from the staged copy.  Executes beyond the log's capacity, and everything when the
            // caller has not promised to keep its buffers until the sync (team_replay), are lost.
            const size_t nbytes = (size_t)SZ << log2n;
            auto overlap = [&](const void* a, size_t abytes, const void* b, size_t bbytes) {
                const char *pa = (const char*)a, *pb = (const char*)b;
                return pa < pb + bbytes && pb < pa + abytes;
            };
            std::vector<char> bad(log.size(), 0);  // the execute's output holds invalid data
            for (size_t k = 0; k < log.size(); k++) {
                const TeamExec& e = log[k];
                const size_t bytes = nbytes * (size_t)e.nb;
                bool ok = team_replay && team_log_dropped == 0 && !overlap(e.in, bytes, e.out, bytes);
                for (size_t j = 0; ok && j < log.size(); j++) {
                    if (j == k) continue;
                    const size_t jb = nbytes * (size_t)log[j].nb;
                    if (!overlap(e.in, bytes, log[j].out, jb)) continue;
                    if (j > k || bad[j]) ok = false;  // written later, or by an execute that could not be repeated
                }
                // a staged copy is this execute's alone only if nothing was staged over it since
                if (ok && e.staged && (k + 1 < log.size())) {
                    for (size_t j = k + 1; ok && j < log.size(); j++)
                        if (log[j].staged && overlap(e.in, bytes, log[j].in, nbytes * (size_t)log[j].nb)) ok = false;
                }
                if (!ok) { bad[k] = 1; lost++; continue; }
                execute(e.in, e.out, e.nb, e.inverse, e.scale_inverse);
            }
            lost += team_log_dropped;
            team_unrecoverable += lost;
        }
        team_log.clear();
        team_log_dropped = 0;
        team_stage_used = 0;
        return lost;
    }
    // a sync has read status OK for team launches of this plan: the teams form and their waits end on this device -- later
    // in-place executes run without the staged copy
    void team_seen_ok() {
        team_proven = true;
        if (team_stage) rt->dfree(team_stage);
        team_stage = nullptr;
        team_stage_bytes = team_stage_used = 0;
    }

    void execute(const cpx<T>* in, cpx<T>* out, int nb, bool inverse, bool scale_inverse = true) {
        const long long n = 1ll << log2n;
        const T scale = (inverse && scale_inverse) ? (T)(1.0L / (long double)n) : (T)1;
        if (log2n == 0) {
            if (in != out || scale != (T)1)
                rt->launch(fftk::scale_copy_kernel<T>, grid_for(nb, 256), 256, (size_t)0, in, out, (long long)nb, scale);
            return;
        }
        if (algo == ALGO_RADIX2_SHFL) {
            long long grid = ((long long)nb + 3) / 4;
            if (grid > (long long)rt->num_cus() * 8) grid = (long long)rt->num_cus() * 8;
            const size_t smem = (size_t)(n / 2 + 4 * n) * SZ;
            const int inv = inverse ? 1 : 0;
            switch (log2n) {
                case 7: rt->launch(fftk::wave_dit_kernel<T, 2>, grid, 256, smem, in, out, (const cpx<T>*)tw_half, (long long)nb, inv, scale); break;
                case 8: rt->launch(fftk::wave_dit_kernel<T, 4>, grid, 256, smem, in, out, (const cpx<T>*)tw_half, (long long)nb, inv, scale); break;
                case 9: rt->launch(fftk::wave_dit_kernel<T, 8>, grid, 256, smem, in, out, (const cpx<T>*)tw_half, (long long)nb, inv, scale); break;
                default: rt->launch(fftk::wave_dit_kernel<T, 16>, grid, 256, smem, in, out, (const cpx<T>*)tw_half, (long long)nb, inv, scale); break;
            }
            return;
        }
        if (algo == ALGO_RADIX2_GLOBAL) {
            const long long total = (long long)nb * n;
            rt->launch(fftk::bitrev_kernel<T>, grid_for(total, 256), 256, (size_t)0, in, out, log2n, total);
            for (int s = 1; s <= log2n; s++)
                rt->launch(fftk::radix2_dit_stage_kernel<T>, grid_for(total / 2, 256), 256, (size_t)0, out,
                           (const cpx<T>*)tw_half, log2n, s, total / 2, inverse ? 1 : 0, s == log2n ? scale : (T)1);
            return;
        }
        if (wide.ok) {  // n = 8192 fp32: one round trip (fft_wide_row.h) instead of the two-pass schedule
            launch_wide(in, out, nb, inverse, scale);
            rt->mark(0);
            return;
        }
        if (passes.size() == 1) {
            launch_pass(0, in, out, nb, inverse, scale);
            rt->mark(0);
            return;
        }
        // Team kernel first; the multi-pass plan below is then queued as its fallback: its launches read the team
        // kernel's status word and return at once unless the teams could not be formed (nothing touched yet).
        run_if = nullptr;
        int mark0 = 0;
        if (team.ok && nb >= team.min_batch && team_suspend > 0) team_suspend--;
        else if (team.ok && nb >= team.min_batch) {
            // In place, and this plan's team kernel has never been seen to end well: run it from a copy of the input, so that a timeout
            // (a member that never arrives: status 2) can be repaired like an out-of-place execute -- fft_gpu_execute has no failure
            // mode (reference include/fft_gpu.h:102).  One device copy per such execute until the first sync proves the kernel; where
            // the copy cannot be allocated the execute runs unstaged (and a timeout then reports it lost).
            const cpx<T>* tin = in;
            bool staged = false;
            if ((const void*)in == (const void*)out && !team_proven && team_replay) {
                const size_t bytes = ((size_t)SZ << log2n) * (size_t)nb;
                if (team_stage_used + bytes > team_stage_bytes && team_stage_used == 0) {
                    if (team_stage) rt->dfree(team_stage);
                    team_stage = (cpx<T>*)rt->dmalloc(bytes);
                    team_stage_bytes = team_stage ? bytes : 0;
                }
                if (team_stage && team_stage_used + bytes <= team_stage_bytes) {
                    cpx<T>* dst = (cpx<T>*)((char*)team_stage + team_stage_used);
                    rt->d2d_async(dst, in, bytes);
                    team_stage_used += bytes;
                    tin = dst;
                    staged = true;
                }
            }
            launch_team(tin, out, nb, inverse, scale);
            team_pending++;
            if (team_log.size() < 4096) team_log.push_back(TeamExec{tin, out, nb, inverse, scale_inverse, staged});
            else team_log_dropped++;
            rt->mark(0);
            run_if = team.ctl + fftk::TEAM_CTL_STATUS;
            mark0 = 1;
        }
        for (int b0 = 0; b0 < nb; b0 += chunk) {
            const int cb = (nb - b0) < chunk ? (nb - b0) : chunk;
            const cpx<T>* src = in + (size_t)b0 * (size_t)n;
            cpx<T>* dst = out + (size_t)b0 * (size_t)n;
            if (passes.size() == 2) {
                launch_pass(0, src, scratch, cb, inverse, (T)1);
                rt->mark(mark0 + 0);
                launch_pass(1, scratch, dst, cb, inverse, scale);
                rt->mark(mark0 + 1);
            } else {
                launch_pass(0, src, scratch, cb, inverse, (T)1);
                rt->mark(mark0 + 0);
                launch_pass(1, scratch, scratch, cb, inverse, (T)1);
                rt->mark(mark0 + 1);
                launch_pass(2, scratch, dst, cb, inverse, scale);
                rt->mark(mark0 + 2);
            }
        }
        run_if = nullptr;
    }
};

// ---------------------------------------------------------------------------
// Bluestein plan (reference algorithms/core/bluestein.c:79-155), any n >= 1.
// The chirp and FFT(b) are computed ONCE at plan time (the reference recomputes
// both per call); the chirp phase k^2 mod 2n is reduced in integers.
// ---------------------------------------------------------------------------
template <typename T, typename RT>
class BluesteinPlan {
  public:
    static constexpr int SZ = (int)sizeof(cpx<T>);
    RT* rt = nullptr;
    int n = 0, log2m = 0, dir = -1, max_batch = 1;
    Pow2Plan<T, RT> core;
    cpx<T>* chirp = nullptr;  // n entries, reference's chirp[k] = exp(i * (-dir) * pi k^2 / n)
    cpx<T>* bfft = nullptr;   // m entries
    cpx<T>* work = nullptr;   // max_batch * m
    bool ok = false;
    bool no_fusion = false;   // tests: run the element-wise steps as kernels of their own
    bool no_chain = false;    // tests: keep the forward transform's last and the inverse's first pass as two kernels

    ~BluesteinPlan() {
        if (!rt) return;
        if (chirp) rt->dfree(chirp);
        if (bfft) rt->dfree(bfft);
        if (work) rt->dfree(work);
    }

    bool build(RT* runtime, int n_, int dir_, int algo, int batch) {
        rt = runtime;
        n = n_;
        dir = dir_;
        max_batch = batch;
        long long m = 1;
        while (m < 2ll * n - 1) m <<= 1;
        log2m = ilog2(m);
        core.prefer_chain = true;  // forward + inverse of length m back to back
        core.wants_hooks = true;
        if (!core.build(rt, log2m, algo, batch)) return false;
        std::vector<cpx<T>> c((size_t)n), b((size_t)m);
        const long double pi = 3.141592653589793238462643383279502884L;
        for (long long k = 0; k < n; k++) {
            const long long p = (k * k) % (2ll * n);
            const long double ang = pi * (long double)p / (long double)n * (long double)(-dir);
            c[(size_t)k].re = (T)cosl(ang);
            c[(size_t)k].im = (T)sinl(ang);
        }
        c.push_back(c[0]);  // padding entry (see the allocation)
        for (long long k = 0; k < m; k++) { b[(size_t)k].re = 0; b[(size_t)k].im = 0; }
        for (long long k = 0; k < n; k++) {
            b[(size_t)k] = c[(size_t)k];
            if (k > 0) b[(size_t)(m - k)] = c[(size_t)k];
        }
        chirp = (cpx<T>*)rt->dmalloc((size_t)(n + 1) * SZ);  // + 1: a 16-byte table read of the last fp32 pair stays inside
        bfft = (cpx<T>*)rt->dmalloc((size_t)m * SZ);
        work = (cpx<T>*)rt->dmalloc((size_t)batch * (size_t)m * SZ);
        if (!chirp || !bfft || !work) return false;
        rt->h2d(chirp, c.data(), (size_t)(n + 1) * SZ);
        rt->h2d(bfft, b.data(), (size_t)m * SZ);
        core.execute(bfft, bfft, 1, false);
        ok = true;
        return true;
    }

    // FFT_m(x conj(chirp), zero padded) * B -> inverse FFT_m -> * conj(chirp), first n.  With hook-capable passes the
    // three element-wise steps ride on the FFT passes (modulate + zero fill in the forward transform's first load,
    // the product with B in its last store, demodulate + truncate + 1/n in the inverse transform's last store): two
    // transforms, no HBM round trip of their own; the zero half of the padded input is never read and the discarded
    // half of the result never written.
    void execute(const cpx<T>* in, cpx<T>* out, int nb) {
        const long long m = 1ll << log2m;
        const T scale = dir > 0 ? (T)(1.0L / (long double)n) : (T)1;
        if (core.hook_capable() && !no_fusion) {
            ExecHooks<T> f;
            f.pre_tab = chirp; f.pre_mode = fftk::HOOK_MUL_CONJ; f.n_in = n; f.in_pitch = n;
            f.post_tab = bfft; f.post_mode = fftk::HOOK_MUL;
            ExecHooks<T> g;
            g.post_tab = chirp; g.post_mode = fftk::HOOK_MUL_CONJ; g.n_out = n; g.out_pitch = n;
            // m fits one tile: modulate -> FFT -> * FFT(b) -> inverse FFT -> demodulate in ONE kernel
            if (!no_chain && core.round_capable()) {
                ExecHooks<T> rr = f;
                rr.mid_tab = bfft; rr.mid_mode = fftk::HOOK_MUL;
                rr.post_tab = chirp; rr.post_tab_b = 0; rr.post_mode = fftk::HOOK_MUL_CONJ; rr.n_out = n; rr.out_pitch = n;
                core.execute_round(in, out, nb, rr, scale);
                return;
            }
            // the forward transform's last pass and the inverse's first as one kernel where their tiles agree (m = 2^21 fp64:
            // 128 x 128 x 128): five HBM round trips of the padded image instead of six
            if (!no_chain && core.chain_capable() && core.execute_chain(in, out, nb, f, g, scale)) return;
            core.execute_hooked(in, work, nb, false, f);
            core.execute_hooked(work, out, nb, true, g, scale);  // the inverse carries the 1/m
            return;
        }
        const unsigned per_block = 256 * BLU_PER_THREAD;
        const unsigned bpr_m = (unsigned)((m + per_block - 1) / per_block);
        const unsigned bpr_n = (unsigned)(((long long)n + per_block - 1) / per_block);
        rt->launch(fftk::pad_mul_kernel<T>, (long long)bpr_m * nb, 256, (size_t)0, in, (long long)n, n, (const cpx<T>*)chirp,
                   (int)fftk::HOOK_MUL_CONJ, work, (int)m, bpr_m);
        core.execute(work, work, nb, false);
        rt->launch(fftk::mul_store_kernel<T>, (long long)bpr_m * nb, 256, (size_t)0, (const cpx<T>*)work, m, (const cpx<T>*)bfft, 0ll,
                   (int)fftk::HOOK_MUL, work, m, (int)m, (T)1, bpr_m);
        core.execute(work, work, nb, true);  // carries the 1/m
        rt->launch(fftk::mul_store_kernel<T>, (long long)bpr_n * nb, 256, (size_t)0, (const cpx<T>*)work, m, (const cpx<T>*)chirp, 0ll,
                   (int)fftk::HOOK_MUL_CONJ, out, (long long)n, n, scale, bpr_n);
    }
};

}  // namespace ffteng
