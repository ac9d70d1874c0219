// fft_kernels.h -- every GPU kernel of the engine (gfx950 / CDNA4).
//
//   tile_fft_kernel         LDS-resident Stockham sub-transforms; the building block of the
//                           one-, two- and three-pass (four-step) power-of-two engine
//   bitrev_kernel           stand-alone bit-reversal permutation (reference radix2_dit.c:70-77)
//   radix2_dit_stage_kernel one in-place radix-2 DIT stage in HBM (reference radix2_dit.c:84-112)
//   pad_mul / mul_store     element-wise ends of a transform when they are not fused into a pass (TileHooks):
//                           Bluestein modulate / pointwise / demodulate (reference bluestein.c:107-141), convolutions
//
// None of this is MFMA work: butterflies are 2x2 / 4x4 with per-element
// twiddles, 6 flop/byte at N = 2^20, i.e. HBM-bound.  What
// matters is that every HBM access is a full 16-byte lane access inside a
// >= 64..128-byte contiguous segment, that the data makes exactly one HBM round
// trip per pass, and that twiddles come from LDS.
#pragma once
#include <type_traits>

#include "fft_codelets.h"

namespace fftk {

#ifndef FFT_WAVES_PER_SIMD
#define FFT_WAVES_PER_SIMD 2
#endif
#ifndef FFT_FORCE_OPAQUE
#define FFT_FORCE_OPAQUE 0
#endif
#ifndef FFT_TILE_NT
#define FFT_TILE_NT 3  // which of TileParams::nt's bits the build honours (non-temporal hint: bit 0 data loads, bit 1 result stores)
#endif
#ifndef FFT_WAVES_PER_SIMD_E4
#define FFT_WAVES_PER_SIMD_E4 4
#endif
#ifndef FFT_WAVES_PER_SIMD_ROWS8
#define FFT_WAVES_PER_SIMD_ROWS8 4  // the fp32 E = 8 rows kernel with its shape baked in fits 122 VGPRs (no spills): two workgroups per CU;
                                    // measured n = 1024 / 2048 / 4096: +5 / +12 / +11 %, n = 512: -6 % (stays at two waves per SIMD)
#endif
enum { LOAD_CCONTIG = 0, LOAD_LCONTIG = 1 };
enum { STORE_CCONTIG = 0, STORE_LCONTIG = 1 };
enum { FAM_SR16 = 0, FAM_R4 = 1, FAM_R2 = 2 };

// One pass = many tiles.  A tile is C "columns" (independent sub-transforms)
// of length L.  Element (l, c) of tile (b, o, ct) lives at
//     in  + b*in_b  + o*in_o  + (ct*C + c)*in_c  + l*in_l
// and result (k, c) goes to
//     out + b*out_b + o*out_o + (ct*C + c)*out_c + k*out_k.
// LOAD_CCONTIG needs in_c == 1 (columns adjacent in memory: the strided
// "column FFT" of the four-step scheme, optimizations/parallel_fft.c:227-237 in
// the reference); LOAD_LCONTIG needs in_l == 1 (each sub-transform contiguous:
// the "row FFT", parallel_fft.c:250-261).  Same for the store side.
//
// The C columns are processed as H groups of CG = C/H columns: all H groups are
// loaded (and later stored) together, so HBM sees C*sizeof(cpx)-byte row
// segments, but only one group at a time lives in LDS.  This halves the LDS
// footprint (two workgroups per CU: one streams HBM while the other computes)
// without narrowing the segments.
//
// Twiddle tables travel as one blob, copied to LDS at kernel start:
//   [ sa | sb | t0 | t1 | t2 ]
//   stage twiddle     W_L^m   = sa[m & (2^sa_bits - 1)] * sb[m >> sa_bits]      (sb unused when sa_bits == log2L)
//   inter-pass twiddle W_Ntw^m = t0[m & ..] * t1[(m >> t0_bits) & ..] * t2[m >> (t0_bits + t1_bits)]
// Fused element-wise work at the two ends of a transform (kernel instantiations with HOOK = true; the first pass of a
// plan carries the load side, the last pass the store side, a single-pass plan both).  This is what Bluestein's
// modulate / zero-pad, pointwise product and demodulate / truncate (reference bluestein.c:107-141), the pointwise
// product of an FFT convolution (applications/convolution.c:52-60) and a window (applications/power_spectrum.c:5-25)
// cost when they ride on an FFT pass instead of being HBM round trips of their own.
//   load : x[idx] = idx < n_in ? in[idx] (* pre_tab[idx] | * conj(pre_tab[idx])) : 0        idx = sample index in the transform
//   store: out[idx] = X[idx] (* post_tab[b * post_tab_b + idx] | * conj(...) | -> |X[idx]|^2), only for idx < n_out
// Different row pitches of the user's arrays (Bluestein: n on the outside, m inside) are the ordinary in_b / out_b
// (in_c / out_c for the single-pass kernel) of TileParams, set by the launcher.
enum { HOOK_NONE = 0, HOOK_MUL = 1, HOOK_MUL_CONJ = 2, HOOK_ABS2 = 3 };
template <typename T>
struct TileHooks {
    const cpx<T>* pre_tab;
    const cpx<T>* post_tab;
    long long post_tab_b;  // elements between the tables of consecutive transforms (0: one table for the whole batch)
    int n_in;              // valid input samples per transform
    int n_out;             // outputs stored per transform
    int pre_mode, post_mode;
    int in_vec_ok, out_vec_ok;  // 16-byte accesses allowed on the user's side (rows 16-byte aligned: even pitch for fp32)
    // HOOK bit 3 (single-pass kernel only): FFT -> spectral product -> inverse FFT in ONE kernel.  After the forward stages the
    // spectrum is multiplied by mid_tab[K] (one table for the whole batch; mid_mode as post_mode, HOOK_ABS2 needs none), the
    // inverse transform runs on the registers (a thread holds the same index set before and after a transform) and the result
    // goes through the store side above (post_tab / n_out), scaled by TileParams::scale.
    const cpx<T>* mid_tab;
    int mid_mode;
};

template <typename T>
struct TileParams {
    const cpx<T>* in;
    cpx<T>* out;
    const cpx<T>* tables;  // device copy of the blob
    int group_bytes;       // LDS bytes reserved per column group (multiple of 16)
    int tables_bytes;      // multiple of 16
    int off_tables;        // byte offset of the blob's LDS copy
    int o_sb, o_t0, o_t1, o_t2;  // element offsets inside the blob (sa at 0)
    int sa_bits, t0_bits, t1_bits, t2_bits;
    int log2L;
    int log2C;  // all H groups together
    int n_ct;   // column tiles per (b, o)
    int n_o;
    long long in_b, in_o, in_c, in_l;
    long long out_b, out_o, out_c, out_k;
    // LOAD_LCONTIG only: the row may be stored as blocks of 2^in_blk_bits elements, block i at i * in_blk_stride
    // (the tile-major scratch image written by the previous pass); in_blk_bits = 30 means one plain contiguous row
    int in_blk_bits;
    long long in_blk_stride;
    long long n_tiles;  // tiles of this launch; a workgroup walks tiles blockIdx, blockIdx + gridDim, ...
    int n_cols;   // columns >= n_cols are padding (read as zero, never stored)
    int inverse;  // 1: inverse transform via the re<->im swap identity
    int order_g;  // > 1: walk tiles transform-fastest within blocks of order_g transforms (see tile_coord)
    int tiles_per_b;
    int pair16;   // tile order: pair half-line neighbours on one XCD (see tile_coord)
    int ablate;   // profiling only (FFT_HIP_ABLATE): 1 skip inter-pass twiddle, 2 skip stages
    long long col_stride;  // fp64 (one column per 16-byte access) only: elements between adjacent columns of a tile on the c-contiguous
                           // sides, 1 everywhere except the staging-free single-pass plan whose "columns" are whole transforms (= n)
    int tw_o;     // 1: the inter-pass twiddle's second index is the tile's `o` index, not its column (two-pass COLUMN transforms of
                  // 2D plans: the columns of the tile are matrix columns, the four-step index n2 is the row offset o)
    int nt;       // non-temporal hint on the pass's HBM streams: bit 0 data loads, bit 1 result stores.  The planner sets a bit
                  // where the side moves whole 128-byte lines (measured: +5 % at n = 64, +13 % on 128^3 fp64; with 64-byte row
                  // segments the hint throws away the half line the neighbouring tile is about to ask for: -6...12 %)
    // not NULL: this launch is the fallback behind a team kernel (fft_team.h) and runs only if that kernel left
    // TEAM_STATUS_NO_TEAMS (1) in the word, i.e. gave up before touching anything
    const unsigned* run_if;
    T scale;      // applied at the store (1/N folded into the last pass)
    TileHooks<T> hk;  // read only by HOOK = true instantiations
};

template <int X>
struct Log2 {
    static constexpr int value = 1 + Log2<X / 2>::value;
};
template <>
struct Log2<1> {
    static constexpr int value = 0;
};

template <typename T>
struct StageTw {
    const cpx<T>* sa;
    const cpx<T>* sb;
    int sa_bits;
    int log2L;
    FFT_DEVICE cpx<T> get(int m) const {
        m &= (1 << log2L) - 1;
        if (sa_bits >= log2L) return sa[m];  // wave-uniform: single-level table (<= 8 KiB)
        return cmul(sa[m & ((1 << sa_bits) - 1)], sb[m >> sa_bits]);
    }
};

// ---------------------------------------------------------------------------
// One Stockham stage of radix R on the E register-resident elements of each of
// the thread's V columns.  Register slot convention: slot e = m + (E/R)*a holds
// input a of the thread's m-th butterfly, and after the stage output k of that
// butterfly.  Butterfly number u = r + TPC*m decomposes as u = kp*Li + q with
// Li = Lprev/R the length of the remaining sub-problems:
//     inputs  idx = kp*Lprev + q + Li*a            (read unless first stage)
//     outputs idx = (kp + P*k)*Li + q,  P = L/Lprev (written unless last stage)
//     twiddle W_Lprev^(q*k) = W_L^(q*P*k)
// After the last stage slot e holds frequency K = r + TPC*e -- the same shape
// the inputs were loaded in, so a c-contiguous store needs no further exchange.
// ---------------------------------------------------------------------------
// General form: the inputs are read from the image at `smem_rd`, the outputs written to the image at `smem`
// (the same image in the tile kernels; the team kernel's first stage reads the LDS-DMA landing buffer and writes
// the work buffer).  `after_read` runs once every wave has finished reading `smem_rd` (not called when first).
// the two workgroup barriers of a stage exchange.  TIMING EXPERIMENT ONLY (-DFFT_EXPERIMENTS -DFFT_ABLATE_STAGE_BARRIERS):
// without the s_barrier the waves of a workgroup drift apart -- results are garbage, but the launch time shows what a
// schedule whose exchanges need no workgroup barrier could gain (the vector-ALU and LDS phases of different waves overlap)
#if defined(FFT_EXPERIMENTS) && defined(FFT_ABLATE_STAGE_BARRIERS) && !defined(FFT_EMU)
#define FFT_STAGE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define FFT_STAGE_SYNC() FFT_SYNC_LDS()
#endif
struct StageNoHook {
    FFT_DEVICE void operator()() const {}
};
// SWZ: bit 0 the inputs are read from, bit 1 the outputs are written to, positions idx ^ ((idx >> 2) & 3) instead of
// idx.  A final radix-4 stage reads elements 4u .. 4u+3 per butterfly: four threads u of a lane group then sit on the
// same banks (4-way conflict, a quarter of all LDS cycles of the team kernel); the XOR spreads them, and the stage
// before writes through the same map (its 16-lane write groups stay conflict-free).
FFT_DEVICE int stage_swizzle(int idx) { return idx ^ ((idx >> 2) & 3); }

template <typename T, int E, int R, int V, int H, int SWZ = 0, class Hook>
FFT_DEVICE void stockham_stage_rw(cpx<T> (&x)[H][E][V], const unsigned char* smem_rd, unsigned char* smem, int group_bytes,
                                  const StageTw<T>& tw, int r, int j, int log2J, int log2TPC, int& log2Lprev, int& log2P,
                                  bool first, bool last, Hook&& after_read, bool swap_in = false);

template <typename T, int E, int R, int V, int H>
FFT_DEVICE void stockham_stage(cpx<T> (&x)[H][E][V], unsigned char* smem, int group_bytes, const StageTw<T>& tw, int r,
                               int j, int log2J, int log2TPC, int& log2Lprev, int& log2P, bool first, bool last) {
    stockham_stage_rw<T, E, R, V, H>(x, smem, smem, group_bytes, tw, r, j, log2J, log2TPC, log2Lprev, log2P, first, last,
                                     StageNoHook());
}

template <typename T, int E, int R, int V, int H, int SWZ, class Hook>
FFT_DEVICE void stockham_stage_rw(cpx<T> (&x)[H][E][V], const unsigned char* smem_rd, unsigned char* smem, int group_bytes,
                                  const StageTw<T>& tw, int r, int j, int log2J, int log2TPC, int& log2Lprev, int& log2P,
                                  bool first, bool last, Hook&& after_read, bool swap_in) {
    // The H column groups of a tile go through every phase TOGETHER (own LDS region each, shared barriers):
    // twice the independent work between two barriers, half the barriers per byte.
    constexpr int G = E / R;
    constexpr int log2R = Log2<R>::value;
    const int log2Li = log2Lprev - log2R;
    const int Li_mask = (1 << log2Li) - 1;

    if (!first) {
        FFT_UNROLL
        for (int h = 0; h < H; h++) {
            const lvec<T, V>* data = reinterpret_cast<const lvec<T, V>*>(smem_rd + h * group_bytes);
            FFT_UNROLL
            for (int m = 0; m < G; m++) {
                const int u = r + (m << log2TPC);
                const int q = u & Li_mask;
                const int kp = u >> log2Li;
                const int base = (kp << log2Lprev) + q;
                FFT_UNROLL
                for (int a = 0; a < R; a++) {
                    const int idx_in = base + (a << log2Li);
                    lvec<T, V> v = data[(((SWZ & 1) ? stage_swizzle(idx_in) : idx_in) << log2J) + j];
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) x[h][m + G * a][vv] = v.c[vv];
                }
            }
        }
        if (swap_in) {  // wave-uniform: one scalar branch, not a select per value
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) x[h][e][vv] = cswap(x[h][e][vv]);
                }
            }
        }
    }

    // Output twiddles W^(tq * k), k = 1 .. R-1, of the thread's G butterflies: NTW table reads.  Left to itself the
    // compiler emits read, wait, multiply per twiddle -- NTW LDS round trips in a row behind the butterflies.  Here the
    // reads go out in batches of TB: the first batch BEFORE the butterflies (it lands under their arithmetic), batch b + 1
    // before the multiplies of batch b.  (Two-level tables: both factors are read, the product is formed at use.)
    // Only where the register budget has room for it (SWZ bit 2: the team kernels, two waves per SIMD).
    // SWZ bit 3 (team kernels): the R - 1 output twiddles of a butterfly are the POWERS of its first one, W^(tq k) = w1^k.
    // One table read (w1) and R - 2 complex products by repeated squaring / products at most log2 R deep, instead of
    // R - 1 table reads with their index arithmetic: the LDS pipe (two waves per SIMD in lock step: its time adds to the
    // vector ALU's) loses 14 of 15 reads per stage, and the reads it loses are the bank-conflicted ones.  Costs
    // rounding: each twiddle carries up to log2 R products (rel. error of a transform 1.3e-7 -> ~4e-7 in fp32).
    constexpr bool TREE_TW = (SWZ & 8) != 0;
    constexpr bool PIPE_TW = (SWZ & 4) != 0 && !TREE_TW;
    constexpr int NTW = PIPE_TW ? G * (R - 1) : 0;
#ifndef FFT_TW_BATCH
#define FFT_TW_BATCH 4
#endif
    constexpr int TB = FFT_TW_BATCH;
    const bool tw_one_level = tw.sa_bits >= tw.log2L;  // wave-uniform
    cpx<T> wa[NTW > 0 ? NTW : 1], wb[NTW > 0 ? NTW : 1];
    auto tw_load_batch = [&](int b) __attribute__((always_inline)) {
        FFT_UNROLL
        for (int f = b * TB; f < (b + 1) * TB; f++) {
            if (f < NTW) {
                const int m = f / (R - 1), k = f % (R - 1) + 1;
                const int u = r + (m << log2TPC);
                const int q = u & Li_mask;
                const int mm = ((q << log2P) * k) & ((1 << tw.log2L) - 1);
                if (tw_one_level) {
                    wa[f] = tw.sa[mm];
                } else {
                    wa[f] = tw.sa[mm & ((1 << tw.sa_bits) - 1)];
                    wb[f] = tw.sb[mm >> tw.sa_bits];
                }
            }
        }
    };
    if (PIPE_TW && !last) {
        tw_load_batch(0);
        FFT_SCHED_BARRIER();
    }

    FFT_UNROLL
    for (int h = 0; h < H; h++) {
        FFT_UNROLL
        for (int m = 0; m < G; m++) {
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++) {
                cpx<T> t[R];
                FFT_UNROLL
                for (int a = 0; a < R; a++) t[a] = x[h][m + G * a][vv];
                dft_inplace<T, R>(t);
                FFT_UNROLL
                for (int a = 0; a < R; a++) x[h][m + G * a][vv] = t[a];
            }
        }
    }

    if (!last && TREE_TW) {
        FFT_UNROLL
        for (int m = 0; m < G; m++) {
            const int u = r + (m << log2TPC);
            const int q = u & Li_mask;
            cpx<T> pw[R];  // pw[k] = w1^k
            pw[1] = tw.get(q << log2P);
            FFT_UNROLL
            for (int k = 2; k < R; k++) {
                const int hb = 1 << (31 - __builtin_clz((unsigned)k));  // highest set bit (k is a compile-time constant once unrolled)
                pw[k] = (k == hb) ? cmul(pw[k >> 1], pw[k >> 1]) : cmul(pw[hb], pw[k - hb]);
            }
            FFT_UNROLL
            for (int k = 1; k < R; k++) {
                FFT_UNROLL
                for (int h = 0; h < H; h++) {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) x[h][m + G * k][vv] = cmul(x[h][m + G * k][vv], pw[k]);
                }
            }
        }
    }
    if (!last) {
        if (!PIPE_TW && !TREE_TW) {
            FFT_UNROLL
            for (int m = 0; m < G; m++) {
                const int u = r + (m << log2TPC);
                const int q = u & Li_mask;
                const int tq = q << log2P;
                FFT_UNROLL
                for (int k = 1; k < R; k++) {
                    const cpx<T> w = tw.get(tq * k);  // one lookup serves all groups and both columns
                    FFT_UNROLL
                    for (int h = 0; h < H; h++) {
                        FFT_UNROLL
                        for (int vv = 0; vv < V; vv++) x[h][m + G * k][vv] = cmul(x[h][m + G * k][vv], w);
                    }
                }
            }
        }
        FFT_UNROLL
        for (int b = 0; b * TB < NTW; b++) {
            if ((b + 1) * TB < NTW) tw_load_batch(b + 1);
            FFT_SCHED_BARRIER();
            FFT_UNROLL
            for (int f = b * TB; f < (b + 1) * TB; f++) {
                if (f < NTW) {
                    const int m = f / (R - 1), k = f % (R - 1) + 1;
                    const cpx<T> w = tw_one_level ? wa[f] : cmul(wa[f], wb[f]);  // one lookup serves all groups and both columns
                    FFT_UNROLL
                    for (int h = 0; h < H; h++) {
                        FFT_UNROLL
                        for (int vv = 0; vv < V; vv++) x[h][m + G * k][vv] = cmul(x[h][m + G * k][vv], w);
                    }
                }
            }
            FFT_SCHED_BARRIER();
        }
        if (!first) {
            FFT_STAGE_SYNC();  // everyone has finished reading the previous exchange
            after_read();
        }
        FFT_UNROLL
        for (int h = 0; h < H; h++) {
            lvec<T, V>* data = reinterpret_cast<lvec<T, V>*>(smem + h * group_bytes);
            FFT_UNROLL
            for (int m = 0; m < G; m++) {
                const int u = r + (m << log2TPC);
                const int q = u & Li_mask;
                const int kp = u >> log2Li;
                FFT_UNROLL
                for (int k = 0; k < R; k++) {
                    lvec<T, V> v;
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) v.c[vv] = x[h][m + G * k][vv];
                    const int idx = ((kp + (k << log2P)) << log2Li) + q;
                    data[(((SWZ & 2) ? stage_swizzle(idx) : idx) << log2J) + j] = v;
                }
            }
        }
        FFT_STAGE_SYNC();
    }
    log2Lprev = log2Li;
    log2P += log2R;
}

// All stages of one tile.  `before_last` runs once, right before the LAST stage reads its inputs from
// LDS (or before the only stage): at that point the thread's data registers are dead (everything sits in
// LDS), which is where the E = 16 kernels issue the next tile's prefetch without raising the register peak.
template <typename T, int E, int FAM, int V, int H, class F>
FFT_DEVICE void stockham_all_stages(cpx<T> (&x)[H][E][V], unsigned char* smem, int group_bytes, const StageTw<T>& tw,
                                    int r, int j, int log2J, int log2TPC, int log2L, F&& before_last) {
    constexpr int RM = (FAM == FAM_SR16) ? E : (FAM == FAM_R4 ? (E < 4 ? E : 4) : 2);
    constexpr int log2RM = Log2<RM>::value;
    int log2Lprev = log2L, log2P = 0;
    const int n_full = log2L / log2RM;
    const int rem = log2L - n_full * log2RM;
    const int total = n_full + (rem ? 1 : 0);
    FFT_UNROLL
    for (int s = 0; s < n_full; s++) {
        if (s == total - 1) before_last();
        stockham_stage<T, E, RM, V, H>(x, smem, group_bytes, tw, r, j, log2J, log2TPC, log2Lprev, log2P, s == 0, s == total - 1);
    }
    if (rem) before_last();
    if (RM > 2 && rem == 1)
        stockham_stage<T, E, 2, V, H>(x, smem, group_bytes, tw, r, j, log2J, log2TPC, log2Lprev, log2P, total == 1, true);
    if (RM > 4 && rem == 2)
        stockham_stage<T, E, (RM > 4 ? 4 : 2), V, H>(x, smem, group_bytes, tw, r, j, log2J, log2TPC, log2Lprev, log2P, total == 1, true);
    if (RM > 8 && rem == 3)
        stockham_stage<T, E, (RM > 8 ? 8 : 2), V, H>(x, smem, group_bytes, tw, r, j, log2J, log2TPC, log2Lprev, log2P, total == 1, true);
}

// ---------------------------------------------------------------------------
// The tile kernel.  Thread (j, r): j = tid mod J selects V adjacent columns of
// each group (J = CG/V lanes cover one 16*J-byte row segment of a group),
// r = tid div J in [0, L/E) selects which E elements of those columns the
// thread owns.  nthreads == J * L/E exactly.
//
// PERSISTENT + PREFETCH: a workgroup walks tiles blockIdx, blockIdx + gridDim,
// ...; the 16-byte loads of the NEXT tile are issued into `nxt` before the
// current tile's stages run, and every barrier inside the stages is LDS-only
// (FFT_SYNC_LDS leaves vmcnt alone), so each CU keeps HBM reads in flight while
// it computes.  A CU can only sustain ~20 GB/s of misses (latency x outstanding
// lines); without this overlap the load, compute and store phases of a tile
// simply add up (measured: copy-only 3.8 ms + stages 1.6 ms + twiddle 0.4 ms).
// ---------------------------------------------------------------------------
template <typename T>
struct TileCoord {
    const cpx<T>* in;
    cpx<T>* out;
    int c0;
    long long b;      // transform index (multi-pass tiles; the single-pass kernel's transforms are its columns)
    long long oidx;   // o * out_o: offset of this tile's outputs inside the transform (hooks)
    long long iidx;   // o * in_o
    int o;            // the tile's o index itself (TileParams::tw_o)
};

template <typename T>
FFT_DEVICE TileCoord<T> tile_coord(const TileParams<T>& p, long long tile) {
    TileCoord<T> tc;
    unsigned t32 = (unsigned)tile;  // launches never exceed 2^31 tiles
    if (p.order_g > 1) {
        // Walk order: within blocks of order_g transforms the TRANSFORM index runs fastest, so the workgroups that
        // are resident together work on the same column tile of order_g different transforms (measured: the column
        // pass gains 10 %).  order_g divides the number of transforms of the launch.
        const unsigned P = (unsigned)p.tiles_per_b, G = (unsigned)p.order_g;
        const unsigned q = t32 / (G * P), i = t32 - q * (G * P);
        t32 = (q * G + i % G) * P + i / G;
    }
    if (p.pair16 && tile < (p.n_tiles & ~15ll)) {
        // Within every 16 consecutive sequence numbers, workgroups g and g + 8 (same XCD under the observed
        // round-robin placement; speed only, never correctness) get column tiles 2k and 2k + 1, i.e. the two
        // halves of the same 128-byte lines, in the same iteration, so that XCD's L2 sees both halves together.
        const unsigned b = t32 & 15u;
        t32 = (t32 & ~15u) | (((b & 7u) << 1) | (b >> 3));
    }
    const unsigned ct = t32 % (unsigned)p.n_ct;
    const unsigned rest = t32 / (unsigned)p.n_ct;
    const unsigned o = rest % (unsigned)p.n_o;
    const long long b = rest / (unsigned)p.n_o;
    tc.c0 = ct << p.log2C;
    tc.b = b;
    tc.oidx = (long long)o * p.out_o;
    tc.iidx = (long long)o * p.in_o;
    tc.o = (int)o;
    long long boff_in = b * p.in_b, boff_out = b * p.out_b;
    if (FFT_ABLATE(p.ablate & 16)) boff_out %= (8 * p.out_b);  // timing experiment only: scratch side wraps into 8 transforms (cache-resident)
    if (FFT_ABLATE(p.ablate & 32)) boff_in %= (8 * p.in_b);
    tc.in = p.in + boff_in + o * p.in_o + (long long)tc.c0 * p.in_c;
    tc.out = p.out + boff_out + o * p.out_o + (long long)tc.c0 * p.out_c;
    return tc;
}

// FIXED != 0 bakes (log2L << 8 | log2C) into the instantiation: every LDS offset becomes an immediate and the
// stage loop unrolls (fewer address VGPRs, less integer VALU); FIXED == 0 reads both from the parameters.
// HOOK: 0 none; bit 0 the load side of TileHooks is compiled in, bit 1 the store side, bit 2 the load-side table values
// are prefetched together with the data (+ E * 4 VGPRs per group: the 2-waves-per-SIMD kernels; without it they are read
// when the data is consumed -- the 4-waves-per-SIMD rows kernel, whose 128-VGPR budget has no room for them)
// waves per SIMD the register budget of an instantiation is sized for.  The fp32 kernels with their shape baked in need only
// 92-122 VGPRs when built for four waves per SIMD (no spills; built for two they take 150-250 because they may), i.e. two
// 512-thread workgroups per CU.  Measured (profiles/r2_ab_rows_fixed.txt): the single-pass rows kernel gains 5...12 % from
// n = 1024 up (n = 512: -6 %); the multi-pass kernels LOSE 9...11 % with two workgroups per CU and keep the two-wave budget.
#ifndef FFT_WAVES_PER_SIMD_FIX32
#define FFT_WAVES_PER_SIMD_FIX32 2
#endif
template <typename T, int E, int LOADM, int STOREM, int FIXED, int HOOK>
constexpr int tile_waves_per_simd() {
    if (E == 4) return FFT_WAVES_PER_SIMD_E4;
    if (sizeof(T) != 4 || FIXED == 0 || (HOOK & 4)) return FFT_WAVES_PER_SIMD;
    if (LOADM == LOAD_LCONTIG && STOREM == STORE_LCONTIG) return (FIXED >> 8) >= 10 ? FFT_WAVES_PER_SIMD_ROWS8 : FFT_WAVES_PER_SIMD;  // rows kernel: n = 512 loses 6 %
    return FFT_WAVES_PER_SIMD_FIX32;
}
template <typename T, int E, int H, int FAM, int LOADM, int STOREM, bool TWIDDLE, int FIXED, int HOOK = 0>
FFT_KERNEL void FFT_LAUNCH_BOUNDS2((E == 4 ? 1024 : 512), (tile_waves_per_simd<T, E, LOADM, STOREM, FIXED, HOOK>())) tile_fft_kernel(TileParams<T> p) {
    constexpr int WAVES = tile_waves_per_simd<T, E, LOADM, STOREM, FIXED, HOOK>();
    constexpr int V = vec16<T>::V;
    constexpr int log2V = Log2<V>::value;
    constexpr int log2E = Log2<E>::value;
    constexpr int log2H = Log2<H>::value;
    constexpr int SZ = (int)sizeof(cpx<T>);
    FFT_DYN_SMEM(smem);
    if (p.run_if && *p.run_if != 1u) return;

    const int tid_invariant = FFT_TID;
    const int tid = tid_invariant;
    const int nthreads = FFT_NTHREADS;
    const bool nt_load = (p.nt & FFT_TILE_NT & 1) != 0, nt_store = (p.nt & FFT_TILE_NT & 2) != 0;
    const int log2L = FIXED ? (FIXED >> 8) : p.log2L;
    const int log2C = FIXED ? (FIXED & 255) : p.log2C;
    const int L = 1 << log2L;
    const int log2TPC = log2L - log2E;
    const int log2CG = log2C - log2H;
    const int CG = 1 << log2CG;
    const int log2J = log2CG - log2V;
    const int J = 1 << log2J;
    const int j_invariant = tid & (J - 1);
    const int r_invariant = tid >> log2J;
    const long long n_tiles = p.n_tiles;
    const long long tile_step = FFT_NBLOCKS;

    // ---- tables -> LDS ("twiddles staged in LDS"), once per workgroup
    {
        const vec16<T>* src = reinterpret_cast<const vec16<T>*>(p.tables);
        vec16<T>* dst = reinterpret_cast<vec16<T>*>(smem + p.off_tables);
        for (int i = tid; i < (p.tables_bytes >> 4); i += nthreads) dst[i] = src[i];
    }
    const cpx<T>* tab = reinterpret_cast<const cpx<T>*>(smem + p.off_tables);
    StageTw<T> tw;
    tw.sa = tab;
    tw.sb = tab + p.o_sb;
    tw.sa_bits = p.sa_bits;
    tw.log2L = log2L;

    const int pitch = L * SZ + 16;      // raw-row pitch of the l-contiguous staging image
    const int group_bytes = p.group_bytes;  // LDS bytes of one column group (exchange area or staging image)
    const int log2CPR = log2L - log2V;  // 16-byte chunks per row (L >= V always)
    const int cpr_mask = (1 << log2CPR) - 1;

    // nxt[h][i]: the 16-byte lane loads of a tile, in flight or landed.
    //   LOAD_CCONTIG: nxt[h][e] = element l = r + TPC*e of columns h*CG + V*j + (0..V-1)
    //   LOAD_LCONTIG: nxt[h][i] = chunk g = tid + i*nthreads of the group's contiguous rows
    //                 (CG rows * L/V chunks == nthreads * E, so every thread moves exactly E chunks)
    // DEPTH tiles are kept in flight per workgroup (double-buffered prefetch registers when E <= 8).
#ifndef FFT_DEPTH
#define FFT_DEPTH 1
#endif
    constexpr int DEPTH = FFT_DEPTH;  // 2 = double-buffered prefetch (+32 VGPRs; needs FFT_FORCE_OPAQUE to stay spill-free)
    vec16<T> nxtbuf[DEPTH][H][E];
    constexpr bool HK_LOAD = (HOOK & 1) != 0, HK_STORE = (HOOK & 2) != 0, HK_TABPF = (HOOK & 4) != 0, HK_ROUND = (HOOK & 8) != 0;
    vec16<T> nxttab[HK_TABPF ? DEPTH : 1][HK_TABPF ? H : 1][HK_TABPF ? E : 1];  // the load-side table values of the same chunks
    const bool pre_on = HK_LOAD && p.hk.pre_mode != HOOK_NONE;  // wave-uniform
    // index (inside its transform) of the first sample of lane-load i of group h, and its tile column / row
    auto load_idx0 = [&](const TileCoord<T>& tc, int h, int i, int r, int j, int tid) __attribute__((always_inline)) -> long long {
        if (LOADM == LOAD_CCONTIG) return tc.iidx + (long long)(r + ((long long)i << log2TPC)) * p.in_l + tc.c0 + h * CG + V * j;
        return (long long)(((tid + i * nthreads) & cpr_mask) * V);
    };
    // (generic on the non-temporal bit: ONE wave-uniform branch per tile, the loads of a tile stay one clause)
    auto prefetch_as = [&](auto nt_tag, long long tile, vec16<T> (&nxt)[H][E], vec16<T> (&ntab)[HK_TABPF ? H : 1][HK_TABPF ? E : 1]) __attribute__((always_inline)) {
        constexpr int NTL = decltype(nt_tag)::value;
        const TileCoord<T> tc = tile_coord(p, tile);
        int r = r_invariant, j = j_invariant, tid = tid_invariant;
        if (E * H >= 16 || (E != 4 && WAVES >= 4) || FFT_FORCE_OPAQUE) {
            FFT_OPAQUE(r);
            FFT_OPAQUE(j);
            FFT_OPAQUE(tid);
        }
        FFT_UNROLL
        for (int h = 0; h < H; h++) {
            FFT_UNROLL
            for (int i = 0; i < E; i++) {
                bool live;
                const cpx<T>* src;
                if (LOADM == LOAD_CCONTIG) {
                    const long long l = r + ((long long)i << log2TPC);
                    live = (tc.c0 + h * CG + V * j) < p.n_cols;
                    src = tc.in + l * p.in_l + (V == 1 ? (long long)(h * CG + j) * p.col_stride : (long long)(h * CG + V * j));
                } else {
                    const int g = tid + i * nthreads;
                    const int t = h * CG + (g >> log2CPR);
                    live = (tc.c0 + t) < p.n_cols;
                    const int l0 = (g & cpr_mask) * V;
                    src = tc.in + (long long)t * p.in_c + (long long)(l0 >> p.in_blk_bits) * p.in_blk_stride +
                          (l0 & ((1 << p.in_blk_bits) - 1));
                }
                if (HK_LOAD) {
                    // zero padding: samples at or beyond n_in are not read; an fp32 pair that straddles the end, or rows
                    // that are not 16-byte aligned (odd pitch), are read value by value
                    const long long idx0 = load_idx0(tc, h, i, r, j, tid);
                    const int n_in = p.hk.n_in;
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) nxt[h][i].c[vv] = mk<T>((T)0, (T)0);
                    if (live && idx0 < n_in) {
                        if (p.hk.in_vec_ok && idx0 + V <= n_in) {
                            nxt[h][i] = fft_ld16<NTL>(reinterpret_cast<const vec16<T>*>(src));
                        } else {
                            FFT_UNROLL
                            for (int vv = 0; vv < V; vv++)
                                if (idx0 + vv < n_in) nxt[h][i].c[vv] = src[vv];
                        }
                        if (HK_TABPF && pre_on) ntab[HK_TABPF ? h : 0][HK_TABPF ? i : 0] = *reinterpret_cast<const vec16<T>*>(p.hk.pre_tab + idx0);  // table padded to a multiple of V
                    }
                } else if (live) {
                    nxt[h][i] = fft_ld16<NTL>(reinterpret_cast<const vec16<T>*>(src));
                } else {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) nxt[h][i].c[vv] = mk<T>((T)0, (T)0);
                }
            }
        }
    };
    auto prefetch = [&](long long tile, vec16<T> (&nxt)[H][E], vec16<T> (&ntab)[HK_TABPF ? H : 1][HK_TABPF ? E : 1]) __attribute__((always_inline)) {
        if (nt_load) prefetch_as(std::integral_constant<int, 1>{}, tile, nxt, ntab);
        else prefetch_as(std::integral_constant<int, 0>{}, tile, nxt, ntab);
    };
    // load-side product, applied to the landed chunks (zero samples stay zero whatever the stale table register holds)
    auto apply_pre = [&](vec16<T> (&nxt)[H][E], vec16<T> (&ntab)[HK_TABPF ? H : 1][HK_TABPF ? E : 1], const TileCoord<T>& tc, int r, int j, int tid) __attribute__((always_inline)) {
        if (!pre_on) return;
        FFT_UNROLL
        for (int h = 0; h < H; h++) {
            FFT_UNROLL
            for (int i = 0; i < E; i++) {
                const long long idx0 = load_idx0(tc, h, i, r, j, tid);
                vec16<T> tv;
                if (HK_TABPF) tv = ntab[HK_TABPF ? h : 0][HK_TABPF ? i : 0];
                else if (idx0 < p.hk.n_in) tv = *reinterpret_cast<const vec16<T>*>(p.hk.pre_tab + idx0);
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) {
                    if (idx0 + vv < p.hk.n_in) {
                        const cpx<T> w = tv.c[vv];
                        nxt[h][i].c[vv] = p.hk.pre_mode == HOOK_MUL_CONJ ? cmul_conj(nxt[h][i].c[vv], w) : cmul(nxt[h][i].c[vv], w);
                    }
                }
            }
        }
    };

    // x[h][e][v] *= W_Ntw^(row * column): row = r + TPC*e is the index along the transformed axis (frequency K in
    // a column pass, sample l in a row pass), column = the tile column.  Two- or three-level LDS tables.
    // tw_col >= 0: every column of the tile takes W^(K * tw_col) (TileParams::tw_o); -1: W^(K * column)
    auto interpass_twiddle = [&](cpx<T> (&x)[H][E][V], int c0, int r, int j, int tw_col) __attribute__((always_inline)) {
        const cpx<T>* t0 = tab + p.o_t0;
        const cpx<T>* t1 = tab + p.o_t1;
        const cpx<T>* t2 = tab + p.o_t2;
        const unsigned m0 = (1u << p.t0_bits) - 1u, m1 = (1u << p.t1_bits) - 1u;
        const int sh2 = p.t0_bits + p.t1_bits;
        FFT_UNROLL
        for (int h = 0; h < H; h++) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const unsigned K = (unsigned)(r + (e << log2TPC));
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) {
                    const unsigned m = K * (tw_col >= 0 ? (unsigned)tw_col : (unsigned)(c0 + h * CG + V * j + vv));
                    cpx<T> w = cmul(t0[m & m0], t1[(m >> p.t0_bits) & m1]);
                    if (p.t2_bits) w = cmul(w, t2[m >> sh2]);
                    x[h][e][vv] = cmul(x[h][e][vv], w);
                }
            }
        }
    };

    long long tile0 = FFT_BID;
    FFT_UNROLL
    for (int d = 0; d < DEPTH; d++)
        if (tile0 + d * tile_step < n_tiles) prefetch(tile0 + d * tile_step, nxtbuf[d], nxttab[HK_TABPF ? d : 0]);
    FFT_SYNC();  // tables visible (also drains the first prefetches; steady state uses LDS-only barriers)

    // one tile: consume `nxt` (landed), refill it with the tile DEPTH steps ahead, compute, store
    auto do_tile = [&](long long tile, vec16<T> (&nxt)[H][E], vec16<T> (&ntab)[HK_TABPF ? H : 1][HK_TABPF ? E : 1]) __attribute__((always_inline)) {
        const long long tile_ahead = tile + DEPTH * tile_step;
        const TileCoord<T> tc = tile_coord(p, tile);
        cpx<T> x[H][E][V];
        // E = 16 only: hide the lane coordinates from the optimizer once per tile, otherwise LICM hoists every
        // per-stage LDS address / twiddle index out of the persistent loop and keeps ~100 of them live (spills at
        // 64 + 64 data VGPRs).  At E = 8 the hoisting fits the budget and SAVES the per-tile recomputation (+7 %).
        int r = r_invariant, j = j_invariant, tid = tid_invariant;
        if (E * H >= 16 || (E != 4 && WAVES >= 4) || FFT_FORCE_OPAQUE) {
            FFT_OPAQUE(r);
            FFT_OPAQUE(j);
            FFT_OPAQUE(tid);
        }

        // ---- consume the landed loads: slot e of group h <- element l = r + TPC*e
        if (HK_LOAD) apply_pre(nxt, ntab, tc, r, j, tid);
        if (LOADM == LOAD_CCONTIG) {
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) x[h][e][vv] = nxt[h][e].c[vv];
                }
            }
        } else {
            FFT_SYNC_LDS();  // the previous user of the LDS regions (last tile's store image / last exchange) is done
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int i = 0; i < E; i++) {
                    const int g = tid + i * nthreads;
                    *reinterpret_cast<vec16<T>*>(smem + h * group_bytes + (size_t)(g >> log2CPR) * pitch +
                                                 (size_t)(g & cpr_mask) * 16) = nxt[h][i];
                }
            }
            FFT_SYNC_LDS();
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    const int l = r + (e << log2TPC);
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++)
                        x[h][e][vv] = *reinterpret_cast<const cpx<T>*>(smem + h * group_bytes + (size_t)(V * j + vv) * pitch +
                                                                       (size_t)l * SZ);
                }
            }
        }
        // Next tile's loads.  E <= 8: a thread's tile share is 32 VGPRs, so the prefetch is issued right here and
        // flies during ALL of this tile's stages.  E == 16 (64 + 64 VGPRs would spill under the 2-waves-per-SIMD
        // budget): issued from inside the last group's stages, at the point where the data registers are dead.
        constexpr bool EARLY = (E <= 8) || (FAM != FAM_SR16);  // radix-2/4 codelets leave room for 64 + 64 data VGPRs
        const bool have_next = tile_ahead < n_tiles;
        if (EARLY && have_next) prefetch(tile_ahead, nxt, ntab);

        if (p.inverse) {  // inverse = forward transform between two re<->im swaps
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) x[h][e][vv] = cswap(x[h][e][vv]);
                }
            }
        }

        // row pass of a multi-pass plan: the inter-pass twiddle W_N^(k1 * n2) is applied HERE, to the loaded samples
        // (n2 = r + TPC*e runs along the row, k1 = the tile column) -- this pass hides arithmetic behind its memory
        // traffic, the column pass before it does not
        if (TWIDDLE && LOADM == LOAD_LCONTIG && !FFT_ABLATE(p.ablate & 1)) interpass_twiddle(x, tc.c0, r, j, p.tw_o ? tc.o : -1);

        if (!FFT_ABLATE(p.ablate & 2)) {
            FFT_SYNC_LDS();  // staging image / previous tile's last exchange fully consumed
            stockham_all_stages<T, E, FAM, V, H>(x, smem, group_bytes, tw, r, j, log2J, log2TPC, log2L, [&]() {
                if (!EARLY && have_next) prefetch(tile_ahead, nxt, ntab);
            });
        } else if (!EARLY && have_next) {
            prefetch(tile_ahead, nxt, ntab);
        }

        // ---- the prefetched tile has landed long ago: take the vmcnt wait now, before this tile's stores
#ifdef FFT_EARLY_WAIT
        if (have_next) {
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int i = 0; i < E; i++) {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) {
                        FFT_WAIT_LOADED(nxt[h][i].c[vv].re);
                        FFT_WAIT_LOADED(nxt[h][i].c[vv].im);
                    }
                }
            }
        }
#endif

        // ---- inter-pass twiddle (column pass: applied to the results, before the store), scale, inverse swap
        if (TWIDDLE && LOADM == LOAD_CCONTIG && !FFT_ABLATE(p.ablate & 1)) interpass_twiddle(x, tc.c0, r, j, p.tw_o ? tc.o : -1);
        if (HK_ROUND) {
            // FFT -> product -> inverse FFT without leaving the CU (TileHooks::mid_tab): slot e holds X[K], K = r + TPC * e, and
            // that is also the slot layout the stages start from, so the inverse (forward between two re<->im swaps) runs on
            // the registers as they are
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    const int K = r + (e << log2TPC);
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) {
                        cpx<T> v = x[h][e][vv];
                        if (p.hk.mid_mode == HOOK_ABS2) v = mk<T>(v.re * v.re + v.im * v.im, (T)0);
                        else if (p.hk.mid_mode == HOOK_MUL_CONJ) v = cmul_conj(v, p.hk.mid_tab[K]);
                        else if (p.hk.mid_mode == HOOK_MUL) v = cmul(v, p.hk.mid_tab[K]);
                        x[h][e][vv] = cswap(v);
                    }
                }
            }
            FFT_SYNC_LDS();  // the forward transform's last exchange is fully consumed
            stockham_all_stages<T, E, FAM, V, H>(x, smem, group_bytes, tw, r, j, log2J, log2TPC, log2L, []() {});
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) x[h][e][vv] = cswap(x[h][e][vv]);
                }
            }
        }
        if (p.inverse) {
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) x[h][e][vv] = cswap(x[h][e][vv]);
                }
            }
        }
        if (p.scale != (T)1) {
            FFT_UNROLL
            for (int h = 0; h < H; h++) {
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) x[h][e][vv] = cscale(x[h][e][vv], p.scale);
                }
            }
        }

        // ---- store: slot e holds frequency K = r + TPC*e
        const bool post_on = HK_STORE && p.hk.post_mode != HOOK_NONE;
        // store-side product of one value: idx = its index inside the transform, tb = the transform's table
        auto post_op = [&](cpx<T> v, const cpx<T>* tb, long long idx) __attribute__((always_inline)) -> cpx<T> {
            if (p.hk.post_mode == HOOK_ABS2) return mk<T>(v.re * v.re + v.im * v.im, (T)0);
            const cpx<T> w = tb[idx];
            return p.hk.post_mode == HOOK_MUL_CONJ ? cmul_conj(v, w) : cmul(v, w);
        };
        auto store_as = [&](auto nt_tag) __attribute__((always_inline)) {
            constexpr int NTS = decltype(nt_tag)::value;
            if (STOREM == STORE_CCONTIG) {
                const cpx<T>* tb = HK_STORE ? p.hk.post_tab + tc.b * p.hk.post_tab_b : nullptr;
                FFT_UNROLL
                for (int e = 0; e < E; e++) {
                    const long long K = r + ((long long)e << log2TPC);
                    FFT_UNROLL
                    for (int h = 0; h < H; h++) {
                        if ((tc.c0 + h * CG + V * j) < p.n_cols) {
                            vec16<T> v;
                            FFT_UNROLL
                            for (int vv = 0; vv < V; vv++) v.c[vv] = x[h][e][vv];
                            cpx<T>* dst = tc.out + K * p.out_k + (V == 1 ? (long long)(h * CG + j) * p.col_stride : (long long)(h * CG + V * j));
                            if (HK_STORE) {
                                const long long idx0 = tc.oidx + K * p.out_k + tc.c0 + h * CG + V * j;
                                if (idx0 < p.hk.n_out) {
                                    if (post_on) {
                                        FFT_UNROLL
                                        for (int vv = 0; vv < V; vv++)
                                            if (idx0 + vv < p.hk.n_out) v.c[vv] = post_op(v.c[vv], tb, idx0 + vv);
                                    }
                                    if (p.hk.out_vec_ok && idx0 + V <= p.hk.n_out) {
                                        fft_st16<NTS>(reinterpret_cast<vec16<T>*>(dst), v);
                                    } else {
                                        FFT_UNROLL
                                        for (int vv = 0; vv < V; vv++)
                                            if (idx0 + vv < p.hk.n_out) dst[vv] = v.c[vv];
                                    }
                                }
                            } else {
                                fft_st16<NTS>(reinterpret_cast<vec16<T>*>(dst), v);
                            }
                        }
                    }
                }
            } else {
                FFT_SYNC_LDS();  // last exchange fully consumed
                FFT_UNROLL
                for (int h = 0; h < H; h++) {
                    FFT_UNROLL
                    for (int e = 0; e < E; e++) {
                        const int K = r + (e << log2TPC);
                        FFT_UNROLL
                        for (int vv = 0; vv < V; vv++)
                            *reinterpret_cast<cpx<T>*>(smem + h * group_bytes + (size_t)(V * j + vv) * pitch + (size_t)K * SZ) = x[h][e][vv];
                    }
                }
                FFT_SYNC_LDS();
                FFT_UNROLL
                for (int h = 0; h < H; h++) {
                    FFT_UNROLL
                    for (int i = 0; i < E; i++) {
                        const int g = tid + i * nthreads;
                        const int t = h * CG + (g >> log2CPR);
                        const int pos = g & cpr_mask;
                        if (tc.c0 + t < p.n_cols) {
                            vec16<T> v = *reinterpret_cast<const vec16<T>*>(smem + h * group_bytes + (size_t)(g >> log2CPR) * pitch + (size_t)pos * 16);
                            cpx<T>* dst = tc.out + (long long)t * p.out_c + (long long)pos * V;
                            if (HK_STORE) {  // the single-pass kernel's transforms are its columns: transform index = c0 + t
                                const long long idx0 = (long long)pos * V;
                                const cpx<T>* tb = p.hk.post_tab + (long long)(tc.c0 + t) * p.hk.post_tab_b;
                                if (idx0 < p.hk.n_out) {
                                    if (post_on) {
                                        FFT_UNROLL
                                        for (int vv = 0; vv < V; vv++)
                                            if (idx0 + vv < p.hk.n_out) v.c[vv] = post_op(v.c[vv], tb, idx0 + vv);
                                    }
                                    if (p.hk.out_vec_ok && idx0 + V <= p.hk.n_out) {
                                        fft_st16<NTS>(reinterpret_cast<vec16<T>*>(dst), v);
                                    } else {
                                        FFT_UNROLL
                                        for (int vv = 0; vv < V; vv++)
                                            if (idx0 + vv < p.hk.n_out) dst[vv] = v.c[vv];
                                    }
                                }
                            } else {
                                fft_st16<NTS>(reinterpret_cast<vec16<T>*>(dst), v);
                            }
                        }
                    }
                }
            }
        };
        if (nt_store) store_as(std::integral_constant<int, 1>{});
        else store_as(std::integral_constant<int, 0>{});
    };

    for (long long tile = tile0; tile < n_tiles; tile += DEPTH * tile_step) {
        FFT_UNROLL
        for (int d = 0; d < DEPTH; d++)
            if (tile + d * tile_step < n_tiles) do_tile(tile + d * tile_step, nxtbuf[d], nxttab[HK_TABPF ? d : 0]);
    }
}

// ---------------------------------------------------------------------------
// Stand-alone bit-reversal permutation: out[b][rev(i)] = in[b][i].
// in == out swaps pairs once (i < rev(i)), like the reference's loop.
// ---------------------------------------------------------------------------
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) bitrev_kernel(const cpx<T>* in, cpx<T>* out, int log2n, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    const unsigned mask = (1u << log2n) - 1u;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) {
        const unsigned k = (unsigned)(i & mask);
        const long long base = i - k;
        const unsigned jrev = bitrev32(k, log2n);
        if (in == out) {
            if (k < jrev) {
                cpx<T> a = out[base + k], c = out[base + jrev];
                out[base + k] = c;
                out[base + jrev] = a;
            }
        } else {
            out[base + jrev] = in[i];
        }
    }
}

// ---------------------------------------------------------------------------
// One in-place radix-2 DIT stage over the whole batch (after bit reversal):
// m = 2^stage, t = k + j, u = t + m/2, w = W_m^j   (reference radix2_dit.c:84-112,
// with table twiddles instead of the w *= w_m recurrence).
// ---------------------------------------------------------------------------
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256)
    radix2_dit_stage_kernel(cpx<T>* x, const cpx<T>* tw_half /* W_n^k, k < n/2 */, int log2n, int stage,
                            long long total_butterflies, int inverse, T scale) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    const int log2h = log2n - 1;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total_butterflies; i += stride) {
        const long long b = i >> log2h;
        const unsigned t = (unsigned)(i & ((1ll << log2h) - 1));
        const unsigned half = 1u << (stage - 1);
        const unsigned jj = t & (half - 1);
        const unsigned lo = ((t >> (stage - 1)) << stage) + jj;
        const unsigned hi = lo + half;
        cpx<T> w = tw_half[(size_t)jj << (log2n - stage)];
        if (inverse) w.im = -w.im;
        cpx<T>* xb = x + (b << log2n);
        const cpx<T> pr = cmul(xb[hi], w);
        const cpx<T> a = xb[lo];
        xb[hi] = cscale(csub(a, pr), scale);
        xb[lo] = cscale(cadd(a, pr), scale);
    }
}

// ---------------------------------------------------------------------------
// wave_dit_kernel -- the reference's radix-2 DIT pipeline (algorithms/core/radix2_dit.c:59-120: bit-reversal
// permutation, then log2 n butterfly stages of stride 1, 2, 4, ...) held by ONE 64-lane wavefront per transform.
//   1. coalesced HBM load, bit-reversal permutation through LDS (write at rev(i), read back contiguous):
//      lane l then owns elements i = l*E .. l*E + E-1 of the permuted array (E = n/64);
//   2. stages with butterfly stride < E run inside the lane's registers;
//   3. stages with stride E .. 32E pair lane l with lane l ^ (stride/E): the partner's value arrives by
//      __shfl_xor -- no LDS traffic and no barrier ("wavefront-level shuffle for the small-stride stages");
//   4. every lane stores its E contiguous results.
// Twiddles W_n^k (k < n/2) are staged in LDS once per workgroup.  n in {128, 256, 512, 1024}.
// ---------------------------------------------------------------------------
template <typename T, int E>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256)
    wave_dit_kernel(const cpx<T>* in, cpx<T>* out, const cpx<T>* tw_half, long long batch, int inverse, T scale) {
    constexpr int N = 64 * E;
    constexpr int LOG2N = 6 + Log2<E>::value;
    constexpr int WAVES = 4;
    FFT_DYN_SMEM(smem);
    cpx<T>* tws = reinterpret_cast<cpx<T>*>(smem);                  // N/2 twiddles
    cpx<T>* perm = reinterpret_cast<cpx<T>*>(smem) + N / 2;         // WAVES x N permutation image
    const int tid = FFT_TID;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    for (int i = tid; i < N / 2; i += FFT_NTHREADS) tws[i] = tw_half[i];

    for (long long t0 = FFT_BID * WAVES; t0 < batch; t0 += FFT_NBLOCKS * WAVES) {
        const long long t = t0 + wave;
        const bool live = t < batch;
        cpx<T>* pw = perm + wave * N;
        FFT_SYNC();  // tables visible / previous transform's image fully read
        if (live) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const int i = lane + 64 * e;
                cpx<T> v = in[t * N + i];
                if (inverse) v = cswap(v);
                pw[bitrev32((unsigned)i, LOG2N)] = v;
            }
        }
        FFT_SYNC();
        cpx<T> y[E];
        FFT_UNROLL
        for (int e = 0; e < E; e++) y[e] = live ? pw[lane * E + e] : mk<T>((T)0, (T)0);

        // stages whose butterfly stride is inside the lane
        FFT_UNROLL
        for (int s = 1; s <= Log2<E>::value; s++) {
            const int half = 1 << (s - 1);
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                if ((e & half) == 0) {
                    const int j = e & (half - 1);
                    const cpx<T> w = tws[j << (LOG2N - s)];
                    const cpx<T> tt = cmul(y[e + half], w);
                    y[e + half] = csub(y[e], tt);
                    y[e] = cadd(y[e], tt);
                }
            }
        }
        // stages whose partner lives in another lane of the same wavefront
        FFT_UNROLL
        for (int s = Log2<E>::value + 1; s <= LOG2N; s++) {
            const int half = 1 << (s - 1);
            const int lane_mask = half / E;
            const bool is_bot = (lane & lane_mask) != 0;
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const int i = lane * E + e;
                const int j = i & (half - 1);
                const cpx<T> w = tws[j << (LOG2N - s)];
                cpx<T> other;
                other.re = FFT_SHFL_XOR(y[e].re, lane_mask);
                other.im = FFT_SHFL_XOR(y[e].im, lane_mask);
                const cpx<T> bot = is_bot ? y[e] : other;
                const cpx<T> top = is_bot ? other : y[e];
                const cpx<T> tt = cmul(bot, w);
                y[e] = is_bot ? csub(top, tt) : cadd(top, tt);
            }
        }
        if (live) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                cpx<T> v = y[e];
                if (inverse) v = cswap(v);
                out[t * N + lane * E + e] = cscale(v, scale);
            }
        }
    }
}

template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) scale_copy_kernel(const cpx<T>* in, cpx<T>* out, long long total, T scale) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) out[i] = cscale(in[i], scale);
}

// ---------------------------------------------------------------------------
// Element-wise ends of a transform as kernels of their own -- the UNFUSED form of fftk::TileHooks, used when a plan's
// passes have no HOOK instantiation (explicit butterfly family, tiny lengths, the team kernel).  Bluestein (reference
// algorithms/core/bluestein.c:107-141; chirp[k] = exp(i * (-dir) * pi k^2 / n) as stored by the reference):
//   modulate + zero fill  a[b][k] = x[b][k] * conj(chirp[k]), k < n; 0, n <= k < m      pad_mul_kernel,   HOOK_MUL_CONJ
//   pointwise             a[b][k] *= B[k], B = FFT_m(b)                                  mul_store_kernel, HOOK_MUL
//   demodulate            y[b][k] = a[b][k] * conj(chirp[k]) * scale, k < n              mul_store_kernel, HOOK_MUL_CONJ
// One workgroup handles 256 * BLU_PER_THREAD consecutive k of ONE transform: block -> (row, k-block) costs a single
// scalar 32-bit division per workgroup, no per-element 64-bit division.
// ---------------------------------------------------------------------------
#define BLU_PER_THREAD 4

// out[b][k] = k < n_in ? in[b * in_pitch + k] (op) tab[k] : 0,   k < m   (out rows have pitch m)
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256)
    pad_mul_kernel(const cpx<T>* in, long long in_pitch, int n_in, const cpx<T>* tab, int mode, cpx<T>* out, int m, unsigned blocks_per_row) {
    const unsigned bid = (unsigned)FFT_BID;
    const unsigned row = bid / blocks_per_row;
    const int k0 = (int)(bid - row * blocks_per_row) * (256 * BLU_PER_THREAD) + FFT_TID;
    const cpx<T>* xr = in + (long long)row * in_pitch;
    cpx<T>* ar = out + (long long)row * m;
    FFT_UNROLL
    for (int u = 0; u < BLU_PER_THREAD; u++) {
        const int k = k0 + u * 256;
        if (k < m) {
            cpx<T> v = mk<T>((T)0, (T)0);
            if (k < n_in) {
                v = xr[k];
                if (mode == HOOK_MUL) v = cmul(v, tab[k]);
                else if (mode == HOOK_MUL_CONJ) v = cmul_conj(v, tab[k]);
            }
            ar[k] = v;
        }
    }
}

// out[b * out_pitch + k] = (in[b * in_pitch + k] (op) tab[b * tab_b + k]) * scale,   k < n_out;  in == out allowed
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256)
    mul_store_kernel(const cpx<T>* in, long long in_pitch, const cpx<T>* tab, long long tab_b, int mode, cpx<T>* out, long long out_pitch,
                     int n_out, T scale, unsigned blocks_per_row) {
    const unsigned bid = (unsigned)FFT_BID;
    const unsigned row = bid / blocks_per_row;
    const int k0 = (int)(bid - row * blocks_per_row) * (256 * BLU_PER_THREAD) + FFT_TID;
    const cpx<T>* ar = in + (long long)row * in_pitch;
    const cpx<T>* tb = tab + (long long)row * tab_b;
    cpx<T>* yr = out + (long long)row * out_pitch;
    FFT_UNROLL
    for (int u = 0; u < BLU_PER_THREAD; u++) {
        const int k = k0 + u * 256;
        if (k < n_out) {
            cpx<T> v = ar[k];
            if (mode == HOOK_MUL) v = cmul(v, tb[k]);
            else if (mode == HOOK_MUL_CONJ) v = cmul_conj(v, tb[k]);
            else if (mode == HOOK_ABS2) v = mk<T>(v.re * v.re + v.im * v.im, (T)0);
            yr[k] = cscale(v, scale);
        }
    }
}

}  // namespace fftk
