// fft_team_quad.h -- team_quad_kernel: a whole transform per team of CUs of one XCD, ONE HBM round trip, every HBM
// access a whole 128-byte line (256-byte row segments at n = 2^20), the exchange between the two steps in L2.
//
// team_fft_kernel / team_defer_kernel (fft_team.h, fft_team_defer.h) cut a transform n = L x L into 64 KiB tiles of
// L rows x 8 columns: 64-byte row segments, which a CU's memory pipeline streams at two thirds of the rate of whole
// lines (profiles/r1e_membench4), three Stockham stages with two LDS exchanges per tile, and five team-wide arrivals
// per transform with the column step's results handed over tile by tile.  Here a seat (workgroup, one per CU) owns
// NC = L / TS ADJACENT columns in the column step and NC adjacent rows in the row step (n = 2^20, TS = 32: 32 x 8 bytes =
// 256-byte segments), i.e. 256 KiB of the transform -- more than LDS holds, so both steps are decimated in time by 4
// (the four-step split of optimizations/parallel_fft.c:213-272 with each length-L transform itself split 4 x L/4):
//
//   column step, chunk a = 0..3: rows j1 = 4 b + a (b < M = L/4) of my NC columns land in LDS (LDS-DMA, 64 KiB, the next
//       chunk flies meanwhile); length-M transforms as two radix-E stages (M = E^2, ONE LDS exchange, run in place);
//       the results x W_L^(a kb) wait in registers (4 x E values per thread);
//   combine: radix-4 butterflies over a, thread-local, then the inter-step twiddle W_n^(k1 j2): the thread now holds
//       X[k1][j2] for 4 E rows k1 of ONE column j2;
//   exchange + row step, round r = 0..3: every seat writes a quarter of its values into the team's window in the
//       XCD's L2 (2 slots x 2 MiB, rewritten every other round: dirty lines are overwritten in L2, not evicted), one
//       team-wide arrival, every seat pulls its 64 KiB image of the round -- the samples j2 = 4 b' + a' of ONE residue
//       class a' of its NC rows -- into LDS (sc1 LDS-DMA), runs the length-M transforms and keeps the results;
//   final: radix-4 over the classes, thread-local; all 4 E results of a thread go out as whole 256-byte row segments.
//
// Which class a seat receives in round r rotates with the seat (a' = r - s' / (TS/4) mod 4) and which quarter of its rows
// a sending wave serves rotates with the wave's own class: both rotations are absorbed by twiddle exponents (a circular
// shift of a DFT's inputs / outputs is a modulation of its outputs / inputs), so every register index is a compile-time
// constant and no code is specialised per wave.  Four arrivals per transform (fft_team.h has five), all in the
// exchange; HBM traffic per transform: n in, n out.
#pragma once

#include "fft_team_quad_decl.h"

// build switches (same-box A/B with tools/ab_quad.sh; the defaults are what measured fastest, profiles/r3_ab_quad.txt:
// table twiddles -3.5 % (LDS latency at two waves per SIMD), shared-power combine +-0, early send +0.5 %, pipelined rounds
// with the arrival at B1 +7 %, at B2 -4.6 %)
#ifndef QUAD_TW_TABLE   // 1: twiddles whose index is (nearly) wave-uniform are read from the LDS table (broadcast reads); 0: powers
#define QUAD_TW_TABLE 0
#endif
#ifndef QUAD_COMBINE    // 1: inter-step twiddle as (x * sp^k) * base_r with shared powers (independent products); 0: a power tree per block
#define QUAD_COMBINE 0
#endif
#ifndef QUAD_ARR5       // 1: a fifth arrival per transform ("my image of round 3 has landed") lets round 1's values go out before the first team wait
#define QUAD_ARR5 1
#endif
#ifndef QUAD_MERGE      // 1: team polls whose condition is long true ride on the next workgroup barrier (one barrier fewer per round)
#define QUAD_MERGE 0
#endif
#ifndef QUAD_FINAL      // 1: the result stores are issued pair of rows by pair of rows between the final radix-4 butterflies, not behind them
#define QUAD_FINAL 1
#endif
#ifndef QUAD_SPREAD     // 1 / 2: the last quarter / half of a transform's result stores goes out under the next transform's first / first two column chunks
#define QUAD_SPREAD 0
#endif

#ifndef QUAD_LDS_SINGLE  // 1: the stage exchanges use single ds_read_b64 / ds_write_b64 (FFT_LDS_LD64 / ST64), never the fused forms
#define QUAD_LDS_SINGLE 1
#endif
#ifndef QUAD_TW_FIRST    // 1: a stage's twiddle powers are formed in front of its butterflies (under the LDS reads' latency)
#define QUAD_TW_FIRST 1
#endif
#if QUAD_LDS_SINGLE
#define QUAD_LD(p) FFT_LDS_LD64(p)
#define QUAD_ST(p, v) FFT_LDS_ST64((p), (v))
#else
#define QUAD_LD(p) (*(p))
#define QUAD_ST(p, v) (*(p) = (v))
#endif
#ifndef QUAD_PAIR_DPP    // 1: the row-pair exchange in front of every 16-byte store as v_cndmask_b32_dpp (select and lane swap in one instruction)
#define QUAD_PAIR_DPP 1
#endif
#ifndef QUAD_CK_FOLD     // 1: the final modulation i^(sigma k) rides on the last radix-4's add / subtract pattern, the scale on the inter-step twiddle
#define QUAD_CK_FOLD 0
#endif
#ifndef QUAD_SLOTS       // window slots per team: 2, or 3 (round r in slot r mod 3: every round's values are in L2 a round earlier; the
#define QUAD_SLOTS 2     // window is 6 MiB per XCD instead of 4).  The planner allocates 3.
#endif
#ifndef QUAD_LOAD_POL    // experiments: cache-policy bits of the column DMA: 2 nt (default), 4 sc0 sc1, 5 sc0 sc1 nt, 6 sc0 nt
#define QUAD_LOAD_POL 2
#endif
#ifndef QUAD_STORE_POL   // experiments: cache-policy bits of the result stores: 1 nt (default), 2 sc0 sc1 nt, 3 sc1 nt, 4 sc0 sc1, 5 sc0 nt
#define QUAD_STORE_POL 1
#endif
#ifndef QUAD_FINE_TRACE  // profiling builds only: time stamps inside the chunks of transform 3 (tools/quad_fine.py)
#define QUAD_FINE_TRACE 0
#endif

namespace fftk {

template <int E, int LOG2L, int LOG2TS>
struct QuadShape {
    static constexpr int L = 1 << LOG2L, TS = 1 << LOG2TS, M = L / 4;
    static constexpr int LOG2NC = LOG2L - LOG2TS, NC = 1 << LOG2NC;
    static constexpr int NTHR = NC * E;
    static constexpr unsigned IMG = (unsigned)NTHR * E * 8u;  // bytes of one chunk image (fp32)
    static_assert(M == E * E, "the length-L/4 transforms are two radix-E stages");
    static_assert(NC >= 16 && TS >= 4, "four classes of at least four columns; a quarter of the seats per row block");
};

// rotation of the column step's exchange image: the value of (row R, column c) sits at position (c + quad_phi(R / E)) mod NC
// of its row.  Stage 1 writes a row with lanes along c (a rotated row is still one contiguous run of banks), stage 2 reads
// with lanes along g = R / E (E rows, a bank row apart) and four columns: the rotation spreads the E rows over all banks
// (NC = 32, E = 16: lanes (g, il) -> position 4 il + (g & 3) + 8 (g >> 2) + const: 32 distinct 8-byte slots per half wave).
template <int NC>
FFT_DEVICE int quad_phi(int g) { return NC >= 32 ? (g & 3) + 8 * (g >> 2) : ((g >> 1) & 3) + 8 * (g >> 3); }
// Where image row R is kept: with NC = 16 an image row is 128 bytes = HALF the banks, and the E rows a stage-2 thread group
// reads (R = E g + r, g = 0..E-1) all start on the same half.  Swapping the rows of a pair where bit log2 E of R is set puts
// the rows of even and odd g on different halves (NC = 16: lanes (g, il) -> half (r ^ g) & 1, position 4 il + ((g >> 1) & 3)
// + 8 (g >> 3) + const: conflict-free; the natural-map stage 2 of the row step likewise).  An involution; the rows a
// stage-1 thread writes (r + E k) are the slots its own wave has just read.
template <int NC, int E>
FFT_DEVICE int quad_slot(int R) { return NC >= 32 ? R : R ^ ((R >> Log2<E>::value) & 1); }

// w[e] = base * step^e, e < E: E - 1 products at most log2 E deep
template <typename T, int E>
FFT_DEVICE void quad_powers(cpx<T> (&w)[E], cpx<T> base, cpx<T> sp) {
    w[0] = base;
    FFT_UNROLL
    for (int bit = 1; bit < E; bit <<= 1) {
        FFT_UNROLL
        for (int e = 0; e < bit; e++) w[e | bit] = cmul(w[e], sp);
        sp = cmul(sp, sp);
    }
}

// Stage 1 of a length-M = E^2 transform of column `col` of the image (rows of W = 2^LOG2W values): thread r takes rows
// r + E e, radix-E butterfly, twiddle W_M^(r k) = W_L^(4 r k) by powers of one table value, results written IN PLACE
// (rows r + E k: the rows it has just read; ROT: at the rotated position, read by lanes of the same wave only).
// pair_rows (fft_team.h) for lanes l, l ^ 1 whose parity IS the row parity: even lane <- (own s0, partner's s0), odd lane <-
// (partner's s1, own s1).  Device: four v_cndmask_b32_dpp (D = vcc ? src1 : quad_perm[1,0,3,2](src0)) instead of six selects and
// two DPP moves; hand-written, so the two wait states a DPP read needs behind a vector write of its source are spelled out.
template <typename T>
FFT_DEVICE void quad_pair(cpx<T> s0, cpx<T> s1, bool odd, vec16<T>& out) {
#if QUAD_PAIR_DPP && !defined(FFT_EMU)
    float a, b, c, d;
    asm("s_nop 1\n\t"
        "s_mov_b64 vcc, %8\n\t"
        "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %9\n\t"
        "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(s0.re), "v"(s0.im), "v"(s1.re), "v"(s1.im), "s"(0x5555555555555555ull), "s"(0xAAAAAAAAAAAAAAAAull)
        : "vcc");
    (void)odd;
    out.c[0].re = a;
    out.c[0].im = b;
    out.c[1].re = c;
    out.c[1].im = d;
#else
    pair_rows<T>(s0, s1, odd, 1, out);
#endif
}

struct QuadNoMark {
    FFT_DEVICE void operator()(int) const {}
};
template <typename T, int E, int LOG2W, int LOG2L, bool ROT, class Mark = QuadNoMark>
FFT_DEVICE void quad_stage1(cpx<T>* img, const cpx<T>* wl, int col, int r, bool swap_in, Mark&& mark = QuadNoMark()) {
    constexpr int W = 1 << LOG2W, L = 1 << LOG2L;
    cpx<T> x[E];
    FFT_UNROLL
    for (int e = 0; e < E; e++) x[e] = QUAD_LD(&img[(quad_slot<W, E>(r + E * e) << LOG2W) + col]);
#if QUAD_TW_FIRST
    cpx<T> pw[E];
#if QUAD_TW_TABLE
    // r takes two values per wave (lanes run along the image row): broadcast reads, no bank conflicts
    static_assert(4 * (E - 1) * (E - 1) < L, "no wrap of the table index");
    FFT_UNROLL
    for (int k = 1; k < E; k++) pw[k] = wl[4 * r * k];
#else
    pw[1] = wl[(4 * r) & (L - 1)];
    FFT_UNROLL
    for (int k = 2; k < E; k++) {
        const int hb = 1 << (31 - __builtin_clz((unsigned)k));
        pw[k] = (k == hb) ? cmul(pw[k >> 1], pw[k >> 1]) : cmul(pw[hb], pw[k - hb]);
    }
#endif
    if (swap_in) {
        FFT_UNROLL
        for (int e = 0; e < E; e++) x[e] = cswap(x[e]);
    }
    mark(1);  // (waits for the reads)
    dft_inplace<T, E>(x);
    mark(0);
#else
    if (swap_in) {
        FFT_UNROLL
        for (int e = 0; e < E; e++) x[e] = cswap(x[e]);
    }
    mark(1);  // (waits for the reads)
    dft_inplace<T, E>(x);
    mark(0);
    cpx<T> pw[E];
#if QUAD_TW_TABLE
    // r takes two values per wave (lanes run along the image row): broadcast reads, no bank conflicts
    static_assert(4 * (E - 1) * (E - 1) < L, "no wrap of the table index");
    FFT_UNROLL
    for (int k = 1; k < E; k++) pw[k] = wl[4 * r * k];
#else
    pw[1] = wl[(4 * r) & (L - 1)];
    FFT_UNROLL
    for (int k = 2; k < E; k++) {
        const int hb = 1 << (31 - __builtin_clz((unsigned)k));
        pw[k] = (k == hb) ? cmul(pw[k >> 1], pw[k >> 1]) : cmul(pw[hb], pw[k - hb]);
    }
#endif
#endif
    FFT_UNROLL
    for (int k = 1; k < E; k++) x[k] = cmul(x[k], pw[k]);
    mark(0);
    if (ROT || W < 32) FFT_WAVE_LOCKSTEP();  // the rotated positions / swapped rows were read by other lanes of this wave
    FFT_UNROLL
    for (int k = 0; k < E; k++) QUAD_ST(&img[(quad_slot<W, E>(r + E * k) << LOG2W) + (ROT ? ((col + quad_phi<W>(k)) & (W - 1)) : col)], x[k]);
}

// Stage 2: thread g of column `col` takes the E values of rows r + E g (written by stage 1's threads r), radix-E butterfly:
// v[k] = Y[g + E k].  `pos` = the column's position in those rows (rotated or not).
template <typename T, int E, int LOG2W>
FFT_DEVICE void quad_stage2(cpx<T> (&v)[E], const cpx<T>* img, int pos, int g) {
    FFT_UNROLL
    for (int r = 0; r < E; r++) v[r] = QUAD_LD(&img[(quad_slot<(1 << LOG2W), E>(r + E * g) << LOG2W) + pos]);
    dft_inplace<T, E>(v);
}

template <typename T, int E, int LOG2L, int LOG2TS>
FFT_KERNEL void FFT_QUAD_BOUNDS(E, LOG2L, LOG2TS) team_quad_kernel(TeamParams<T> p) {
    static_assert(vec16<T>::V == 2, "fp32: a 16-byte access holds the values of two adjacent rows");
    using S = QuadShape<E, LOG2L, LOG2TS>;
    constexpr int L = S::L, TS = S::TS, M = S::M, NC = S::NC, LOG2NC = S::LOG2NC, NTHR = S::NTHR;
    constexpr int log2E = Log2<E>::value;
    constexpr int NCH = E / 2;   // 16-byte pieces of a chunk image per thread
    constexpr int PPR = NC / 2;  // 16-byte pieces per image row
    constexpr int SZ = 8;
    constexpr unsigned IMG = S::IMG;
    constexpr size_t SLOT = (size_t)TS * IMG;  // one window slot: every seat's image of one round
    constexpr long long n = (long long)L * L;
    FFT_DYN_SMEM(smem);

    const int tid0 = FFT_TID;
    const int tid = tid0;
    unsigned char* const img_b[2] = {smem, smem + IMG};
    unsigned char* const tab_bytes = smem + 2 * (size_t)IMG;
    const unsigned img_lds0 = FFT_LDS_ADDR(smem);
    {
        const vec16<T>* src = reinterpret_cast<const vec16<T>*>(p.tables);
        vec16<T>* dst = reinterpret_cast<vec16<T>*>(tab_bytes);
        for (int i = tid; i < (p.tables_bytes >> 4); i += NTHR) dst[i] = src[i];
    }
    const cpx<T>* const t0 = reinterpret_cast<const cpx<T>*>(tab_bytes);  // W_n^x, x < L / 2 (12 KiB of tables: two workgroups per CU fit)
    const cpx<T>* const wl = t0 + L / 2;                                   // W_L^y, y < L (also W_n^(L y))
    unsigned* const sh = reinterpret_cast<unsigned*>(tab_bytes + p.tables_bytes);  // [slot, xcc, ok, timed out]

    // ---- team formation (team_form, fft_team.h)
    if (tid == 0) team_form(p, sh);
    FFT_SYNC();
    FFT_LDS_FRESH();
    if (!sh[2]) return;
    const unsigned seat = (FFT_UNIFORM(sh[0]) + (unsigned)p.seat_rot) & ((1u << p.log2seats) - 1u);
    const int s = (int)(seat & (unsigned)(TS - 1));  // my seat in the team
    const int team = (int)((FFT_UNIFORM(sh[1]) << (p.log2seats - LOG2TS)) + (seat >> LOG2TS));
    const int n_teams = p.n_xcc << (p.log2seats - LOG2TS);
    const int NTR = team < p.nb ? (p.nb - team + n_teams - 1) / n_teams : 0;  // transforms of this team
    if (NTR == 0) return;
    if (FFT_TEST_DROP()) return;  // emulation only: a member that never arrives

    unsigned char* const sbase = p.scratch + (size_t)team * 3 * SLOT;  // (three slots allocated; QUAD_SLOTS of them used)
    unsigned* const flags = p.ctl + TEAM_CTL_FLAGS + 32 * team;

    int n_ev = 0;
    auto ev = [&]() __attribute__((always_inline)) {
        if (p.trace && tid == 0 && n_ev < p.trace_events - 1) {
            p.trace[(long long)FFT_BID * p.trace_events + n_ev] = FFT_CLOCK();
            n_ev++;
        }
    };
    ev();
    if (p.trace && tid == 0 && p.trace_events > 1) p.trace[(long long)FFT_BID * p.trace_events + p.trace_events - 1] = (team << 8) | s;

    // everybody has made arrival number g <=> the team's counter >= TS * g (nobody makes arrival g + 1 before everybody has
    // made g).  Polled by the first wave with scalar loads, the others wait at the workgroup barrier (fft_team.h).
    auto wait_all = [&](int g) __attribute__((always_inline)) {
        FFT_LDS_FRESH();
        if (sh[3]) return;
        if (tid < FFT_TEAM_POLL_LANES) {
            const long long tstart = FFT_CLOCK();
            while ((int)(FFT_L2_COUNT_POLL(flags) - ((unsigned)g << LOG2TS)) < 0) {
                if (FFT_CLOCK() - tstart > p.timeout_ticks) {
                    team_report_timeout(p);
                    sh[3] = 1;
                    break;
                }
                FFT_SLEEP();
            }
        }
        FFT_SYNC_LDS();
    };
    // the same poll WITHOUT the barrier: the first wave spins, the others walk on -- to the workgroup barrier the caller has next
    // anyway (which then also publishes "everybody has arrived" to them).  For waits whose condition is normally long true.
    auto poll_all = [&](int g) __attribute__((always_inline)) {
        FFT_LDS_FRESH();
        if (sh[3]) return;
        if (tid < FFT_TEAM_POLL_LANES) {
            const long long tstart = FFT_CLOCK();
            while ((int)(FFT_L2_COUNT_POLL(flags) - ((unsigned)g << LOG2TS)) < 0) {
                if (FFT_CLOCK() - tstart > p.timeout_ticks) {
                    team_report_timeout(p);
                    sh[3] = 1;
                    break;
                }
                FFT_SLEEP();
            }
        }
    };
    auto arrive = [&]() __attribute__((always_inline)) {  // call behind a workgroup barrier, every wave's stores complete
        if (tid == 0) FFT_L2_COUNT_ADD(flags);
    };

    // ---- thread coordinates.  Every phase derives them afresh from an opaque copy of the thread id (FFT_OPAQUE): left to
    // itself the optimizer hoists every LDS address, DMA source and window pointer of every phase out of the transform
    // loop -- all of them are loop-invariant -- and keeps hundreds of registers live across it.
    // natural map (stage 1 of both steps, stage 2 and stores of the row step): lanes along the image row
    //   ncol = t & (NC - 1), nr = t >> LOG2NC
    // column-step stage 2 / sender map: a wave = E values of g x 4 columns of ONE class ap = j2 mod 4
    //   g = t & (E - 1), il = (t >> log2E) & 3, wq = t >> (log2E + 2): ap = wq & 3, cc = il + 4 (wq >> 2), c2 = ap + 4 cc
    const int ap = FFT_UNIFORM((tid >> (log2E + 2)) & 3);
    const int sigma = s >> (LOG2TS - 2);          // my block of rows: k1 in [M sigma, M sigma + M)
    const cpx<T> whalf = p.tables[L / 2 + L];  // W_n^(L/2), behind the two tables in the blob
    auto wn = [&](unsigned x) __attribute__((always_inline)) {  // W_n^x = W_n^(x mod L/2) [* W_n^(L/2)] * W_L^(x / L)
        const cpx<T> lo = cmul(t0[x & (L / 2 - 1)], wl[(x >> LOG2L) & (L - 1)]);
        const cpx<T> hi = cmul(lo, whalf);
        return (x & (L / 2)) ? hi : lo;
    };

    auto in_of = [&](int it) __attribute__((always_inline)) { return p.in + (long long)(team + (long long)it * n_teams) * n; };
    auto out_of = [&](int it) __attribute__((always_inline)) { return p.out + (long long)(team + (long long)it * n_teams) * n; };

    // LDS-DMA of column chunk a (rows 4 b + a of my NC columns) into image `im`: lane-linear 16-byte pieces, piece
    // sigma = i NTHR + tid is image row sigma / PPR, columns 2 (sigma mod PPR) ..
    auto dma_chunk = [&](const cpx<T>* inb, int a, int im) __attribute__((always_inline)) {
        int tid = tid0;
        FFT_OPAQUE(tid);
        // (NTHR / PPR rows per wave-front of pieces, a multiple of 2 E: the slot swap is the same in every front)
        const cpx<T>* src = inb + ((long long)(4 * quad_slot<NC, E>(tid / PPR) + a) << LOG2L) + NC * s + 2 * (tid % PPR);
        constexpr long long step = (long long)(4 * (NTHR / PPR)) << LOG2L;
        static_assert((NTHR / PPR) % (2 * E) == 0 || NC >= 32, "row swap pattern repeats per wave-front");
        const unsigned lds = img_lds0 + (unsigned)im * IMG;
        if (p.nt_mask & 1) {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) {
#if defined(FFT_EMU)
                FFT_DMA16_NT(src + i * step, img_b[im], lds, (unsigned)(i * NTHR + tid) * 16u);
#else
                fft_dma16<QUAD_LOAD_POL>(src + i * step, lds + (unsigned)(i * NTHR + tid) * 16u);
#endif
            }
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16(src + i * step, img_b[im], lds, (unsigned)(i * NTHR + tid) * 16u);
        }
    };
    // LDS-DMA of my image of window slot `slot` (IMG contiguous bytes, served by the XCD's L2)
    auto dma_window = [&](int slot, int im) __attribute__((always_inline)) {
        int tid = tid0;
        FFT_OPAQUE(tid);
        const unsigned char* src = sbase + (size_t)slot * SLOT + (size_t)s * IMG + (size_t)tid * 16;
        const unsigned lds = img_lds0 + (unsigned)im * IMG;
        if (p.nt_mask & 4) {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16_L2_NT(src + (size_t)i * NTHR * 16, img_b[im], lds, (unsigned)(i * NTHR + tid) * 16u);
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16_L2(src + (size_t)i * NTHR * 16, img_b[im], lds, (unsigned)(i * NTHR + tid) * 16u);
        }
    };

    // final modulation of the row step: result k of the radix-4 over the classes is due scale * i^(sigma k) (the rounds
    // deliver the classes rotated by sigma)
    cpx<T> ck[4];
    FFT_UNROLL
    for (int k = 0; k < 4; k++) {
        const int pw4 = (sigma * k) & 3;
        const T sc = QUAD_CK_FOLD ? (T)1 : p.scale;
        ck[k] = mk<T>(pw4 == 0 ? sc : pw4 == 2 ? -sc : (T)0, pw4 == 1 ? sc : pw4 == 3 ? -sc : (T)0);
    }

    // transposed store of results ka of a row (k2 = g' + E k + M ka): X[k1 + L k2], k1 = NC s + rho; the lanes of rows rho, rho ^ 1
    // pair up so that every store is 16 bytes and every wave instruction writes whole NC-row segments
    auto store_results = [&](cpx<T>* outb, const cpx<T> (&y)[E], int ka) __attribute__((always_inline)) {
        int t = tid0;
        FFT_OPAQUE(t);
        const int ncol = t & (NC - 1), nr = t >> LOG2NC;
        const bool odd = (ncol & 1) != 0;
        cpx<T>* const line0 = outb + NC * s + (ncol & ~1);
        FFT_UNROLL
        for (int i = 0; i < E / 2; i++) {
            vec16<T> v;
            quad_pair<T>(y[2 * i], y[2 * i + 1], odd, v);
            const long long k2 = nr + E * (2 * i + (odd ? 1 : 0)) + M * ka;
            vec16<T>* const dst = reinterpret_cast<vec16<T>*>(line0 + (k2 << LOG2L));
            if (p.nt_mask & 2) FFT_STORE16_NT(dst, v);
            else *dst = v;
        }
    };
    constexpr int NARR = QUAD_ARR5 ? 5 : 4;  // arrivals per transform
    constexpr int NPEND = QUAD_SPREAD;  // result blocks ka = 4 - NPEND .. 3 of a transform wait for the next one's column chunks 0 .. NPEND - 1
    cpx<T> pend[NPEND ? NPEND : 1][E];

    dma_chunk(in_of(0), 0, 0);
    for (int it = 0; it < NTR; it++) {
        const cpx<T>* inb = in_of(it);
        cpx<T>* outb = out_of(it);
        const int G = NARR * it;  // arrivals made before this transform

        // ================= column step: four chunks, length-M transforms, results x W_L^(a kb) kept
        cpx<T> blk[4][E];
        FFT_UNROLL
        for (int a = 0; a < 4; a++) {
            // my pieces of the chunk have landed ... everybody's have; the other image was last read before this barrier.
            // What may still be in flight are the result stores of the previous transform issued BEHIND this chunk's DMA
            // (vmcnt counts in issue order): (4 - NPEND) E / 2 behind chunk 0, E / 2 behind each of chunks 1 .. NPEND
            if (it > 0 && a == 0) FFT_WAIT_VM_LE((4 - NPEND) * E / 2);
            else if (it > 0 && a <= NPEND) FFT_WAIT_VM_LE(E / 2);
            else FFT_WAIT_VM0();
            FFT_SYNC_LDS();
            ev();
            if (a + 1 < 4) dma_chunk(inb, a + 1, (a + 1) & 1);
            if (it > 0 && a < NPEND) store_results(out_of(it - 1), pend[a], 4 - NPEND + a);
            cpx<T>* img = reinterpret_cast<cpx<T>*>(img_b[a & 1]);
            int t = tid0;
            FFT_OPAQUE(t);
#if QUAD_FINE_TRACE
            // stamps of every wave's first lane inside the chunks of transform 3: p.trace[((block * 8 + wave) * 64) + i]
            int n_fine = a * 12;
            auto fine = [&](int drain) __attribute__((always_inline)) {
                if (it != 3 || !p.trace) return;
                FFT_SCHED_BARRIER();
                if (drain) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if ((tid0 & 63) == 0) p.trace[((long long)FFT_BID * 8 + (tid0 >> 6)) * 64 + n_fine] = FFT_CLOCK();
                n_fine++;
                FFT_SCHED_BARRIER();
            };
            fine(0);  // 0: B1 passed
            quad_stage1<T, E, LOG2NC, LOG2L, true>(img, wl, t & (NC - 1), t >> LOG2NC, p.inverse != 0, fine);  // 1 reads landed, 2 dft, 3 twiddle
            fine(1);  // 4: writes done
            FFT_SYNC_LDS();
            fine(0);  // 5: B2 passed
            FFT_OPAQUE(t);
            const int g = t & (E - 1), c2 = ap + 4 * (((t >> log2E) & 3) + 4 * (t >> (log2E + 4)));
            cpx<T> v[E];
            {
                const int pos = (c2 + quad_phi<NC>(g)) & (NC - 1);
                FFT_UNROLL
                for (int rr = 0; rr < E; rr++) v[rr] = QUAD_LD(&img[(quad_slot<NC, E>(rr + E * g) << LOG2NC) + pos]);
            }
            fine(1);  // 6: stage-2 reads landed
            dft_inplace<T, E>(v);
            fine(0);  // 7: dft
#else
            quad_stage1<T, E, LOG2NC, LOG2L, true>(img, wl, t & (NC - 1), t >> LOG2NC, p.inverse != 0);
            FFT_SYNC_LDS();
            FFT_OPAQUE(t);
            const int g = t & (E - 1), c2 = ap + 4 * (((t >> log2E) & 3) + 4 * (t >> (log2E + 4)));
            cpx<T> v[E];
            quad_stage2<T, E, LOG2NC>(v, img, (c2 + quad_phi<NC>(g)) & (NC - 1), g);
#endif
            if (a == 0) {
                FFT_UNROLL
                for (int k = 0; k < E; k++) blk[0][k] = v[k];
            } else {
                // W_L^(a (g + E k) - M a ap): the class shift ap rotates the radix-4's OUTPUTS (block r = rows M (r - ap))
                cpx<T> w[E];
#if QUAD_TW_TABLE
                FFT_UNROLL
                for (int k = 0; k < E; k++) w[k] = wl[(unsigned)(a * (g + E * k) - M * a * ap) & (L - 1)];
#else
                quad_powers<T, E>(w, wl[(unsigned)(a * g - M * a * ap) & (L - 1)], wl[(a * E) & (L - 1)]);
#endif
                FFT_UNROLL
                for (int k = 0; k < E; k++) blk[a][k] = cmul(v[k], w[k]);
            }
#if QUAD_FINE_TRACE
            fine(0);  // 8: chunk twiddle
#endif
        }
        // block r of my registers goes out in round r: rows k1 = g + E k + M q of column j2, to the seats of row block q
        auto send = [&](int r) __attribute__((always_inline)) {
            int t = tid0;
            FFT_OPAQUE(t);
            const int g = t & (E - 1), cc = ((t >> log2E) & 3) + 4 * (t >> (log2E + 4));
            const int q = (r - ap) & 3;
            unsigned char* const wslot = sbase + (size_t)(r % QUAD_SLOTS) * SLOT;
            const bool odd = (g & 1) != 0;
            const int bprime = (NC / 4) * s + cc;  // (j2 - ap) / 4: my column's place in its class
            FFT_UNROLL
            for (int i = 0; i < E / 2; i++) {
                vec16<T> v;
                quad_pair<T>(blk[r][2 * i], blk[r][2 * i + 1], odd, v);  // even lane: rows (g, g + 1) of slot 2 i; odd lane: rows (g - 1, g) of slot 2 i + 1
                const int k1 = (g & ~1) + E * (2 * i + (odd ? 1 : 0)) + M * q;
                const int dst_seat = k1 >> LOG2NC, rho = k1 & (NC - 1);
                *reinterpret_cast<vec16<T>*>(wslot + (size_t)dst_seat * IMG + (size_t)(((quad_slot<NC, E>(bprime) << LOG2NC) + rho) * SZ)) = v;
            }
        };
        // ---- combine: radix-4 over the chunks, then W_n^(k1 j2), k1 = g + E k + M q, q = (r - ap) mod 4 for block r
        {
            int t = tid0;
            FFT_OPAQUE(t);
            const int g = t & (E - 1), c2 = ap + 4 * (((t >> log2E) & 3) + 4 * (t >> (log2E + 4)));
            const unsigned j2 = (unsigned)(NC * s + c2);  // my column of the transform
            const cpx<T> f1 = wn((unsigned)M * j2), f2 = cmul(f1, f1), f3 = cmul(f2, f1);
            const cpx<T> base0 = QUAD_CK_FOLD ? cscale(wn((unsigned)g * j2), p.scale) : wn((unsigned)g * j2);
            const cpx<T> sp = wn((unsigned)E * j2);
            FFT_UNROLL
            for (int k = 0; k < E; k++) {
                cpx<T> u[4];
                FFT_UNROLL
                for (int a = 0; a < 4; a++) u[a] = blk[a][k];
                dft_inplace<T, 4>(u);
                FFT_UNROLL
                for (int a = 0; a < 4; a++) blk[a][k] = u[a];
            }
#if QUAD_COMBINE
            cpx<T> pk[E];  // sp^k, shared by the four blocks
            pk[1] = sp;
            FFT_UNROLL
            for (int k = 2; k < E; k++) {
                const int hb = 1 << (31 - __builtin_clz((unsigned)k));
                pk[k] = (k == hb) ? cmul(pk[k >> 1], pk[k >> 1]) : cmul(pk[hb], pk[k - hb]);
            }
#endif
            FFT_UNROLL
            for (int r = 0; r < 4; r++) {
                const int q = (r - ap) & 3;  // wave-uniform
                const cpx<T> fq = mk<T>(q == 0 ? (T)1 : q == 1 ? f1.re : q == 2 ? f2.re : f3.re, q == 0 ? (T)0 : q == 1 ? f1.im : q == 2 ? f2.im : f3.im);
                const cpx<T> base = cmul(base0, fq);
#if QUAD_COMBINE
                blk[r][0] = cmul(blk[r][0], base);
                FFT_UNROLL
                for (int k = 1; k < E; k++) blk[r][k] = cmul(cmul(blk[r][k], pk[k]), base);
#else
                cpx<T> w[E];
                quad_powers<T, E>(w, base, sp);
                FFT_UNROLL
                for (int k = 0; k < E; k++) blk[r][k] = cmul(blk[r][k], w[k]);
#endif
                if (r == 0) {
                    if (QUAD_SLOTS == 3) wait_all(G);  // slot 0 was last read in the previous transform's round 3
                    send(0);  // drains under block 1's twiddles
                }
                if (r == 1) {
                    // ================= exchange + row step begins: the team learns that my round-0 values are in L2 while I
                    // still twiddle blocks 2 and 3 (the first team wait absorbs the column step's skew: work behind the
                    // arrival is free)
                    if (QUAD_ARR5 && QUAD_SLOTS == 2) {  // everybody's image of the previous transform's round 3 has landed: long true
                        if (QUAD_MERGE) poll_all(G);
                        else wait_all(G);
                    }
                    FFT_WAIT_VM0();
                    FFT_SYNC_LDS();
                    arrive();  // arrival G + 1
                    ev();
                    if (QUAD_ARR5) send(1);  // slot 1 was last read in that round 3 (three slots: in the previous transform's round 1)
                }
            }
        }
        ev();
        cpx<T> zt[4][E];
        wait_all(G + 1);
        ev();
        dma_window(0, 0);
        if (!QUAD_ARR5) send(1);
        static_assert(QUAD_SLOTS == 2 || QUAD_ARR5, "three slots need the fifth arrival");
        FFT_UNROLL
        for (int r = 0; r < 4; r++) {
            FFT_WAIT_VM0();  // the round's image has landed and my round-(r + 1) values are in L2 ...
            FFT_SYNC_LDS();  // ... everybody's
            if (r < 3 || QUAD_ARR5) arrive();  // arrival G + r + 2 (r = 3, QUAD_ARR5: G + 5, "my image of round 3 has landed")
            ev();
            if (r == 3 && it + 1 < NTR) dma_chunk(in_of(it + 1), 0, 0);  // image 0 was last read in round 2
            cpx<T>* img = reinterpret_cast<cpx<T>*>(img_b[r & 1]);
            int t = tid0;
            FFT_OPAQUE(t);
            quad_stage1<T, E, LOG2NC, LOG2L, false>(img, wl, t & (NC - 1), t >> LOG2NC, false);
            // (QUAD_MERGE: everybody's arrival G + r + 2 was made a whole first stage ago)
            if (QUAD_MERGE && r < 3) poll_all(G + r + 2);
            FFT_SYNC_LDS();
            if (QUAD_MERGE && r < 3) {
                // the next round's image is requested NOW (the other image was last read in round r - 1) and flies under this
                // round's second stage; behind it the values of round r + 2, into the slot this round's image came from
                ev();
                dma_window((r + 1) % QUAD_SLOTS, (r + 1) & 1);
                if (r < 2) send(r + 2);
                static_assert(!(QUAD_MERGE && QUAD_SLOTS == 3), "three slots: not with QUAD_MERGE");
            }
            FFT_OPAQUE(t);
            const int nr = t >> LOG2NC;
            cpx<T> v[E];
            {
                const int pos = t & (NC - 1);
                FFT_UNROLL
                for (int rr = 0; rr < E; rr++) v[rr] = QUAD_LD(&img[(quad_slot<NC, E>(rr + E * nr) << LOG2NC) + pos]);
            }
            if (!QUAD_MERGE && r < 3) {
                // the next round's image is requested NOW (the other image was last read in round r - 1) and flies under this
                // round's second stage; behind it the values of round r + 2, into the slot this round's image came from
                // (everybody's image of round r has landed: arrival G + r + 2 says so)
                wait_all(G + r + 2);
                ev();
                dma_window((r + 1) % QUAD_SLOTS, (r + 1) & 1);
                if (QUAD_SLOTS == 3) {
                    // slot 2 was last read in the previous transform's round 2 (everybody's arrival G says more than that);
                    // slot 0 takes round 3 once everybody's image of round 0 has landed: arrival G + 2, just waited for
                    if (r == 0) {
                        send(2);
                        send(3);
                    }
                } else if (r < 2) {
                    send(r + 2);
                }
            }
            dft_inplace<T, E>(v);
            const int apr = (r - sigma) & 3;  // the class this round delivered (workgroup-uniform)
            if (apr != 0) {
                cpx<T> w[E];
#if QUAD_TW_TABLE
                FFT_UNROLL
                for (int k = 0; k < E; k++) w[k] = wl[apr * (nr + E * k)];  // < 3 M: no wrap
#else
                quad_powers<T, E>(w, wl[(apr * nr) & (L - 1)], wl[(apr * E) & (L - 1)]);
#endif
                FFT_UNROLL
                for (int k = 0; k < E; k++) zt[r][k] = cmul(v[k], w[k]);
            } else {
                FFT_UNROLL
                for (int k = 0; k < E; k++) zt[r][k] = v[k];
            }
        }
        // ---- final radix-4 over the rounds, modulation, transposed store
#if QUAD_FINAL
        static_assert(NPEND == 0, "QUAD_FINAL stores every block at once");
        ev();
        {
            int t = tid0;
            FFT_OPAQUE(t);
            const int ncol = t & (NC - 1), nr = t >> LOG2NC;
            const bool odd = (ncol & 1) != 0;
            cpx<T>* const line0 = outb + NC * s + (ncol & ~1);
            FFT_UNROLL
            for (int i = 0; i < E / 2; i++) {
                cpx<T> y[2][4];
                FFT_UNROLL
                for (int h = 0; h < 2; h++) {
#if QUAD_CK_FOLD
                    // y[k] = i^(sigma k) * DFT4(u)[k]: with A = u0 + u2, B = u1 + u3, C = u0 - u2, D = u1 - u3 the four cases are the
                    // same eight additions with operands swapped (sigma is workgroup-uniform: one scalar branch)
                    const cpx<T> u0 = zt[0][2 * i + h], u1 = zt[1][2 * i + h], u2 = zt[2][2 * i + h], u3 = zt[3][2 * i + h];
                    const cpx<T> A = cadd(u0, u2), B = cadd(u1, u3);
                    y[h][0] = cadd(A, B);
                    if (sigma == 0) {
                        const cpx<T> C = csub(u0, u2), D = csub(u1, u3);
                        y[h][2] = csub(A, B);
                        y[h][1] = cadd_mni(C, D);  // C - i D
                        y[h][3] = csub_mni(C, D);  // C + i D
                    } else if (sigma == 1) {
                        const cpx<T> C = csub(u0, u2), D = csub(u1, u3);
                        y[h][2] = csub(B, A);
                        y[h][1] = csub_mni(D, C);  // i (C - i D) = D + i C
                        y[h][3] = cadd_mni(D, C);  // -i (C + i D) = D - i C
                    } else if (sigma == 2) {
                        const cpx<T> Cn = csub(u2, u0), D = csub(u1, u3);
                        y[h][2] = csub(A, B);
                        y[h][1] = csub_mni(Cn, D);  // -(C - i D) = -C + i D
                        y[h][3] = cadd_mni(Cn, D);  // -(C + i D) = -C - i D
                    } else {
                        const cpx<T> C = csub(u0, u2), Dn = csub(u3, u1);
                        y[h][2] = csub(B, A);
                        y[h][1] = cadd_mni(Dn, C);  // -i (C - i D) = -D - i C
                        y[h][3] = csub_mni(Dn, C);  // i (C + i D) = -D + i C
                    }
                    if (p.inverse) {
                        FFT_UNROLL
                        for (int r = 0; r < 4; r++) y[h][r] = cswap(y[h][r]);
                    }
#else
                    FFT_UNROLL
                    for (int r = 0; r < 4; r++) y[h][r] = zt[r][2 * i + h];
                    dft_inplace<T, 4>(y[h]);
                    FFT_UNROLL
                    for (int r = 0; r < 4; r++) {
                        y[h][r] = cmul(y[h][r], ck[r]);
                        if (p.inverse) y[h][r] = cswap(y[h][r]);
                    }
#endif
                }
                FFT_UNROLL
                for (int ka = 0; ka < 4; ka++) {
                    vec16<T> v;
                    quad_pair<T>(y[0][ka], y[1][ka], odd, v);
                    const long long k2 = nr + E * (2 * i + (odd ? 1 : 0)) + M * ka;
                    vec16<T>* const dst = reinterpret_cast<vec16<T>*>(line0 + (k2 << LOG2L));
#if defined(FFT_EMU)
                    *dst = v;
#else
                    if (p.nt_mask & 2) fft_store16_pol<QUAD_STORE_POL>(dst, v);
                    else *dst = v;
#endif
                }
            }
        }
        ev();
#else
        FFT_UNROLL
        for (int k = 0; k < E; k++) {
            cpx<T> u[4];
            FFT_UNROLL
            for (int r = 0; r < 4; r++) u[r] = zt[r][k];
            dft_inplace<T, 4>(u);
            FFT_UNROLL
            for (int r = 0; r < 4; r++) {
                u[r] = cmul(u[r], ck[r]);
                zt[r][k] = p.inverse ? cswap(u[r]) : u[r];
            }
        }
        ev();
        FFT_UNROLL
        for (int ka = 0; ka < 4; ka++) {
            if (ka >= 4 - NPEND && it + 1 < NTR) {
                FFT_UNROLL
                for (int k = 0; k < E; k++) pend[ka - (4 - NPEND) < 0 ? 0 : ka - (4 - NPEND)][k] = zt[ka][k];
            } else {
                store_results(outb, zt[ka], ka);
            }
        }
#endif
        ev();
    }
}

}  // namespace fftk
