// fft_team_quad.h -- team_quad_kernel: a whole transform per team of CUs of one XCD, ONE HBM round trip, every HBM
// access a run of whole 128-byte lines (256-byte row segments at n = 2^20, 512 at 2^18, 1 KiB at 2^16), the exchange
// between the two steps in four rounds through the XCD's L2.
//
// team_fft_kernel / team_defer_kernel (fft_team.h, fft_team_defer.h) cut a transform n = L1 x L2 into 64 KiB tiles of
// L rows x 8 columns: 64-byte row segments (n = 2^20), which a CU's memory pipeline streams at two thirds of the rate of
// whole lines (profiles/r1e_membench4), three Stockham stages with two LDS exchanges per tile, and five team-wide
// arrivals per transform with the column step's results handed over tile by tile.  Here a seat (workgroup, one per
// CU) owns NC = L2 / TS ADJACENT columns in the column step and NR = L1 / TS adjacent rows in the row step, i.e. n / TS
// values (256 KiB) -- more than LDS holds, so both steps are decimated in time by 4 (the four-step split of
// optimizations/parallel_fft.c:213-272 with the length-L1 and length-L2 transforms themselves split 4 x MA, 4 x MB):
//
//   column step, chunk a = 0..3: rows j1 = 4 b + a (b < MA) of my NC columns land in LDS (LDS-DMA, 64 KiB, the next chunk
//       flies meanwhile); length-MA transforms as a radix-E and a radix-RA stage (MA = E RA, E = 16 values per thread, ONE
//       LDS exchange, run in place); the results x W_L1^(a kb) wait in registers (4 x E values per thread);
//   combine: radix-4 butterflies over a, thread-local, then the inter-step twiddle W_n^(k1 j2): the thread now holds
//       X[k1][j2] for 4 E rows k1 of ONE column j2;
//   exchange + row step, round r = 0..3: every seat writes a quarter of its values into the team's window in the
//       XCD's L2 (2 slots of TS x 64 KiB, rewritten every other round -- or one, see SLOTS), one team-wide arrival (or,
//       SLOTS = 3, per-seat counters and no team-wide wait at all: the pair protocol, n = 2^20 and 2^19), every
//       seat pulls its 64 KiB image of the round -- the samples j2 = 4 b' + a' of ONE residue class a' for each of its NR rows --
//       into LDS (sc1 LDS-DMA), runs the length-MB transforms (MB = E RB) and keeps the results;
//   final: radix-4 over the classes, thread-local; all 4 E results of a thread go out as whole NR-row segments.
//
// Which class a row receives in round r rotates with the row's block q = k1 / MA (a' = r - q mod 4) and which block of
// rows a sending wave serves rotates with the wave's own class: both rotations are absorbed by twiddle exponents (a
// circular shift of a DFT's inputs / outputs is a modulation of its outputs / inputs), so every register index is a
// compile-time constant and no code is specialised per wave.  Five arrivals per transform (eight with one slot), four of
// them in the exchange and never waited for on the spot; HBM traffic per transform: n in, n out.
//
// Shapes (fft_team_quad_decl.h): n = 2^20 = 1024 x 1024 (RA = RB = 16, teams of 32), 2^19 = 1024 x 512 (16, 8; teams of 16),
// 2^18 = 512 x 512 (8, 8; 8), 2^17 = 512 x 256 (8, 4; 4), 2^16 = 256 x 256 (4, 4; 2), 2^15 = 256 x 128 (4, 2; a "team" of ONE: the
// transform's 256 KiB live in one CU's registers, the exchange is the CU's own 64 KiB of L2); the emulation runs E = 4 with
// 64 x 64, 64 x 32 and 32 x 32.
#pragma once

#include "fft_team_quad_decl.h"

// build switches (same-box A/B with tools/ab_quad.sh; the defaults are what measured fastest, profiles/r3_ab_quad.txt).
// Measured and dropped from the source (same file): twiddles read from LDS tables instead of formed as powers (-3.5 %, LDS
// latency at two waves per SIMD), the arrival of a round at the stage barrier instead of the image barrier (-4.6 %), team
// polls merged into stage barriers (-1.5 %), two-deep column prefetch (-1 %), result stores spread over the next column
// step (-5 ... -15 %), a third window slot (-12 %: 6 MiB of window per XCD thrash the 4 MiB L2), two 256-thread
// workgroups per CU with teams of 64 (-21 %), folding the final modulation into the last radix-4 (-0.7 %), other
// cache-policy bits on the streams (+-0; without `nt` -7 %), round 1's values sent behind the first team wait of the exchange instead
// of behind a wait of their own in the combine (-2.5 %), the next transform's index read by every wave with a scalar load of its own in
// round 3 (-10 %: 256 waves per XCD queue behind the result stores; the index now rides on the team wait's poll, see `pub`).
// Kept although within the noise (profiles/r3_ab_early_prefetch.txt: +0.5 % at 2^20, +1 % at 2^18 / 2^16): the next transform's first two
// chunks are requested as soon as their LDS images are free (QUAD_EARLY_CHUNK0 / 1) -- the result stores, not the request time, hold them up.
// Team t of the eight started t * 2.5 ... 10 us late, so that the XCDs' column steps do not meet in HBM: no difference (profiles/r3_ab_stagger.txt; again on
// the pair protocol, t * 4.9 / 10 us: profiles/r4_ab_stagger.txt).
#ifndef QUAD_LDS_SINGLE  // 1: the stage exchanges use single ds_read_b64 / ds_write_b64 (FFT_LDS_LD64 / ST64), never the fused forms (+5 %)
#define QUAD_LDS_SINGLE 1
#endif
#if QUAD_LDS_SINGLE
#define QUAD_LD(p) FFT_LDS_LD64(p)
#define QUAD_ST(p, v) FFT_LDS_ST64((p), (v))
#else
#define QUAD_LD(p) (*(p))
#define QUAD_ST(p, v) (*(p) = (v))
#endif
namespace fftk {
// one value of the exchange image: 8 bytes (fp32: the single-form access above) or 16 (fp64: one ds_read_b128 / ds_write_b128)
template <typename T>
FFT_DEVICE cpx<T> quad_ld(const cpx<T>* p) {
    if constexpr (sizeof(T) == 4) return QUAD_LD(p);
    else return *p;
}
template <typename T>
FFT_DEVICE void quad_st(cpx<T>* p, cpx<T> v) {
    if constexpr (sizeof(T) == 4) QUAD_ST(p, v);
    else *p = v;
}
}  // namespace fftk
#ifndef QUAD_PAIR_DPP    // 1: the row-pair exchange in front of a 16-byte store as v_cndmask_b32_dpp (select and lane swap in one instruction)
#define QUAD_PAIR_DPP 1
#endif
#ifndef QUAD_EARLY_CHUNK0  // 1: chunk 0 of the team's next transform is requested in round 2 (behind the request for round 3's image), not in round 3
#define QUAD_EARLY_CHUNK0 1
#endif
#ifndef QUAD_EARLY_CHUNK1  // 1: chunk 1 of the team's next transform is requested in round 3 too (behind the round's last image reads), not at chunk 0's barrier
#define QUAD_EARLY_CHUNK1 1
#endif
#ifndef QUAD_DEFER_STORES  // 1: a transform's result stores go out in four parts from the column step of the team's NEXT transform (see `final_part`);
#define QUAD_DEFER_STORES -1  // 2 / 3: each part by a quarter / a half of the waves at a time; 0: all of them when the last round is over; -1: 1 for teams of
#endif                        // 32 and 16 (n = 2^20: +1.3 ... 2 % on two slots, +5 % on the pair protocol; 2^19 on the pair protocol +1.1 %), 0 below (2^18: -3 %, 2^17 and
                              // 2^16 +-0.5 %).  profiles/r4_ab_defer_*.txt, r4_ab_pair_protocol_knobs.txt, r4_ab_defer_small_teams.txt
#ifndef QUAD_PAIR_SIGNAL_AT  // pair protocol: where a wave tells the seats it has written to -- 0: behind stage 1 of the round, 1 / 2 / 3: from inside that stage
#define QUAD_PAIR_SIGNAL_AT 2  // (reads in / behind the butterflies / behind the twiddles).  2: +0.7 % at n = 2^20, +1.2 % at 2^19; 1 and 3: +-0
#endif                        // (profiles/r4_ab_pair_signal_point.txt)
#ifndef QUAD_COL_AHEAD  // 1 (teams of 32 and the fp64 shapes): column chunk a + 2 is requested as soon as chunk a's image is free (behind its stage-2
#define QUAD_COL_AHEAD 1  // reads), not when chunk a + 1 has landed: two chunks in flight per CU during most of the column step (12.3 instead of
#endif                    // 13.5 us; +0.9 % at n = 2^20 in three same-box pairs, profiles/r4_ab_column_ahead.txt); 2: every shape
#ifndef QUAD_ABL  // timing experiments only (tools/ab_quad.sh variants; results invalid): 1 no result stores, 2 no stage barrier in the column
#define QUAD_ABL 0  // step, 4 none in the row step, 8 no column-step arithmetic, 16 no window stores, 32 no row-step arithmetic, 64 no landing barrier in the column step (with 2: its waves run free of each other); pair protocol: 128 no guards, 256 no waits for the senders, 512 a guard is ONE poll
#endif
#ifndef QUAD_FINE_TRACE  // profiling builds only: time stamps inside the column chunks of transform 3 (tools/quad_fine.py)
#define QUAD_FINE_TRACE 0
#endif

namespace fftk {

// n = L1 x L2 (L1 rows j1 / k1, L2 columns j2 / k2), L1 = 4 MA, MA = E RA (the column step's sub-transforms), L2 = 4 MB, MB = E RB (the
// row step's); a seat owns NC = L2 / TS columns and NR = L1 / TS rows; NC RA = NR RB threads.
template <int E, int LOG2RA, int LOG2RB, int LOG2L1, int LOG2L2, int LOG2TS>
struct QuadShape {
    static constexpr int RA = 1 << LOG2RA, RB = 1 << LOG2RB, GA = E / RA, GB = E / RB, MA = E * RA, MB = E * RB;
    static constexpr int L1 = 1 << LOG2L1, L2 = 1 << LOG2L2, TS = 1 << LOG2TS;
    static constexpr int LOG2NC = LOG2L2 - LOG2TS, NC = 1 << LOG2NC, LOG2NR = LOG2L1 - LOG2TS, NR = 1 << LOG2NR;
    static constexpr int NTHR = NC * RA;
    static constexpr unsigned IMG_VALUES = (unsigned)NC * MA;  // values of one chunk image = NR MB
    static constexpr int ILN = (64 / RA < NC / 4) ? 64 / RA : NC / 4;  // columns of a class per sender group (a wave on the device)
    static_assert(L1 == 4 * MA && L2 == 4 * MB && RA <= E && RA >= 2 && RB <= E && RB >= 2, "L = 4 M, M = E R: a radix-E and a radix-R stage");
    static_assert(NC * RA == NR * RB, "both steps use every thread");
    // the block of MA rows a row of the row step lies in must be the same for a whole wave (a group of 64 threads along the rows)
    static_assert(NC >= 8 && TS >= 1 && (NR <= 2 * MA || MA % 64 == 0 || E == 4), "four classes of at least two columns; row blocks of whole waves");
};

// rotation of the column step's exchange image: the value of (row R, column c) sits at position (c + quad_phi(R / E)) mod NC
// of its row.  Stage 1 writes a row with lanes along c (a rotated row is still contiguous runs of banks), stage 2 reads
// with lanes along g = R / E (rows a multiple of 256 bytes apart: the same banks) and 64 / R2 columns of one class: the rotation
// spreads them (per half wave the positions 4 il + (g & 3) + (128 / R2) (g >> 2) are 32 different 8-byte slots).
template <int R2>
FFT_DEVICE int quad_phi(int g) { return (g & 3) + (128 / R2) * (g >> 2); }
// ... WITHIN windows of 64 columns: stage 1 rewrites a row in place, and what makes that safe without a barrier is that the
// lanes that read a window are the lanes of ONE wave (natural map: 64 consecutive columns of one row).  A sender group's columns
// (one class, ILN of them, 4 apart) lie inside one window, so the spread over the banks is the same.
template <int R2, int W>
FFT_DEVICE int quad_rot(int c, int g) {
    constexpr int RW = W < 64 ? W : 64;
    return (c & ~(RW - 1)) | ((c + quad_phi<R2>(g)) & (RW - 1));
}

// w[e] = base * step^e, e < K: K - 1 products at most log2 K deep
template <typename T, int K>
FFT_DEVICE void quad_powers(cpx<T>* w, cpx<T> base, cpx<T> sp) {
    w[0] = base;
    FFT_UNROLL
    for (int bit = 1; bit < K; bit <<= 1) {
        FFT_UNROLL
        for (int e = 0; e < bit; e++) w[e | bit] = cmul(w[e], sp);
        sp = cmul(sp, sp);
    }
}

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

// Stage 1 of a length-M = E R2 transform of column `col` of the image (rows of W = 2^LOG2W values): thread r < R2 takes rows
// r + R2 e (e < E), radix-E butterfly, twiddle W_M^(r k) = W_L^(4 r k) by powers of one table value (formed in front of the
// butterflies: under the latency of the reads), results written IN PLACE (rows r + R2 k: the rows it has just read; ROT: at the
// rotated position, read by lanes of the same wave only).
template <typename T, int E, int R2, int LOG2W, int LOG2L, bool ROT, class Mark = QuadNoMark>
FFT_DEVICE void quad_stage1(cpx<T>* img, const cpx<T>* wl, int col, int r, bool swap_in, Mark&& mark = QuadNoMark()) {
    constexpr int W = 1 << LOG2W, G2 = E / R2;
    static_assert(4 * (R2 - 1) * (E - 1) < (1 << LOG2L), "no wrap of the table index");
    cpx<T> x[E];
    FFT_UNROLL
    for (int e = 0; e < E; e++) x[e] = quad_ld<T>(&img[((r + R2 * e) << LOG2W) + col]);
    cpx<T> pw[E];
    pw[1] = wl[4 * r];
    FFT_UNROLL
    for (int k = 2; k < E; k++) {
        const int hb = 1 << (31 - __builtin_clz((unsigned)k));
        pw[k] = (k == hb) ? cmul(pw[k >> 1], pw[k >> 1]) : cmul(pw[hb], pw[k - hb]);
    }
    if (swap_in) {
        FFT_UNROLL
        for (int e = 0; e < E; e++) x[e] = cswap(x[e]);
    }
    mark(1);  // (waits for the reads)
    dft_inplace<T, E>(x);
    mark(0);
    FFT_UNROLL
    for (int k = 1; k < E; k++) x[k] = cmul(x[k], pw[k]);
    mark(0);
    if (ROT) FFT_WAVE_LOCKSTEP();  // the rotated positions were read by other lanes of this wave
    FFT_UNROLL
    for (int k = 0; k < E; k++) quad_st<T>(&img[((r + R2 * k) << LOG2W) + (ROT ? quad_rot<R2, W>(col, k / G2) : col)], x[k]);
}

// Stage 2, the reads: thread g < R2 of column `col` takes the E values of rows E g + rr, i.e. for each of its G2 = E / R2
// butterflies k' = G2 g + i the R2 values stage 1's threads r wrote to rows r + R2 k' (v[i R2 + r]).  `pos` = the column's
// position in those rows (rotated or not).
template <typename T, int E, int LOG2W>
FFT_DEVICE void quad_stage2_read(cpx<T> (&v)[E], const cpx<T>* img, int pos, int g) {
    FFT_UNROLL
    for (int rr = 0; rr < E; rr++) v[rr] = quad_ld<T>(&img[((rr + E * g) << LOG2W) + pos]);
}
// ... and the butterflies: v[i R2 + k] = Y[kb], kb = G2 g + i + E k  (quad_kb)
template <typename T, int E, int R2>
FFT_DEVICE void quad_stage2_dft(cpx<T> (&v)[E]) {
    FFT_UNROLL
    for (int i = 0; i < E / R2; i++) dft_inplace<T, R2>(&v[i * R2]);
}
template <int E, int R2>
FFT_DEVICE int quad_kb(int g, int j) { return (E / R2) * g + j / R2 + E * (j % R2); }

// x[j] *= W_L^(mul * kb(g, j) + off), j < E: per butterfly i one table value and the powers of W_L^(mul E)
template <typename T, int E, int R2, int LOG2L>
FFT_DEVICE void quad_twiddle_kb(cpx<T> (&x)[E], const cpx<T> (&v)[E], const cpx<T>* wl, int mul, int g, int off) {
    constexpr int G2 = E / R2, L = 1 << LOG2L;
    const cpx<T> sp = wl[(mul * E) & (L - 1)];
    FFT_UNROLL
    for (int i = 0; i < G2; i++) {
        cpx<T> w[R2];
        quad_powers<T, R2>(w, wl[(unsigned)(mul * (G2 * g + i) + off) & (L - 1)], sp);
        FFT_UNROLL
        for (int k = 0; k < R2; k++) x[i * R2 + k] = cmul(v[i * R2 + k], w[k]);
    }
}

// SLOTS: window slots per team.  2 (4 MiB of window per XCD): round r + 2 is written into the slot round r came from, every team wait
// has a whole round of slack -- but 4 MiB of window do not stay in the 4 MiB L2: every window byte is written back once and a fifth
// of the window reads miss (memory-side traffic 1.58 x the algorithmic bytes, profiles/r3_pmc_oneslot.txt).  1 (2 MiB per XCD): the
// slot is rewritten in L2 (traffic 1.08 x), at the price of strict alternation -- write, everybody reads, write -- with eight
// arrivals per transform and two team waits per round that sit on the critical path.  Measured (profiles/r3_ab_quad.txt): teams of 2
// (n = 2^16) +6.5 % with one slot, teams of 8 (2^18) +-1 %, teams of 32 (2^20) -20 %.  3 (round 4): one slot's window and schedule with the
// pair protocol instead of the team's counter (`pair_guard` below): teams of 32 +1.7 %, of 16 +3.7 % over two slots, and the traffic of one.
template <typename T, int E, int LOG2RA, int LOG2RB, int LOG2L1, int LOG2L2, int LOG2TS, int SLOTS>
FFT_KERNEL void FFT_QUAD_BOUNDS(LOG2RA, LOG2L2, LOG2TS) team_quad_kernel(TeamParams<T> p) {
    constexpr bool PAIR = SLOTS == 3;  // one image per seat in the window, per-seat counters instead of the team's (see `pair_guard`)
    constexpr bool QUAD_ONE_SLOT = SLOTS == 1 || PAIR;
    constexpr int DEFER = QUAD_DEFER_STORES >= 0 ? QUAD_DEFER_STORES : ((LOG2TS >= 4 && sizeof(T) == 4) ? 1 : 0);
    static_assert(SLOTS == 1 || SLOTS == 2 || SLOTS == 3, "one or two window slots, or one with the pair protocol");
    constexpr int V = vec16<T>::V;  // values per 16-byte access: 2 (fp32: the values of two adjacent rows travel together), 1 (fp64)
    using S = QuadShape<E, LOG2RA, LOG2RB, LOG2L1, LOG2L2, LOG2TS>;
    constexpr int L1 = S::L1, L2 = S::L2, TS = S::TS, MA = S::MA, MB = S::MB, NC = S::NC, LOG2NC = S::LOG2NC, NR = S::NR, LOG2NR = S::LOG2NR;
    constexpr int NTHR = S::NTHR, RA = S::RA, RB = S::RB, GA = S::GA, ILN = S::ILN;
    constexpr int LOG2MA = LOG2L1 - 2;
    constexpr int LOG2ILN = Log2<ILN>::value;
    constexpr int NCH = E / V;   // 16-byte pieces of a chunk image per thread
    constexpr int PPR = NC / V;  // 16-byte pieces per image row
    constexpr int SZ = (int)sizeof(cpx<T>);
    constexpr unsigned IMG = S::IMG_VALUES * (unsigned)SZ;  // bytes of one chunk image
    constexpr size_t SLOT = (size_t)TS * IMG;  // one window slot: every seat's image of one round
    constexpr long long n = (long long)L1 * L2;
    static_assert(NTHR * NCH * 16 == (int)IMG && NTHR % PPR == 0, "a chunk image is NCH pieces per thread");
    FFT_DYN_SMEM(smem);
    const long long t_entry = p.trace ? FFT_CLOCK() : 0;  // (tools/quad_trace.py teams: what formation costs)

    const int tid0 = FFT_TID;
    const int tid = tid0;
    // (a function, not an array of two pointers: an array indexed by a loop variable that the unroller leaves alone -- it does when the loop
    // body grows -- is promoted to a constant global, whose initializer, an LDS address, the backend cannot express)
    auto img_b = [&](int i) __attribute__((always_inline)) -> unsigned char* { return smem + (size_t)i * IMG; };
    unsigned char* const tab_bytes = smem + 2 * (size_t)IMG;
    const unsigned img_lds0 = FFT_LDS_ADDR(smem);
    {
        const vec16<T>* src = reinterpret_cast<const vec16<T>*>(p.tables);
        vec16<T>* dst = reinterpret_cast<vec16<T>*>(tab_bytes);
        for (int i = tid; i < (p.tables_bytes >> 4); i += NTHR) dst[i] = src[i];
    }
    const cpx<T>* const t0 = reinterpret_cast<const cpx<T>*>(tab_bytes);  // W_n^x, x < L2 / 2
    const cpx<T>* const wlA = t0 + L2 / 2;                                 // W_L1^y, y < L1 (also W_n^(L2 y))
    const cpx<T>* const wlB = wlA + L1;                                    // W_L2^y, y < L2
    unsigned* const sh = reinterpret_cast<unsigned*>(tab_bytes + p.tables_bytes);  // [slot, xcc, ok, timed out, next transform + 1]

    // ---- team formation (team_form, fft_team.h)
    if (tid == 0) {
        sh[4] = 0;
        team_form(p, sh);
    }
    FFT_SYNC();
    FFT_LDS_FRESH();
    if (!sh[2]) return;
    const unsigned seat = (FFT_UNIFORM(sh[0]) + (unsigned)p.seat_rot) & ((1u << p.log2seats) - 1u);
    const int s = (int)(seat & (unsigned)(TS - 1));  // my seat in the team
    const int team = (int)((FFT_UNIFORM(sh[1]) << (p.log2seats - LOG2TS)) + (seat >> LOG2TS));
    const int n_teams = p.n_xcc << (p.log2seats - LOG2TS);
    if (team >= p.nb) return;  // more teams than transforms
    if (FFT_TEST_DROP()) return;  // emulation only: a member that never arrives

    unsigned char* const sbase = p.scratch + (size_t)team * (PAIR ? 1 : SLOTS) * SLOT;
    // profiling only (experiments build, results invalid): the second slot aliases half (bit 16) or all (bit 32) of the first -- the
    // two-slot protocol on a window of 3 / 2 MiB per XCD instead of 4: what a smaller window would be worth
    const size_t slot_stride = FFT_ABLATE(p.ablate & 32) ? 0 : FFT_ABLATE(p.ablate & 16) ? SLOT / 2 : SLOT;
    unsigned* const flags = p.ctl + TEAM_CTL_FLAGS + 32 * team;

    int n_ev = 0;
    auto ev = [&]() __attribute__((always_inline)) {
        if (p.trace && tid == 0 && n_ev < p.trace_events - 2) {  // (the last two slots: kernel-entry clock, team / seat)
            p.trace[(long long)FFT_BID * p.trace_events + n_ev] = FFT_CLOCK();
            n_ev++;
        }
    };
    ev();
    if (p.trace && tid == 0 && p.trace_events > 1) p.trace[(long long)FFT_BID * p.trace_events + p.trace_events - 1] = (team << 8) | s;
    if (p.trace && tid == 0 && p.trace_events > 2) p.trace[(long long)FFT_BID * p.trace_events + p.trace_events - 2] = t_entry;

    // everybody has made arrival number g <=> the team's counter >= TS * g (nobody makes arrival g + 1 before everybody has
    // made g).  Polled by the first wave with scalar loads, the others wait at the workgroup barrier (fft_team.h).
    // learn: the poll that sees the arrivals also brings the word behind the counter -- the transform the team takes next (see `pub`
    // below) -- and leaves it in sh[4] for the workgroup
    auto wait_all = [&](int g, bool learn = false) __attribute__((always_inline)) {
        FFT_LDS_FRESH();
        if (sh[3]) return;
        if (FFT_ABLATE(p.ablate & 8)) {  // profiling only (experiments build): no poll -- what the team waits cost (results invalid)
            FFT_SYNC_LDS();
            return;
        }
        if (tid < FFT_TEAM_POLL_LANES) {
            const long long tstart = FFT_CLOCK();
            for (;;) {
                const unsigned long long both = FFT_L2_COUNT_POLL2(flags);
                if ((int)((unsigned)both - ((unsigned)g << LOG2TS)) >= 0) {
                    if (learn && tid == 0) sh[4] = (unsigned)(both >> 32);
                    break;
                }
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
    auto arrive = [&]() __attribute__((always_inline)) {  // call behind a workgroup barrier, every wave's stores complete
        if (tid == 0) FFT_L2_COUNT_ADD(flags);
    };
    // ---- SLOTS = 3, the pair protocol: ONE image per seat in the window (2 MiB per XCD at n = 2^20 -- it stays in the L2, which the two slots
    // of the team protocol do not: tools/membench8.hip) and no team-wide wait.  Rounds are numbered through the launch, R = 4 it + r.
    //   * a sender UNIT (a wave on the device: its store instruction i goes to ONE seat) tells each seat it has written to, as soon as ITS
    //     stores are complete -- no workgroup barrier --: one atomic add per store instruction on the seat's counter `rcv`;
    //   * a seat pulls its image of round R when its own counter says that all (R + 1) x CNT instructions are in, and says so in its word of
    //     the team's `landed` line (plain stores of R + 1 into a copy of that line for every seat: one store instruction of TS lanes);
    //   * a unit writes round R into a seat's image when that seat's word says R (rounds < R have landed): it reads the words of the seats
    //     it is about to write to in its own seat's copy -- on the device the eight adjacent words of its quarter of the team in ONE scalar
    //     load.
    // A seat waits for its own 64 sender waves and a wave for its own 8 receivers, never for the slowest of the whole team: the store -> pull
    // chain is 3.6 us against 5.4 with the team's counter on one slot (tools/membench9.hip, profiles/r4_membench9_handoff_chain.txt).
    // The `landed` line exists once PER READING SEAT: with one line for the team, 256 waves of an XCD polled the line 32 seats store into and
    // every poll took 2.3 us (3.6 instead of 2.49 ms per launch, profiles/r4_ab_pair_protocol_polls.txt).
    // No wait in here skips a BARRIER when a timeout has been seen: sh[3] can be set by any wave at any time.
    constexpr int NINSTR = (V == 1) ? E : E / 2;                 // window stores of a thread per round
    constexpr int PAIR_CNT = NTHR / FFT_PAIR_UNIT * NINSTR;      // sender instructions per seat and round (an image is NTHR x NINSTR stores)
    // device: store instruction i of a wave goes to ONE seat, and a wave's seats are the SPQ = TS / 4 adjacent ones of block q = (r - ap) mod 4
    // of the rows (k1 = kb + MA q, kb < MA = SPQ NR): true of every device shape with TS >= 4 (a wave = RA lanes g x ILN columns of one class)
    constexpr int SPQ = TS >= 4 ? TS / 4 : 1;
#if !defined(FFT_EMU)
    static_assert(!PAIR || (TS >= 4 && MA == SPQ * NR && NINSTR == 8 && NTHR % 64 == 0), "pair protocol: teams of at least 4, eight window stores per thread and round");
#endif
    unsigned* const pair_rcv = p.ctl + TEAM_CTL_PAIR + 32 * (2 * TS + 1) * team;  // + 32 * seat
    unsigned* const pair_landed = pair_rcv + 32 * TS;                            // + 32 * reading seat + seat
    unsigned* const pair_pub = pair_rcv + 64 * TS;
    const int ap_ = FFT_UNIFORM((tid >> (LOG2RA + LOG2ILN)) & 3);
    auto pair_dst_seat = [&](int t, int i, int r) __attribute__((always_inline)) -> int {  // the seat store instruction i of round r goes to (`send`)
        const int g = t & (RA - 1), q = (r - ap_) & 3;
        int k1;
        if constexpr (V == 1) k1 = quad_kb<E, RA>(g, i) + MA * q;
        else if constexpr (GA >= 2) k1 = quad_kb<E, RA>(g, (i / RA) * 2 * RA + (i % RA)) + MA * q;
        else k1 = (g & ~1) + E * (2 * i + (g & 1)) + MA * q;
        return k1 >> LOG2NR;
    };
    auto pair_signal = [&](int r) __attribute__((always_inline)) {  // call behind FFT_WAIT_VM0: my unit's values of round r are in L2
        int t = tid0;
        FFT_OPAQUE(t);
        if constexpr (FFT_PAIR_UNIT == 1) {
            FFT_UNROLL
            for (int i = 0; i < NINSTR; i++) FFT_L2_COUNT_ADD(pair_rcv + 32 * pair_dst_seat(t, i, r));
        } else {
            const int lane = t & (FFT_PAIR_UNIT - 1);
            if (lane < NINSTR) FFT_L2_COUNT_ADD(pair_rcv + 32 * pair_dst_seat(t, lane, r));
        }
    };
    auto pair_guard = [&](int r, unsigned R) __attribute__((always_inline)) {  // the seats I am about to write round R to have pulled round R - 1
        if (R == 0) return;
        if ((QUAD_ABL & 128) && p.nb >= 0) return;  // (timing experiment: nobody waits for a seat to have pulled its image)
        FFT_LDS_FRESH();
        if (sh[3]) return;
        int t = tid0;
        FFT_OPAQUE(t);
        const long long tstart = FFT_CLOCK();
        for (;;) {
            bool ok = true;
            if constexpr (FFT_PAIR_UNIT == 1) {
                FFT_UNROLL
                for (int i = 0; i < NINSTR; i++) ok = ok && (int)(FFT_L2_FLAG_LOAD(pair_landed + 32 * s + pair_dst_seat(t, i, r)) - R) >= 0;
            } else {
#if !defined(FFT_EMU)
                const unsigned* const mine = pair_landed + 32 * s + SPQ * ((r - ap_) & 3);
                if constexpr (SPQ == 8) {
                    const fft_u32x8 w8 = fft_scalar_load8_glc(mine);
                    FFT_UNROLL
                    for (int i = 0; i < 8; i++) ok = ok && (int)(w8[i] - R) >= 0;
                } else if constexpr (SPQ == 4) {
                    const fft_u32x4s w4 = fft_scalar_load4_glc(mine);
                    FFT_UNROLL
                    for (int i = 0; i < 4; i++) ok = ok && (int)(w4[i] - R) >= 0;
                } else if constexpr (SPQ == 2) {
                    const unsigned long long w2 = fft_scalar_load2_glc(mine);
                    ok = (int)((unsigned)w2 - R) >= 0 && (int)((unsigned)(w2 >> 32) - R) >= 0;
                } else {
                    ok = (int)(fft_scalar_load_glc(mine) - R) >= 0;
                }
#endif
            }
            if (ok) break;
            if ((QUAD_ABL & 512) && p.nb >= 0) break;  // (timing experiment: one poll, whatever it says)
            if (FFT_CLOCK() - tstart > p.timeout_ticks) {
                if ((t & (FFT_PAIR_UNIT - 1)) == 0) team_report_timeout(p);
                sh[3] = 1;
                break;
            }
            FFT_SLEEP();
        }
    };
    // every sender's values of my image of round R are in L2 (the first wave polls, the workgroup waits at the barrier).  learn_it >= 0: the
    // poll also fetches the team's next transform (`pair_pub`: index + 1, tagged with the iteration it is for) into sh[4]
    auto pair_wait = [&](unsigned R, int learn_it) __attribute__((always_inline)) {
        if (tid < FFT_TEAM_POLL_LANES && (!(QUAD_ABL & 256) || p.nb < 0)) {  // (256: timing experiment, no seat waits for its senders)
            FFT_LDS_FRESH();
            const long long tstart = FFT_CLOCK();
            bool counted = false;
            while (!sh[3]) {
                if (!counted && (int)(FFT_L2_COUNT_POLL(pair_rcv + 32 * s) - (unsigned)PAIR_CNT * (R + 1u)) >= 0) counted = true;
                if (counted) {
                    if (learn_it < 0) break;
                    const unsigned long long both = FFT_L2_LOAD64(pair_pub);
                    if ((unsigned)(both >> 32) == (unsigned)learn_it + 1u) {
                        if (tid == 0) sh[4] = (unsigned)both;
                        break;
                    }
                }
                if (FFT_CLOCK() - tstart > p.timeout_ticks) {
                    if (tid == 0) team_report_timeout(p);
                    sh[3] = 1;
                    break;
                }
                FFT_SLEEP();
                FFT_LDS_FRESH();
            }
        }
        FFT_SYNC_LDS();
    };

    // ---- thread coordinates.  Every phase derives them afresh from an opaque copy of the thread id (FFT_OPAQUE): left to
    // itself the optimizer hoists every LDS address, DMA source and window pointer of every phase out of the transform
    // loop -- all of them are loop-invariant -- and keeps hundreds of registers live across it.
    // natural map (stage 1 of both steps, stage 2 and stores of the row step): lanes along the image row
    //   ncol = t & (NC - 1), nr = t >> LOG2NC (< R2)
    // column-step stage 2 / sender map: a group (a wave on the device) = R2 values of g x ILN columns of ONE class ap = j2 mod 4
    //   g = t & (R2 - 1), il = (t >> LOG2R2) & (ILN - 1), grp = t >> (LOG2R2 + LOG2ILN): ap = grp & 3, cc = il + ILN (grp >> 2), c2 = ap + 4 cc
    const int ap = FFT_UNIFORM((tid >> (LOG2RA + LOG2ILN)) & 3);
    auto sender_cc = [&](int t) __attribute__((always_inline)) { return ((t >> LOG2RA) & (ILN - 1)) + ILN * (t >> (LOG2RA + LOG2ILN + 2)); };
    // the block of M rows my row of the row step lies in (TS >= 4: the seat's; TS = 2: by the row -- a wave's on the device)
    const int sigma = FFT_UNIFORM((NR * s + (tid & (NR - 1))) >> LOG2MA);
    const cpx<T> whalf = p.tables[L2 / 2 + L1 + L2];  // W_n^(L2/2), behind the tables in the blob
    auto wn = [&](unsigned x) __attribute__((always_inline)) {  // W_n^x = W_n^(x mod L2/2) [* W_n^(L2/2)] * W_L1^(x / L2)
        const cpx<T> lo = cmul(t0[x & (L2 / 2 - 1)], wlA[(x >> LOG2L2) & (L1 - 1)]);
        const cpx<T> hi = cmul(lo, whalf);
        return (x & (L2 / 2)) ? hi : lo;
    };

    // Which transform a team takes next: team + it * n_teams (static), or -- p.dynamic -- the next one nobody has claimed: the teams of
    // different XCDs run 2 - 4 % apart (profiles/r3_quad_teams.txt), and a static split ends with seven XCDs waiting for the eighth.  The
    // first seat claims transform it + 1 from a device-wide counter during transform it and stores it in the word behind the team's arrival
    // counter before its arrival of round 1; the team wait of round 2 -- which waits for a later arrival -- reads counter and word in one
    // access (wait_all(.., learn)), the workgroup finds it in LDS in round 3 (value = index + 1).
    unsigned* const claim = p.ctl + TEAM_CTL_NEXT;
    unsigned* const pub = flags + 1;

    // LDS-DMA of column chunk a (rows 4 b + a of my NC columns) into image `im`: lane-linear 16-byte pieces, piece
    // i NTHR + tid is image row (i NTHR + tid) / PPR, columns 2 ((i NTHR + tid) mod PPR) ..
    auto dma_chunk = [&](const cpx<T>* inb, int a, int im) __attribute__((always_inline)) {
        int tid = tid0;
        FFT_OPAQUE(tid);
        const cpx<T>* src = inb + ((long long)(4 * (tid / PPR) + a) << LOG2L2) + NC * s + V * (tid % PPR);
        constexpr long long step = (long long)(4 * (NTHR / PPR)) << LOG2L2;
        const unsigned lds = img_lds0 + (unsigned)im * IMG;
        if (p.nt_mask & 1) {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16_NT(src + i * step, img_b(im), lds, (unsigned)(i * NTHR + tid) * 16u);
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16(src + i * step, img_b(im), lds, (unsigned)(i * NTHR + tid) * 16u);
        }
    };
    // LDS-DMA of my image of window slot `slot` (IMG contiguous bytes, served by the XCD's L2)
    auto dma_window = [&](int slot, int im) __attribute__((always_inline)) {
        int tid = tid0;
        FFT_OPAQUE(tid);
        const unsigned char* src = sbase + (size_t)slot * slot_stride + (size_t)s * IMG + (size_t)tid * 16;
        const unsigned lds = img_lds0 + (unsigned)im * IMG;
        if (p.nt_mask & 4) {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16_L2_NT(src + (size_t)i * NTHR * 16, img_b(im), lds, (unsigned)(i * NTHR + tid) * 16u);
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++) FFT_DMA16_L2(src + (size_t)i * NTHR * 16, img_b(im), lds, (unsigned)(i * NTHR + tid) * 16u);
        }
    };

    // final modulation of the row step: result k of the radix-4 over the rounds is due scale * i^(sigma k) (the rounds
    // deliver the classes rotated by the row block sigma)
    cpx<T> ck[4];
    FFT_UNROLL
    for (int k = 0; k < 4; k++) {
        const int pw4 = (sigma * k) & 3;
        ck[k] = mk<T>(pw4 == 0 ? p.scale : pw4 == 2 ? -p.scale : (T)0, pw4 == 1 ? p.scale : pw4 == 3 ? -p.scale : (T)0);
    }

    // ---- final radix-4 over the rounds, modulation, transposed store: X[k1 + L1 k2], k1 = NR s + rho, k2 = kb + MB ka; the
    // lanes of rows rho, rho ^ 1 pair up so that every store is 16 bytes and every wave instruction writes whole NR-row
    // segments.  The stores are the most expensive single item of a transform (profiles/r4_price_list.txt: without them the launch
    // takes 2.16 instead of 2.55 ms at n = 2^20 x 512): 8 MiB per XCD in one burst is more than the L2 buffers, the XCD's write port
    // (~ 1.1 TB/s) then paces the store instructions, and every wave of the CU sits in their issue for 5.5 - 7.5 of the transform's 39 us.
    // QUAD_DEFER_STORES sends part `a` of four (the store groups a NP / 4 .. (a + 1) NP / 4) out in front of column chunk a of the team's
    // NEXT transform (part a frees the 2 E registers chunk a's results need).  That moves the burst, it does not hide it: the port paces
    // the stores wherever they are issued, and a wave that waits in a store's issue keeps the other seven at the next barrier (+1.3 %;
    // cut into slices spread over the chunk's arithmetic, or the last part under the first team wait: +-0 / worse -- the registers spill).
    cpx<T> zt[4][E];
    constexpr int NP = (V == 1) ? E : E / 2;  // store groups of a thread: one value (fp64) or one value pair (fp32) x the 4 outputs of the radix-4
    cpx<T> yk[2][4];
    auto final_part = [&](int a, cpx<T>* outb) __attribute__((always_inline)) {
        int t = tid0;
        FFT_OPAQUE(t);
        const int ncol = t & (NR - 1), nr = t >> LOG2NR;
        const bool odd = (ncol & 1) != 0;
        cpx<T>* const line0 = outb + NR * s + (ncol & ~1);
        FFT_UNROLL
        for (int i = 0; i < NP; i++) {
            if (i < a * NP / 4 || i >= (a + 1) * NP / 4) continue;
            if constexpr (V == 1) {  // fp64: every value is a 16-byte store of its own, a wave instruction writes NR-row segments all the same
                FFT_UNROLL
                for (int r = 0; r < 4; r++) yk[0][r] = zt[r][i];
                dft_inplace<T, 4>(yk[0]);
                FFT_UNROLL
                for (int ka = 0; ka < 4; ka++) {
                    vec16<T> v;
                    v.c[0] = cmul(yk[0][ka], ck[ka]);
                    if (p.inverse) v.c[0] = cswap(v.c[0]);
                    const long long k2 = quad_kb<E, RB>(nr, i) + MB * ka;
                    vec16<T>* const dst = reinterpret_cast<vec16<T>*>(outb + NR * s + ncol + (k2 << LOG2L1));
                    if ((QUAD_ABL & 1) && p.nb >= 0) continue;
                    if (p.nt_mask & 2) FFT_STORE16_NT(dst, v);
                    else *dst = v;
                }
            } else {
                FFT_UNROLL
                for (int h = 0; h < 2; h++) {
                    FFT_UNROLL
                    for (int r = 0; r < 4; r++) yk[h][r] = zt[r][2 * i + h];
                    dft_inplace<T, 4>(yk[h]);
                    FFT_UNROLL
                    for (int r = 0; r < 4; r++) {
                        yk[h][r] = cmul(yk[h][r], ck[r]);
                        if (p.inverse) yk[h][r] = cswap(yk[h][r]);
                    }
                }
                FFT_UNROLL
                for (int ka = 0; ka < 4; ka++) {
                    vec16<T> v;
                    quad_pair<T>(yk[0][ka], yk[1][ka], odd, v);
                    const long long k2 = (odd ? quad_kb<E, RB>(nr, 2 * i + 1) : quad_kb<E, RB>(nr, 2 * i)) + MB * ka;
                    vec16<T>* const dst = reinterpret_cast<vec16<T>*>(line0 + (k2 << LOG2L1));
                    if ((QUAD_ABL & 1) && p.nb >= 0) continue;
                    if (p.nt_mask & 2) FFT_STORE16_NT(dst, v);
                    else *dst = v;
                }
            }
        }
    };
    cpx<T>* out_prev = nullptr;  // DEFER: where the results still in zt belong

    long long cur = team;  // the transform in hand
    dma_chunk(p.in + cur * n, 0, 0);
    for (int it = 0; cur < p.nb; it++) {
        FFT_LDS_FRESH();
        if (sh[3]) break;  // a team wait has timed out: the launch is void (team_report_timeout), stop here
        const cpx<T>* inb = p.in + cur * n;
        cpx<T>* outb = p.out + cur * n;
        const int G = (QUAD_ONE_SLOT ? 8 : 5) * it;  // arrivals made before this transform
        unsigned claimed = 0;
        long long nxt = p.nb;
        auto learn_next = [&]() __attribute__((always_inline)) {
            if (p.dynamic) {
                FFT_LDS_FRESH();
                const unsigned v = FFT_UNIFORM(sh[4]);
                nxt = v ? (long long)v - 1 : (long long)p.nb;
            } else {
                nxt = cur + n_teams;
            }
        };

        // ================= column step: four chunks, length-M transforms, results x W_L^(a kb) kept
        cpx<T> blk[4][E];
        FFT_UNROLL
        for (int a = 0; a < 4; a++) {
            // my pieces of the chunk have landed ... everybody's have; the other image was last read before this barrier.
            // What may still be in flight behind chunk 0's DMA are the previous transform's 4 E / 2 result stores (vmcnt counts
            // in issue order)
            // (where: teams of 32 with the deferred stores +0.9 %; fp64 n = 2^14 ... 2^16 +7 / +4 / +4 %; the other fp32 shapes +1 ... -1.3 %: not
            // there -- profiles/r4_ab_column_ahead.txt, r4_ab_column_ahead_all_shapes.txt; QUAD_COL_AHEAD = 2: every shape)
            constexpr bool AHEAD = (DEFER == 1 || (DEFER == 0 && (sizeof(T) == 8 || QUAD_COL_AHEAD >= 2))) && QUAD_COL_AHEAD;
            static_assert(!AHEAD || (QUAD_EARLY_CHUNK0 && QUAD_EARLY_CHUNK1), "the counted waits assume chunks 0 and 1 of a team's next transform are requested in rounds 2 and 3");
            if (AHEAD && DEFER) {
                // in issue order: chunk a, [part a - 1's stores], chunk a + 1 -- the eight youngest requests may still fly, except in front of the
                // last chunk and of the launch's first
                if (a == 3 || (a == 0 && it == 0)) FFT_WAIT_VM0();
                else FFT_WAIT_VM_LE(NCH);
            } else if (AHEAD) {
                // results not deferred; in issue order: chunk 0, chunk 1, the previous transform's 4 E / V result stores, chunk 2, chunk 3
                if (a == 0) {
                    if (it > 0) FFT_WAIT_VM_LE(4 * E / V + NCH);
                    else FFT_WAIT_VM0();
                } else if (a == 1) {
                    if (it > 0) FFT_WAIT_VM_LE(4 * E / V + NCH);  // (chunk 1 is older than those stores)
                    else FFT_WAIT_VM_LE(NCH);
                } else if (a == 2) {
                    FFT_WAIT_VM_LE(NCH);
                } else {
                    FFT_WAIT_VM0();
                }
            } else if (DEFER) {
                // (chunk a's pieces are the youngest thing this thread has issued: its request follows part a - 1's stores -- that order measured
                // faster than the request first --, so the wait for the pieces is a wait for everything)
                FFT_WAIT_VM0();
            } else {
            if (it > 0 && a == 0) FFT_WAIT_VM_LE(4 * E / V + (QUAD_EARLY_CHUNK1 ? NCH : 0));  // (+ chunk 1's pieces, requested in round 3)
            else FFT_WAIT_VM0();
            }
            if (!(QUAD_ABL & 64) || p.nb < 0) FFT_SYNC_LDS();
            ev();
            // the previous transform's results, part a.  QUAD_DEFER_STORES = 2: not by every wave at once -- a quarter of the waves at each of
            // four points of the chunk (here, behind stage 1's butterflies, behind the stage barrier, under stage 2's reads): the waves that
            // sit in the store queue are then never the SIMD partners... of ALL the waves that could compute meanwhile
            // (= 3: two halves, here and behind the stage barrier)
            const int sgrp = DEFER == 3 ? 2 * ((tid0 * 2) / NTHR) : (tid0 * 4) / NTHR;  // wave-uniform: NTHR / 4 is a multiple of the wave size on the device
            const bool drain = DEFER && it > 0;
            if (drain && (DEFER == 1 || sgrp == 0)) final_part(a, out_prev);
            int n_mark = 0;
            auto store_mark = [&](int) __attribute__((always_inline)) {
                if (DEFER == 2 && drain && sgrp == 1 && n_mark == 1) final_part(a, out_prev);
                n_mark++;
            };
            // the claim for the transform after this one (used at the combine: its latency is hidden).  Issued HERE, behind the wait that
            // counts the previous transform's result stores (vmcnt counts in issue order: in front of it the claim would make its wave
            // wait for the first of those stores, and the team for that wave: -10 % at n = 2^20)
            if (a == 3 && p.dynamic && s == 0 && tid == 0) claimed = FFT_ATOMIC_ADD_AGENT_RELAXED(claim, 1u) + (unsigned)n_teams;
            if (a + 1 < 4 && !(QUAD_EARLY_CHUNK1 && a == 0 && it > 0) && !(AHEAD && a >= 1)) dma_chunk(inb, a + 1, (a + 1) & 1);
            cpx<T>* img = reinterpret_cast<cpx<T>*>(img_b(a & 1));
            int t = tid0;
            FFT_OPAQUE(t);
            cpx<T> v[E];
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
            quad_stage1<T, E, RA, LOG2NC, LOG2L1, true>(img, wlA, t & (NC - 1), t >> LOG2NC, p.inverse != 0, fine);  // 1 reads landed, 2 dft, 3 twiddle
            fine(1);  // 4: writes done
            FFT_SYNC_LDS();
            fine(0);  // 5: B2 passed
            FFT_OPAQUE(t);
            const int g = t & (RA - 1), c2 = ap + 4 * sender_cc(t);
            quad_stage2_read<T, E, LOG2NC>(v, img, quad_rot<RA, NC>(c2, g), g);
            fine(1);  // 6: stage-2 reads landed
            quad_stage2_dft<T, E, RA>(v);
            fine(0);  // 7: dft
#else
            if (!(QUAD_ABL & 8) || p.nb < 0) quad_stage1<T, E, RA, LOG2NC, LOG2L1, true>(img, wlA, t & (NC - 1), t >> LOG2NC, p.inverse != 0, store_mark);
            if (!(QUAD_ABL & 2) || p.nb < 0) FFT_SYNC_LDS();
            if (DEFER >= 2 && drain && sgrp == 2) final_part(a, out_prev);
            FFT_OPAQUE(t);
            const int g = t & (RA - 1), c2 = ap + 4 * sender_cc(t);
            if (!(QUAD_ABL & 8) || p.nb < 0) {
                quad_stage2_read<T, E, LOG2NC>(v, img, quad_rot<RA, NC>(c2, g), g);
                if (DEFER == 2 && drain && sgrp == 3) final_part(a, out_prev);
                if (AHEAD && a + 2 < 4) {
                    FFT_SYNC_LDS();  // everybody has read this image for the last time
                    dma_chunk(inb, a + 2, a & 1);
                }
                quad_stage2_dft<T, E, RA>(v);
            } else {
                FFT_UNROLL
                for (int k = 0; k < E; k++) v[k] = mk<T>((T)(k + a), (T)t);
            }
#endif
            if (a == 0) {
                FFT_UNROLL
                for (int k = 0; k < E; k++) blk[0][k] = v[k];
            } else {
                // W_L^(a kb - M a ap): the class shift ap rotates the radix-4's OUTPUTS (block r = rows M (r - ap))
                quad_twiddle_kb<T, E, RA, LOG2L1>(blk[a], v, wlA, a, g, -MA * a * ap);
            }
#if QUAD_FINE_TRACE
            fine(0);  // 8: chunk twiddle
#endif
        }
        // block r of my registers goes out in round r: rows k1 = kb + M q of column j2, to the seats that own them
        auto send = [&](int r) __attribute__((always_inline)) {
            int t = tid0;
            FFT_OPAQUE(t);
            const int g = t & (RA - 1);
            const int q = (r - ap) & 3;
            unsigned char* const wslot = sbase + (size_t)(QUAD_ONE_SLOT ? 0 : (r & 1)) * slot_stride;
            const int bprime = (NC / 4) * s + sender_cc(t);  // (j2 - ap) / 4: my column's place in its class
            if constexpr (V == 1) {  // fp64: one value per 16-byte store
                FFT_UNROLL
                for (int k = 0; k < E; k++) {
                    const int k1 = quad_kb<E, RA>(g, k) + MA * q;
                    const int dst_seat = k1 >> LOG2NR, rho = k1 & (NR - 1);
                    vec16<T> v;
                    v.c[0] = blk[r][k];
                    *reinterpret_cast<vec16<T>*>(wslot + (size_t)dst_seat * IMG + (size_t)(((bprime << LOG2NR) + rho) * SZ)) = v;
                }
            } else {
            FFT_UNROLL
            for (int i = 0; i < E / 2; i++) {
                vec16<T> v;
                int k1;
                if constexpr (GA >= 2) {
                    // two of my butterflies are adjacent rows: kb(g, j) and kb(g, j + RA)
                    const int j = (i / RA) * 2 * RA + (i % RA);
                    v.c[0] = blk[r][j];
                    v.c[1] = blk[r][j + RA];
                    k1 = quad_kb<E, RA>(g, j) + MA * q;
                } else {
                    const bool odd = (g & 1) != 0;
                    quad_pair<T>(blk[r][2 * i], blk[r][2 * i + 1], odd, v);  // even lane: rows (g, g + 1) of slot 2 i; odd lane: rows (g - 1, g) of slot 2 i + 1
                    k1 = (g & ~1) + E * (2 * i + (odd ? 1 : 0)) + MA * q;
                }
                const int dst_seat = k1 >> LOG2NR, rho = k1 & (NR - 1);
                if (!(QUAD_ABL & 16) || p.nb < 0) *reinterpret_cast<vec16<T>*>(wslot + (size_t)dst_seat * IMG + (size_t)(((bprime << LOG2NR) + rho) * SZ)) = v;
            }
            }
        };
        // ---- combine: radix-4 over the chunks, then W_n^(k1 j2), k1 = kb + M q, q = (r - ap) mod 4 for block r
        {
            int t = tid0;
            FFT_OPAQUE(t);
            const int g = t & (RA - 1), c2 = ap + 4 * sender_cc(t);
            const unsigned j2 = (unsigned)(NC * s + c2);  // my column of the transform
            const cpx<T> f1 = wn((unsigned)MA * j2), f2 = cmul(f1, f1), f3 = cmul(f2, f1);
            cpx<T> base0[GA];  // W_n^((GA g + i) j2)
            base0[0] = wn((unsigned)(GA * g) * j2);
            if (GA > 1) {
                const cpx<T> u1 = wn(j2);
                FFT_UNROLL
                for (int i = 1; i < GA; i++) base0[i] = cmul(base0[i - 1], u1);
            }
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
            FFT_UNROLL
            for (int r = 0; r < 4; r++) {
                const int q = (r - ap) & 3;  // wave-uniform
                const cpx<T> fq = mk<T>(q == 0 ? (T)1 : q == 1 ? f1.re : q == 2 ? f2.re : f3.re, q == 0 ? (T)0 : q == 1 ? f1.im : q == 2 ? f2.im : f3.im);
                FFT_UNROLL
                for (int i = 0; i < GA; i++) {
                    cpx<T> w[RA];
                    quad_powers<T, RA>(w, cmul(base0[i], fq), sp);
                    FFT_UNROLL
                    for (int k = 0; k < RA; k++) blk[r][i * RA + k] = cmul(blk[r][i * RA + k], w[k]);
                }
                if (r == 0) {
                    if (PAIR) pair_guard(0, 4u * (unsigned)it);  // my seats' images were last read in the previous transform's round 3: long true
                    else if (QUAD_ONE_SLOT) wait_all(G);  // the slot was last read in the previous transform's round 3: long true
                    send(0);  // drains under block 1's twiddles
                }
                if (r == 1) {
                    // ================= exchange + row step begins: the team learns that my round-0 values are in L2 while I
                    // still twiddle blocks 2 and 3 (the first team wait absorbs the column step's skew: work behind the
                    // arrival is free)
                    if (!QUAD_ONE_SLOT) wait_all(G);  // everybody's image of the previous transform's round 3 has landed: long true
                    FFT_WAIT_VM0();
                    if (PAIR) {
                        pair_signal(0);  // (per wave: no barrier)
                    } else {
                        FFT_SYNC_LDS();
                        arrive();  // arrival G + 1
                    }
                    ev();
                    if (!QUAD_ONE_SLOT) send(1);  // slot 1 was last read in that round 3
                }
            }
        }
        ev();
        if (PAIR) pair_wait(4u * (unsigned)it, -1);
        else wait_all(G + 1);
        ev();
        dma_window(0, 0);
        if constexpr (QUAD_ONE_SLOT) {
        // One slot: round r + 1 may be written once EVERYBODY's image of round r has landed, and read once everybody's values
        // are in L2 -- arrivals S_r = G + 2 r + 1 ("my values of round r are in L2") and L_r = G + 2 r + 2 ("my image of round r
        // has landed") alternate.  The values of round r + 1 go out right behind L_r and drain under round r's first stage; the
        // image of round r + 1 is requested at the stage barrier and flies under the second stage.
        FFT_UNROLL
        for (int r = 0; r < 4; r++) {
            const unsigned R = 4u * (unsigned)it + (unsigned)r;  // (pair protocol: the round's number in the launch)
            if (r == 1 && p.dynamic && s == 0 && tid == 0) {
                if (PAIR) FFT_L2_STORE64(pair_pub, claimed + 1u, (unsigned)it + 1u);
                else FFT_L2_FLAG_STORE(pub, claimed + 1u);
            }
            // the round's image has landed ... (round 3: the next transform's chunk 0, requested behind it, may still fly: vmcnt counts in issue order)
            if (QUAD_EARLY_CHUNK0 && r == 3 && nxt < p.nb) FFT_WAIT_VM_LE(NCH);
            else FFT_WAIT_VM0();
            FFT_SYNC_LDS();  // ... everybody's of this workgroup
            if (PAIR) {
                if (tid < TS) FFT_L2_FLAG_STORE(pair_landed + 32 * tid + s, R + 1u);  // L_r: my image may be overwritten (a copy for every seat)
            } else {
                arrive();  // L_r
            }
            ev();
            if (!QUAD_EARLY_CHUNK0 && r == 3) {
                learn_next();
                if (nxt < p.nb) dma_chunk(p.in + nxt * n, 0, 0);  // image 0 was last read in round 2
            }
            if (r < 3) {
                // (pair protocol, these three hand-offs one step later each -- guard and values behind the stage barrier, signal behind the
                // stage-2 reads, image request behind the stage-2 butterflies --: 2.58 against 2.49 ms, profiles/r4_ab_pair_protocol_landed_copies.txt)
                // (guard and values from INSIDE stage 1, behind the issue of its LDS reads: +-0, profiles/r4_ab_guard_in_stage.txt)
                if (PAIR) pair_guard(r + 1, R + 1u);  // (per wave: the seats it writes to)
                else wait_all(G + 2 * r + 2, r == 2);  // the team's
                send(r + 1);
            }
            cpx<T>* img = reinterpret_cast<cpx<T>*>(img_b(r & 1));
            int t = tid0;
            FFT_OPAQUE(t);
            // QUAD_PAIR_SIGNAL_AT (pair protocol): the wave's signal from inside stage 1 -- 1: when its LDS reads are in, 2: behind its butterflies, 3: behind
            // its twiddles -- instead of behind its LDS writes
            constexpr int SIG_AT = PAIR ? QUAD_PAIR_SIGNAL_AT : 0;
            int n_mark_s = 0;
            auto signal_mark = [&](int) __attribute__((always_inline)) {
                if (SIG_AT && r < 3 && n_mark_s == SIG_AT - 1) {
                    FFT_WAIT_VM0();
                    pair_signal(r + 1);
                }
                n_mark_s++;
            };
            quad_stage1<T, E, RB, LOG2NR, LOG2L2, false>(img, wlB, t & (NR - 1), t >> LOG2NR, false, signal_mark);
            if (r < 3 && !SIG_AT) FFT_WAIT_VM0();  // my round-(r + 1) values are in L2
            if (PAIR && r < 3 && !SIG_AT) pair_signal(r + 1);
            FFT_SYNC_LDS();
            if (!PAIR && r < 3) arrive();  // S_(r+1)
            FFT_OPAQUE(t);
            const int nr = t >> LOG2NR;
            cpx<T> v[E];
            quad_stage2_read<T, E, LOG2NR>(v, img, t & (NR - 1), nr);
            if (QUAD_EARLY_CHUNK1 && r == 3 && nxt < p.nb) {
                FFT_SYNC_LDS();  // everybody has read image 1 for the last time
                dma_chunk(p.in + nxt * n, 1, 1);
            }
            if (r < 3) {
                if (PAIR) pair_wait(R + 1u, (r == 2 && p.dynamic) ? it : -1);
                else wait_all(G + 2 * r + 3);  // everybody's values of round r + 1 are in L2
                ev();
                dma_window(0, (r + 1) & 1);
                if (QUAD_EARLY_CHUNK0 && r == 2) {  // image 0 was read for the last time in front of that wait's barrier (behind this round's butterflies or twiddles instead: +-0.3 %)
                    learn_next();
                    if (nxt < p.nb) dma_chunk(p.in + nxt * n, 0, 0);
                }
            }
            quad_stage2_dft<T, E, RB>(v);
            const int apr = (r - sigma) & 3;  // the class this round delivered to my row
            if (apr != 0) quad_twiddle_kb<T, E, RB, LOG2L2>(zt[r], v, wlB, apr, nr, 0);  // W_L2^(apr kb): < 3 MB
            else {
                FFT_UNROLL
                for (int k = 0; k < E; k++) zt[r][k] = v[k];
            }
        }
        } else {
        FFT_UNROLL
        for (int r = 0; r < 4; r++) {
            if (r == 1 && p.dynamic && s == 0 && tid == 0) FFT_L2_FLAG_STORE(pub, claimed + 1u);
            // the round's image has landed and my round-(r + 1) values are in L2 ... (round 3: the next transform's chunk 0 may still fly)
            if (QUAD_EARLY_CHUNK0 && r == 3 && nxt < p.nb) FFT_WAIT_VM_LE(NCH);
            else FFT_WAIT_VM0();
            FFT_SYNC_LDS();  // ... everybody's
            arrive();  // arrival G + r + 2 (r = 3: G + 5, "my image of round 3 has landed")
            ev();
            if (!QUAD_EARLY_CHUNK0 && r == 3) {
                learn_next();
                if (nxt < p.nb) dma_chunk(p.in + nxt * n, 0, 0);  // image 0 was last read in round 2
            }
            cpx<T>* img = reinterpret_cast<cpx<T>*>(img_b(r & 1));
            int t = tid0;
            FFT_OPAQUE(t);
            if (!(QUAD_ABL & 32) || p.nb < 0) quad_stage1<T, E, RB, LOG2NR, LOG2L2, false>(img, wlB, t & (NR - 1), t >> LOG2NR, false);
            if (!(QUAD_ABL & 4) || p.nb < 0) FFT_SYNC_LDS();
            FFT_OPAQUE(t);
            const int nr = t >> LOG2NR;
            cpx<T> v[E];
            if (!(QUAD_ABL & 32) || p.nb < 0) quad_stage2_read<T, E, LOG2NR>(v, img, t & (NR - 1), nr);
            else {
                FFT_UNROLL
                for (int k = 0; k < E; k++) v[k] = mk<T>((T)(k + r), (T)t);
            }
            if (QUAD_EARLY_CHUNK1 && r == 3 && nxt < p.nb) {
                FFT_SYNC_LDS();  // everybody has read image 1 for the last time
                dma_chunk(p.in + nxt * n, 1, 1);
            }
            if (r < 3) {
                // the next round's image is requested NOW (the other image was last read in round r - 1) and flies under this
                // round's second stage; behind it the values of round r + 2, into the slot this round's image came from
                // (everybody's image of round r has landed: arrival G + r + 2 says so)
                // (requested at the START of the round instead -- the team wait right behind the arrival it waits for -- : -10 % at n = 2^20, -4 % at
                // 2^19, profiles/r4_ab_early_request.txt: the wait then costs the seats' whole landing skew, here it hides behind stage 1)
                wait_all(G + r + 2, r == 2);
                ev();
                dma_window((r + 1) & 1, (r + 1) & 1);
                if (r < 2) send(r + 2);
                if (QUAD_EARLY_CHUNK0 && r == 2) {  // image 0 was read for the last time in front of that wait's barrier
                    learn_next();
                    if (nxt < p.nb) dma_chunk(p.in + nxt * n, 0, 0);
                }
            }
            if (!(QUAD_ABL & 32) || p.nb < 0) quad_stage2_dft<T, E, RB>(v);
            const int apr = (r - sigma) & 3;  // the class this round delivered to my row
            if (apr != 0) quad_twiddle_kb<T, E, RB, LOG2L2>(zt[r], v, wlB, apr, nr, 0);  // W_L2^(apr kb): < 3 MB
            else {
                FFT_UNROLL
                for (int k = 0; k < E; k++) zt[r][k] = v[k];
            }
        }
        }
        // ---- final radix-4 over the rounds and the result stores: now (QUAD_DEFER_STORES = 0) or from the next transform's column step
        ev();
        if (DEFER) {
            out_prev = outb;
        } else {
            FFT_UNROLL
            for (int a = 0; a < 4; a++) final_part(a, outb);
        }
        ev();
        cur = nxt;
    }
    if (DEFER && out_prev) {  // the team's last transform
        FFT_LDS_FRESH();
        if (!sh[3]) {
            FFT_UNROLL
            for (int a = 0; a < 4; a++) final_part(a, out_prev);
        }
    }
}

}  // namespace fftk
