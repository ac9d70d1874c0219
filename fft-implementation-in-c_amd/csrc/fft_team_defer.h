// fft_team_defer.h -- team_defer_kernel: the team kernel of fft_team.h with the LAST row phase of every transform
// deferred until after the column step of the next one.
//
// In team_fft_kernel every workgroup reaches the turn from the column step (A) to the row step (B) and then has
// nothing to do while (i) its last hand-over stores drain, (ii) the slowest member of the team arrives and (iii) the
// first row tile makes its trip from L2 / Infinity Cache: 8 us of 60 at n = 2^20.  Here the turn is filled with
// useful work that does not depend on it -- row phase 3 of the PREVIOUS transform:
//
//     A(i) t0 t1 t2 t3 | B(i-1) phase 3 | B(i) phase 0, 1, 2 | A(i+1) ... | B(i) phase 3 | ...
//
// The row tile of that deferred phase is fetched under the last column tile, the first row tile of transform i under
// the deferred phase's stages.  Phase 3 is handed over into a third window (S2), because S1 is rewritten by the
// next column step before the deferred phase reads.  Four arrivals per transform (generation 4 it + k):
//   X1  column step done: my hand-over of phases 0, 1 is in L2, the deferred row tile (phase 3 of it-1) has landed here
//   X2  row tile 0 has landed here (S0 may be rewritten)
//   X3  end of phase 0: row tile 1 has landed, my hand-over of phase 2 (into S0) is in L2
//   X4  end of phase 1: row tile 2 has landed, my hand-over of phase 3 (into S2) is in L2
#pragma once

#include "fft_team.h"

namespace fftk {

// PAIR (fp32): a seat's row tiles of phases (0, 1) and of (2, 3) are adjacent blocks of CB rows (team_row_dest); the
// results of the even phase wait in 2 * E registers and are written together with the odd phase's as 2 CB-row segments
// -- 128 bytes at n = 2^19, 2^20, where CB = 8 rows are all that fit a 64 KiB tile.  A CU's store pipeline moves 128-byte
// segments half again as fast as 64-byte ones (profiles/r1e_membench4: 22 vs 14 GB/s), and the row phases are bound by
// their result stores.  The held results of phase 2 cross the next transform's column step (phase 3 is the deferred one).
// NODEFER: the same kernel WITHOUT the deferral -- phase 3 runs right after phase 2 and is handed over into S1 (free
// once row tile 1 has been read), so a team has TWO live windows instead of three (4 MiB per XCD instead of 6 against
// the 4 MiB L2), at the price of the idle turn the deferral fills; five arrivals per transform (X5: row tile 3 landed).
// alternating tile images with in-place first stage (see the kernel): measured +-0 at 2^16..2^20, -1 % at 2^19 and fp64
// (profiles/r2_ab_team_variants.txt (6)), and it costs the PAIR kernel 4 VGPRs it does not have: off, kept for experiments
#ifndef FFT_TEAM_INPLACE
#define FFT_TEAM_INPLACE 0
#endif
// ALLL2 (with NODEFER): EVERY phase is handed over during the column step (four windows per team, no kept registers):
// two arrivals per transform -- X1 "my column step's hand-over is in L2", X2 "my four row tiles have landed" -- instead of
// five, at the price of a window footprint of the whole intermediate (8 MiB per XCD at n = 2^20: Infinity Cache, not L2).
template <typename T, int E, int GEO, bool PAIR = false, bool NODEFER = false, bool ALLL2 = false>
FFT_KERNEL void FFT_LAUNCH_BOUNDS2(4096 * vec16<T>::V / E, 16 * vec16<T>::V / E) team_defer_kernel(TeamParams<T> p) {
    constexpr int NT = 4;
    constexpr bool TREE = TeamTwTree<T, GEO>::value;
    static_assert(!PAIR || vec16<T>::V == 2, "paired result stores: fp32 only");
    static_assert(!ALLL2 || (NODEFER && PAIR), "ALLL2 is a variant of the paired NODEFER schedule");
    constexpr int V16 = vec16<T>::V;
    constexpr int log2V16 = Log2<V16>::value;
    constexpr int log2E = Log2<E>::value;
    constexpr int NCH = E / V16;
    constexpr int SZ = (int)sizeof(cpx<T>);
    constexpr int EP = E / NT;
    constexpr int NK = 2;
    constexpr int log2NT = 2;
    // result stores a row phase leaves in flight when it closes (everything older must be complete): PAIR -- none after
    // the even phase of a pair (its results wait in registers), twice as many after the odd one
    constexpr int NRS_EVEN = PAIR ? 0 : NCH, NRS_ODD = PAIR ? 2 * NCH : NCH;
    FFT_DYN_SMEM(smem);

    const int tid_invariant = FFT_TID;
    const int nthreads = FFT_NTHREADS;
    const int log2L1 = GEO ? (GEO & 31) : p.log2L1;
    const int log2L2 = GEO ? ((GEO >> 5) & 31) : p.log2L2;
    const int log2CA = GEO ? ((GEO >> 10) & 31) : p.log2CA;
    const int log2CB = GEO ? ((GEO >> 15) & 31) : p.log2CB;
    const int log2TS = GEO ? ((GEO >> 20) & 31) : p.log2TS;
    const int TS = 1 << log2TS;
    const int log2TPCA = log2L1 - log2E;
    const int log2TPCB = log2L2 - log2E;
    const long long n = 1ll << (log2L1 + log2L2);
    const unsigned tile_bytes = (unsigned)SZ << (log2L1 + log2CA);
    const unsigned phase_bytes = tile_bytes << log2TS;

    // Two tile images.  FFT_TEAM_INPLACE: they ALTERNATE -- the tile being transformed sits in image `par` and runs every
    // stage exchange in place there (the first stage too), the next tile lands in the other one; its DMA can therefore go
    // out when the current tile STARTS (the image was last read by the tile before it), not only once the current tile's
    // first stage has read a shared landing image: a column tile's HBM reads are in flight for a whole tile period.
    // Otherwise: image 0 is the landing image, image 1 the exchange ("work") image of every tile.
    constexpr bool INPL = FFT_TEAM_INPLACE != 0;
    unsigned char* const buf0 = smem;
    unsigned char* const buf1 = smem + tile_bytes;
    unsigned char* const tab_bytes = smem + 2 * tile_bytes;
    const unsigned buf0_lds = FFT_LDS_ADDR(buf0);
    int par = 0;
    auto stage_src = [&]() __attribute__((always_inline)) -> unsigned char* { return INPL ? (par ? buf1 : buf0) : buf0; };
    auto stage_work = [&]() __attribute__((always_inline)) -> unsigned char* { return INPL ? (par ? buf1 : buf0) : buf1; };
    auto dma_img = [&]() __attribute__((always_inline)) -> unsigned char* { return INPL ? (par ? buf0 : buf1) : buf0; };
    auto dma_lds = [&]() __attribute__((always_inline)) -> unsigned { return INPL ? (par ? buf0_lds : buf0_lds + tile_bytes) : buf0_lds; };
    auto tile_done = [&]() __attribute__((always_inline)) { if (INPL) par ^= 1; };
    {
        const vec16<T>* src = reinterpret_cast<const vec16<T>*>(p.tables);
        vec16<T>* dst = reinterpret_cast<vec16<T>*>(tab_bytes);
        for (int i = tid_invariant; i < (p.tables_bytes >> 4); i += nthreads) dst[i] = src[i];
    }
    const cpx<T>* tab = reinterpret_cast<const cpx<T>*>(tab_bytes);
    unsigned* const sh = reinterpret_cast<unsigned*>(tab_bytes + p.tables_bytes);  // [slot, xcc, ok, timed out]
    StageTw<T> twA, twB;
    twA.sa = tab;
    twA.sb = tab + p.o_sb1;
    twA.sa_bits = team_stage_table_bits(SZ, log2L1);
    twA.log2L = log2L1;
    twB.sa = tab + p.o_sa2;
    twB.sb = tab + p.o_sb2;
    twB.sa_bits = team_stage_table_bits(SZ, log2L2);
    twB.log2L = log2L2;

    // ---- team formation (team_form, fft_team.h)
    if (tid_invariant == 0) team_form(p, sh);
    FFT_SYNC();
    FFT_LDS_FRESH();
    if (!sh[2]) return;
    const unsigned seat = (FFT_UNIFORM(sh[0]) + (unsigned)p.seat_rot) & ((1u << p.log2seats) - 1u);
    const int c = (int)(seat & (unsigned)(TS - 1));
    const int team = (int)((FFT_UNIFORM(sh[1]) << (p.log2seats - log2TS)) + (seat >> log2TS));
    const int n_teams = p.n_xcc << (p.log2seats - log2TS);
    const int M = team < p.nb ? (p.nb - team + n_teams - 1) / n_teams : 0;  // transforms of this team
    if (M == 0) return;
    if (FFT_TEST_DROP()) return;  // emulation only: a member that never arrives -- its team's waits must run into their bound

    unsigned char* const sbase = p.scratch + (size_t)team * 4 * phase_bytes;  // windows S0, S1, S2 (ALLL2: and S3)
    unsigned* const flags = p.ctl + TEAM_CTL_FLAGS + 32 * team;

    int n_ev = 0;
    auto ev = [&]() __attribute__((always_inline)) {
        if (p.trace && tid_invariant == 0 && n_ev < p.trace_events - 1) {
            p.trace[(long long)FFT_BID * p.trace_events + n_ev] = FFT_CLOCK();
            n_ev++;
        }
    };
    ev();
    if (p.trace && tid_invariant == 0 && p.trace_events > 1) p.trace[(long long)FFT_BID * p.trace_events + p.trace_events - 1] = (team << 8) | c;

    // everybody has arrived at g <=> the team's counter >= TS * g: every member waits for everybody's X(g) before its own
    // X(g + 1).  Polled by the first wave with scalar loads (see FFT_L2_COUNT_POLL), the others wait at the barrier.
    auto wait_all = [&](int g) __attribute__((always_inline)) {
        FFT_LDS_FRESH();
        if (g <= 0 || sh[3]) return;
        if (tid_invariant < FFT_TEAM_POLL_LANES) {
            const long long t0 = FFT_CLOCK();
            while ((int)(FFT_L2_COUNT_POLL(flags) - ((unsigned)g << log2TS)) < 0) {
                if (FFT_CLOCK() - t0 > p.timeout_ticks) {
                    team_report_timeout(p);
                    sh[3] = 1;
                    break;
                }
                FFT_SLEEP();
            }
        }
        FFT_SYNC_LDS();
    };
    auto arrive = [&](int) __attribute__((always_inline)) {
        if (tid_invariant == 0) FFT_L2_COUNT_ADD(flags);
    };
    auto column_block = [&](int t) __attribute__((always_inline)) { return (t << log2TS) + ((c + p.tile_rot * t) & (TS - 1)); };
    auto in_of = [&](int it) __attribute__((always_inline)) { return p.in + (long long)(team + (long long)it * n_teams) * n; };
    auto out_of = [&](int it) __attribute__((always_inline)) { return p.out + (long long)(team + (long long)it * n_teams) * n; };
    unsigned char* const S0 = sbase;
    unsigned char* const S1 = sbase + phase_bytes;
    unsigned char* const S2 = sbase + 2 * (size_t)phase_bytes;
    unsigned char* const S3 = sbase + 3 * (size_t)phase_bytes;

    cpx<T> keep[NT][NK * EP];  // the hand-over of phases 2, 3

    auto dma_column_tile = [&](const cpx<T>* inb, int t, int i0, int i1) __attribute__((always_inline)) {
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        const int log2CPR = log2CA - log2V16;
        const int c0 = column_block(t) << log2CA;
        const cpx<T>* src = inb + ((long long)(tid >> log2CPR) << log2L2) + c0 + V16 * (tid & ((1 << log2CPR) - 1));
        const long long step = (long long)(nthreads >> log2CPR) << log2L2;
        if (FFT_ABLATE(p.ablate & 8)) return;  // profiling (-DFFT_EXPERIMENTS only): no input stream
        if (p.nt_mask & 1) {  // one branch per call, not one per chunk
            FFT_UNROLL
            for (int i = 0; i < NCH; i++)
                if (i >= i0 && i < i1) FFT_DMA16_NT(src + i * step, dma_img(), dma_lds(), (unsigned)(i * nthreads + tid) * 16u);
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++)
                if (i >= i0 && i < i1) FFT_DMA16(src + i * step, dma_img(), dma_lds(), (unsigned)(i * nthreads + tid) * 16u);
        }
    };
    auto dma_row_tile = [&](const unsigned char* sb, int i0, int i1) __attribute__((always_inline)) {
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        const unsigned char* src = sb + (size_t)c * tile_bytes + (size_t)tid * 16;
        if (p.nt_mask & 4) {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++)
                if (i >= i0 && i < i1) FFT_DMA16_L2_NT(src + (size_t)i * nthreads * 16, dma_img(), dma_lds(), (unsigned)(i * nthreads + tid) * 16u);
        } else {
            FFT_UNROLL
            for (int i = 0; i < NCH; i++)
                if (i >= i0 && i < i1) FFT_DMA16_L2(src + (size_t)i * nthreads * 16, dma_img(), dma_lds(), (unsigned)(i * nthreads + tid) * 16u);
        }
    };
    // chunks of a landing DMA issued from slot s of a tile with `total` slots: two halves from slots 0 and 1
    auto slot_i0 = [&](int s) __attribute__((always_inline)) { return s == 0 ? 0 : FFT_TEAM_DMA_FIRST(NCH); };
    auto slot_i1 = [&](int s, int total) __attribute__((always_inline)) { return s == 0 ? (total >= 2 ? FFT_TEAM_DMA_FIRST(NCH) : NCH) : NCH; };
    // hand over phase ph (2 or 3) from the kept registers into window `sb`, all four column tiles.  Which kept slots belong
    // to phase ph follows from the row mapping (team_row_dest): plain -- the first EP kept slots are phase 2, the rest
    // phase 3; PAIR -- a matter of the row's block parity (for n = 2^20: of the wave), every kept slot is looked at
    auto hand_over_kept = [&](unsigned char* sb, int ph, int rA, int jA) __attribute__((always_inline)) {
        FFT_UNROLL
        for (int tt = 0; tt < NT; tt++) {
            if constexpr (PAIR) {
                cpx<T> y[NK * EP];
                FFT_UNROLL
                for (int k = 0; k < NK * EP; k++) y[k] = keep[tt][k];
                team_hand_over_rows<T, NK * EP, true>([&](int phase) -> unsigned char* { return phase == ph ? sb : nullptr; }, y, 2 * EP,
                                                      (column_block(tt) << log2CA) + jA, rA, log2TPCA, log2CB, log2TS, tile_bytes, 1 << log2CA);
            } else {
                const bool third = (ph == 3);
                cpx<T> y[EP];
                FFT_UNROLL
                for (int ee = 0; ee < EP; ee++) {  // value by value (see pair_rows)
                    const cpx<T> a = keep[tt][ee], b3 = keep[tt][EP + ee];
                    y[ee] = mk<T>(third ? b3.re : a.re, third ? b3.im : a.im);
                }
                team_hand_over<T, EP>(sb, y, (column_block(tt) << log2CA) + jA, rA, log2TPCA, log2CB, tile_bytes, 1 << log2CA);
            }
        }
    };
    vec16<T> vhold[PAIR ? E / 2 : 1];  // PAIR: the results of the even phase of a pair, until the odd phase stores both
    // one row phase: the tile has landed; stages with `traffic(s, total)` in their slots; transposed result store
    auto row_body = [&](cpx<T>* outb, int ph, auto&& traffic) __attribute__((always_inline)) {
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        const int jB = tid & ((1 << log2CB) - 1), rB = tid >> log2CB;
        cpx<T> x[1][E][1];
        if (!FFT_ABLATE(p.ablate & 2)) {
            team_all_stages<T, E, TREE>(x, stage_src(), stage_work(), twB, rB, jB, log2CB, log2TPCB, log2L2, traffic, false);
        } else {  // profiling: the memory side alone -- the hooks' traffic without the stages between them
            FFT_UNROLL
            for (int e = 0; e < E; e++) x[0][e][0] = mk<T>((T)(tid + e), (T)ph);
            FFT_SYNC_LDS();
            traffic(0, 3);
            traffic(1, 3);
            traffic(2, 3);
        }
        const long long k1 = team_tile_row0<PAIR>(ph, c, log2CB, log2TS);
        if (p.inverse) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) x[0][e][0] = cswap(x[0][e][0]);
        }
        if (p.scale != (T)1) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) x[0][e][0] = cscale(x[0][e][0], p.scale);
        }
        if constexpr (PAIR) {
            const bool odd = (jB & 1) != 0;
            if ((ph & 1) == 0) {  // even phase of a pair: nothing is stored yet
                FFT_UNROLL
                for (int q = 0; q < E / 2; q++) pair_rows<T>(x[0][2 * q][0], x[0][2 * q + 1][0], odd, 1, vhold[q]);
            } else {
                // Lane pair (e, o) = (jB even, jB + 1).  After pair_rows e holds rows (jB, jB + 1) of output line k2(2q), o of
                // line k2(2q + 1) -- for the held tile A (rows 0 .. CB-1 of the pair) and for this tile B (rows CB .. 2CB-1).
                // One more exchange (e's B piece <-> o's A piece) and every store instruction writes WHOLE 2 CB-row
                // segments: first line k2(2q) [A from e | B from o], then line k2(2q + 1) [A from e | B from o].
                const long long kpair = k1 - ((long long)1 << log2CB);  // row 0 of the pair's first tile
                cpx<T>* const line0 = outb + ((long long)rB << log2L1) + kpair + (odd ? (1 << log2CB) + (jB & ~1) : jB);
                const long long lstep = 1ll << (log2TPCB + log2L1);  // one slot further on
                FFT_UNROLL
                for (int q = 0; q < E / 2; q++) {
                    vec16<T> vb;
                    pair_rows<T>(x[0][2 * q][0], x[0][2 * q + 1][0], odd, 1, vb);
                    const vec16<T> va = vhold[q];
                    vec16<T> first, second;  // what this lane stores into line k2(2q) and into line k2(2q + 1)
                    FFT_UNROLL
                    for (int w = 0; w < 2; w++) {
                        const T s_re = odd ? va.c[w].re : vb.c[w].re, s_im = odd ? va.c[w].im : vb.c[w].im;  // o sends its A, e its B
                        const T r_re = FFT_XOR_EXCHANGE(s_re, 1, odd), r_im = FFT_XOR_EXCHANGE(s_im, 1, odd);
                        first.c[w] = mk<T>(odd ? r_re : va.c[w].re, odd ? r_im : va.c[w].im);    // e: own A;  o: e's B
                        second.c[w] = mk<T>(odd ? vb.c[w].re : r_re, odd ? vb.c[w].im : r_im);   // e: o's A;  o: own B
                    }
                    vec16<T>* const d0 = reinterpret_cast<vec16<T>*>(line0 + (long long)(2 * q) * lstep);
                    vec16<T>* const d1 = reinterpret_cast<vec16<T>*>(line0 + (long long)(2 * q + 1) * lstep);
                    if (FFT_ABLATE(p.ablate & 4)) {  // profiling: no result stream
                    } else if (FFT_ABLATE(p.nt_mask & 8)) {  // experiment: write-through stores that leave nothing in the L2
                        FFT_STORE16_SC1(d0, first);
                        FFT_STORE16_SC1(d1, second);
                    } else if (p.nt_mask & 2) {
                        FFT_STORE16_NT(d0, first);
                        FFT_STORE16_NT(d1, second);
                    } else {
                        *d0 = first;
                        *d1 = second;
                    }
                }
            }
        } else if constexpr (V16 == 2) {
            const bool odd = (jB & 1) != 0;
            vec16<T> v[E / 2];
            FFT_UNROLL
            for (int q = 0; q < E / 2; q++) pair_rows<T>(x[0][2 * q][0], x[0][2 * q + 1][0], odd, 1, v[q]);
            vec16<T>* const dst0 = reinterpret_cast<vec16<T>*>(outb + ((long long)(rB + ((odd ? 1 : 0) << log2TPCB)) << log2L1) + k1 + (jB & ~1));
            const long long dstep = (2ll << (log2TPCB + log2L1)) / V16;  // two slots further on, in 16-byte units
            if (p.nt_mask & 2) {
                FFT_UNROLL
                for (int q = 0; q < E / 2; q++) FFT_STORE16_NT(dst0 + q * dstep, v[q]);
            } else {
                FFT_UNROLL
                for (int q = 0; q < E / 2; q++) dst0[q * dstep] = v[q];
            }
        } else {
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const long long K = rB + (e << log2TPCB);
                *reinterpret_cast<cpx<T>*>(outb + (K << log2L1) + k1 + jB) = x[0][e][0];
            }
        }
        ev();
        tile_done();
    };

    if (INPL) par = 1;  // the first tile lands in image 0 and is transformed there
    dma_column_tile(in_of(0), 0, 0, NCH);
    if (INPL) par = 0;
    for (int it = 0; it < M; it++) {
        const cpx<T>* inb = in_of(it);
        cpx<T>* outb = out_of(it);
        const int G = (ALLL2 ? 2 : NODEFER ? 5 : 4) * it;  // this transform's arrivals are G + 1 .. G + 4 (NODEFER: .. G + 5; ALLL2: G + 1, G + 2)
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        const int jA = tid & ((1 << log2CA) - 1), rA = tid >> log2CA;

        // ================= column step; under its last tile the deferred row tile (phase 3 of it-1) is fetched
        // fp32: fully unrolled (no conditional register moves into keep[t]; 229 -> 201 VGPRs, +0.3..1.2 %); fp64: a loop
        // (unrolled it spills 492 bytes per lane and loses a quarter of its throughput)
        FFT_UNROLL_N((sizeof(T) == 4 ? NT : 1))
        for (int t = 0; t < NT; t++) {
            cpx<T> x[1][E][1];
            FFT_WAIT_VM0();
            FFT_SYNC_LDS();
            if (INPL && t + 1 < NT) dma_column_tile(inb, t + 1, 0, FFT_TEAM_DMA_FIRST(NCH));  // the other image is free since this barrier
            auto column_traffic = [&](int s, int total) __attribute__((always_inline)) {
                if (s > 1) return;
                if (t + 1 < NT) {
                    if (!INPL) dma_column_tile(inb, t + 1, slot_i0(s), slot_i1(s, total));
                    else if (s == 0) dma_column_tile(inb, t + 1, FFT_TEAM_DMA_FIRST(NCH), NCH);
                } else if (it > 0 && !NODEFER) {
                    if (s == 0) wait_all(G);  // X4 of it-1: everybody's hand-over of its phase 3 is in L2
                    dma_row_tile(S2, slot_i0(s), slot_i1(s, total));
                }
            };
            if (!FFT_ABLATE(p.ablate & 2)) {
                team_all_stages<T, E, TREE>(x, stage_src(), stage_work(), twA, rA, jA, log2CA, log2TPCA, log2L1, column_traffic, p.inverse != 0);
                const unsigned n2 = (unsigned)(column_block(t) << log2CA) + (unsigned)jA;
                team_interpass_twiddle<T, E, TREE>(x, tab + p.o_t0, tab + p.o_t1, p.t0_bits, (unsigned)rA * n2, n2 << log2TPCA);
            } else {  // profiling: the memory side alone
                FFT_UNROLL
                for (int e = 0; e < E; e++) x[0][e][0] = mk<T>((T)(tid + e), (T)t);
                FFT_SYNC_LDS();
                column_traffic(0, 3);
                column_traffic(1, 3);
            }
            if (t == 0 && it > 0) wait_all(G);  // S0 / S1 were last read by row tiles 2 / 1 of it-1 (X4 covers both)
            if constexpr (ALLL2) {
                cpx<T> y[E];
                FFT_UNROLL
                for (int ee = 0; ee < E; ee++) y[ee] = x[0][ee][0];
                team_hand_over_rows<T, E, true>([&](int phase) -> unsigned char* { return sbase + (size_t)phase * phase_bytes; }, y, 0,
                                                (column_block(t) << log2CA) + jA, rA, log2TPCA, log2CB, log2TS, tile_bytes, 1 << log2CA);
            } else if constexpr (PAIR) {
                cpx<T> y[2 * EP];
                FFT_UNROLL
                for (int ee = 0; ee < 2 * EP; ee++) y[ee] = x[0][ee][0];
                team_hand_over_rows<T, 2 * EP, true>([&](int phase) -> unsigned char* { return phase ? S1 : S0; }, y, 0,
                                                     (column_block(t) << log2CA) + jA, rA, log2TPCA, log2CB, log2TS, tile_bytes, 1 << log2CA);
            } else {
                FFT_UNROLL
                for (int ph = 0; ph < 2; ph++) {
                    cpx<T> y[EP];
                    FFT_UNROLL
                    for (int ee = 0; ee < EP; ee++) y[ee] = x[0][ph * EP + ee][0];
                    team_hand_over<T, EP>(ph ? S1 : S0, y, (column_block(t) << log2CA) + jA, rA, log2TPCA, log2CB, tile_bytes, 1 << log2CA);
                }
            }
            if constexpr (!ALLL2) {
                FFT_UNROLL
                for (int tt = 0; tt < NT; tt++) {
                    if (t == tt) {
                        FFT_UNROLL
                        for (int k = 0; k < NK * EP; k++) keep[tt][k] = x[0][2 * EP + k][0];
                    }
                }
            }
            ev();
            tile_done();
        }
        FFT_WAIT_VM0();
        FFT_SYNC_LDS();
        arrive(G + 1);  // X1

        if constexpr (ALLL2) {
            // every phase is in its own window since X1: four row phases without a single wait between them
            wait_all(G + 1);
            if (INPL) par ^= 1;
            dma_row_tile(S0, 0, NCH);
            if (INPL) par ^= 1;
            FFT_WAIT_VM0();
            FFT_SYNC_LDS();
            row_body(outb, 0, [&](int s, int total) {
                if (s <= 1) dma_row_tile(S1, slot_i0(s), slot_i1(s, total));
            });
            FFT_WAIT_VM_LE(NRS_EVEN);
            FFT_SYNC_LDS();
            row_body(outb, 1, [&](int s, int total) {
                if (s <= 1) dma_row_tile(S2, slot_i0(s), slot_i1(s, total));
            });
            FFT_WAIT_VM_LE(NRS_ODD);
            FFT_SYNC_LDS();
            row_body(outb, 2, [&](int s, int total) {
                if (s <= 1) dma_row_tile(S3, slot_i0(s), slot_i1(s, total));
            });
            FFT_WAIT_VM_LE(NRS_EVEN);  // row tile 3 has landed
            FFT_SYNC_LDS();
            arrive(G + 2);  // X2: my four row tiles have landed -- the windows may be rewritten once everybody is here
            row_body(outb, 3, [&](int s, int total) {
                if (s > 1) return;
                if (it + 1 < M) dma_column_tile(in_of(it + 1), 0, slot_i0(s), slot_i1(s, total));
            });
            continue;
        }

        // ================= the turn: the deferred phase 3 of it-1 runs while X1 spreads and row tile 0 makes its trip
        if (it > 0 && !NODEFER) {
            row_body(out_of(it - 1), 3, [&](int s, int total) {
                if (s > 1) return;
                if (s == 0) wait_all(G + 1);
                dma_row_tile(S0, slot_i0(s), slot_i1(s, total));
            });
            FFT_WAIT_VM_LE(NRS_ODD);  // everything but the deferred phase's result stores: row tile 0 has landed
        } else {
            wait_all(G + 1);
            if (INPL) par ^= 1;  // between two tiles: `par` already names the UPCOMING tile's image -- land there
            dma_row_tile(S0, 0, NCH);
            if (INPL) par ^= 1;
            FFT_WAIT_VM0();
        }
        FFT_SYNC_LDS();
        arrive(G + 2);  // X2

        // ================= row phases 0, 1, 2 of this transform
        row_body(outb, 0, [&](int s, int total) {
            if (s <= 1) dma_row_tile(S1, slot_i0(s), slot_i1(s, total));  // phase 1: in L2 since X1
            if (s == (total >= 2 ? 1 : 0)) {
                wait_all(G + 2);  // everybody has read row tile 0: S0 takes phase 2
                hand_over_kept(S0, 2, rA, jA);
            }
        });
        FFT_WAIT_VM_LE(NRS_EVEN);
        FFT_SYNC_LDS();
        arrive(G + 3);  // X3
        row_body(outb, 1, [&](int s, int total) {
            if (s > 1) return;
            if (s == 0) wait_all(G + 3);
            dma_row_tile(S0, slot_i0(s), slot_i1(s, total));
            // S2 was last read by the deferred row tile, landed everywhere since X1
            if (s == (total >= 2 ? 1 : 0)) hand_over_kept(NODEFER ? S1 : S2, 3, rA, jA);  // NODEFER: S1's row tile 1 has landed everywhere (X3)
        });
        FFT_WAIT_VM_LE(NRS_ODD);
        FFT_SYNC_LDS();
        arrive(G + 4);  // X4
        row_body(outb, 2, [&](int s, int total) {
            if (s > 1) return;
            if (it + 1 < M && !NODEFER) {
                dma_column_tile(in_of(it + 1), 0, slot_i0(s), slot_i1(s, total));
            } else {  // the last transform (NODEFER: every transform) fetches its own phase 3 right away
                if (s == 0) wait_all(G + 4);
                dma_row_tile(NODEFER ? S1 : S2, slot_i0(s), slot_i1(s, total));
            }
        });
        if constexpr (NODEFER) {
            FFT_WAIT_VM_LE(NRS_EVEN);  // row tile 3 has landed
            FFT_SYNC_LDS();
            arrive(G + 5);  // X5: S0 / S1 may be rewritten by the next column step once everybody is here
            row_body(outb, 3, [&](int s, int total) {
                if (s > 1) return;
                if (it + 1 < M) dma_column_tile(in_of(it + 1), 0, slot_i0(s), slot_i1(s, total));
            });
        }
    }
    // ================= phase 3 of the last transform
    if constexpr (!NODEFER) {
        FFT_WAIT_VM0();
        FFT_SYNC_LDS();
        row_body(out_of(M - 1), 3, [&](int, int) {});
    }
}

}  // namespace fftk
