// fft_kernels_chain.h -- tile_fft_ba_kernel: the LAST pass of a forward transform and the FIRST pass of the inverse
// transform that follows it, as ONE kernel (round 2).
//
// FFT -> element-wise -> inverse FFT is the shape of Bluestein's inner loop (reference bluestein.c:107-141) and of every
// FFT convolution / correlation (applications/convolution.c:52-62, power_spectrum.c:141-151).  In the multi-pass
// (four-step) plan the forward transform's last pass holds, per tile, the C spectra columns k1 .. k1 + C - 1 (at fixed
// middle index k2) over ALL values of the slowest frequency digit k3 -- and that is exactly the tile the inverse
// transform's first pass would load (its sub-transform runs over the slowest digit of ITS input index).  So the tile
// never has to leave the CU: row FFTs (forward, last pass) -> spectral product (TileHooks store side) -> column FFTs
// (inverse, first pass) -> inter-pass twiddle -> tile-major store.  One HBM round trip of the whole image less per
// forward / inverse pair: 6 -> 5 passes for a three-pass size, 4 -> 3 for a two-pass one.  Needs both passes to have
// the same tile (sub-transform length and column count): the planner prefers splits whose first and last factor are
// equal for plans that run forward + inverse back to back (Pow2Plan::prefer_chain: 2^16 = 256 x 256, 2^18 = 512 x 512,
// 2^20 = 1024 x 1024, 2^21 = 128^3, 2^22 = 128 x 256 x 128, ...).  Measurements: profiles/r2_ab_chain.txt.
#pragma once

#include "fft_kernels.h"

namespace fftk {

template <typename T>
struct ChainParams {
    TileParams<T> b;  // the forward transform's last pass: LOAD_LCONTIG side (in, tile-major gather), stage tables, store-side hook
    TileParams<T> a;  // the inverse transform's first pass: stage + inter-pass twiddle tables, STORE_CCONTIG side (out)
    int off_tables_a;  // LDS byte offset of a's table blob (b's sits at b.off_tables)
};

// E = 8 elements per thread, up to 512 threads, radix-8 split-radix codelets (FAM_SR16) for the row part and FFT_CHAIN_FAM_A
// for the column part.
#ifndef FFT_CHAIN_ORDER
#define FFT_CHAIN_ORDER 1  // 1: spectral table requested before the row stages, next tile after the product; 0: next tile first, table in line
#endif
#ifndef FFT_CHAIN_FAM_A
#define FFT_CHAIN_FAM_A FAM_SR16  // butterfly family of the column part: radix-8 (one exchange less than AUTO's radix-4 column passes) measured +2...9 % here
#endif
#ifndef FFT_CHAIN_WAVES_PER_SIMD
#define FFT_CHAIN_WAVES_PER_SIMD 2  // measured: 157 VGPRs without spills beat 128 with (tools/ab_chain.py, profiles/r2_ab_chain.txt)
#endif
// FIXED != 0 bakes (log2L << 8 | log2C) into the instantiation, as in tile_fft_kernel
template <typename T, int FIXED = 0>
FFT_KERNEL void FFT_LAUNCH_BOUNDS2(512, FFT_CHAIN_WAVES_PER_SIMD) tile_fft_ba_kernel(ChainParams<T> q) {
    constexpr int E = 8, H = 1;
    constexpr int V = vec16<T>::V;
    constexpr int log2V = Log2<V>::value;
    constexpr int log2E = Log2<E>::value;
    constexpr int SZ = (int)sizeof(cpx<T>);
    FFT_DYN_SMEM(smem);
    const TileParams<T>& pb = q.b;
    const TileParams<T>& pa = q.a;

    const int tid_invariant = FFT_TID;
    const int nthreads = FFT_NTHREADS;
    const bool nt_load = (pb.nt & FFT_TILE_NT & 1) != 0, nt_store = (pa.nt & FFT_TILE_NT & 2) != 0;  // TileParams::nt
    const int log2L = FIXED ? (FIXED >> 8) : pb.log2L, log2C = FIXED ? (FIXED & 255) : pb.log2C;  // == pa's (the planner checks)
    const int L = 1 << log2L;
    const int log2TPC = log2L - log2E;
    const int log2J = log2C - log2V;
    const int J = 1 << log2J;
    const int CG = 1 << log2C;
    const int j_invariant = tid_invariant & (J - 1);
    const int r_invariant = tid_invariant >> log2J;
    const long long n_tiles = pb.n_tiles;
    const long long tile_step = FFT_NBLOCKS;

    {  // both table blobs -> LDS
        const vec16<T>* src = reinterpret_cast<const vec16<T>*>(pb.tables);
        vec16<T>* dst = reinterpret_cast<vec16<T>*>(smem + pb.off_tables);
        for (int i = tid_invariant; i < (pb.tables_bytes >> 4); i += nthreads) dst[i] = src[i];
        src = reinterpret_cast<const vec16<T>*>(pa.tables);
        dst = reinterpret_cast<vec16<T>*>(smem + q.off_tables_a);
        for (int i = tid_invariant; i < (pa.tables_bytes >> 4); i += nthreads) dst[i] = src[i];
    }
    const cpx<T>* tabb = reinterpret_cast<const cpx<T>*>(smem + pb.off_tables);
    const cpx<T>* taba = reinterpret_cast<const cpx<T>*>(smem + q.off_tables_a);
    StageTw<T> twb, twa;
    twb.sa = tabb; twb.sb = tabb + pb.o_sb; twb.sa_bits = pb.sa_bits; twb.log2L = log2L;
    twa.sa = taba; twa.sb = taba + pa.o_sb; twa.sa_bits = pa.sa_bits; twa.log2L = log2L;

    const int pitch = L * SZ + 16;
    const int log2CPR = log2L - log2V;
    const int cpr_mask = (1 << log2CPR) - 1;

    vec16<T> nxt[H][E];
    auto prefetch_as = [&](auto nt_tag, long long tile) __attribute__((always_inline)) {
        constexpr int NTL = decltype(nt_tag)::value;
        const TileCoord<T> tc = tile_coord(pb, tile);
        int tid = tid_invariant;
        FFT_OPAQUE(tid);
        FFT_UNROLL
        for (int i = 0; i < E; i++) {
            const int g = tid + i * nthreads;
            const int t = g >> log2CPR;
            const bool live = (tc.c0 + t) < pb.n_cols;
            const int l0 = (g & cpr_mask) * V;
            const cpx<T>* src = tc.in + (long long)t * pb.in_c + (long long)(l0 >> pb.in_blk_bits) * pb.in_blk_stride + (l0 & ((1 << pb.in_blk_bits) - 1));
            if (live) {
                nxt[0][i] = fft_ld16<NTL>(reinterpret_cast<const vec16<T>*>(src));
            } else {
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) nxt[0][i].c[vv] = mk<T>((T)0, (T)0);
            }
        }
    };
    auto prefetch = [&](long long tile) __attribute__((always_inline)) {  // one wave-uniform branch per tile, the loads stay one clause
        if (nt_load) prefetch_as(std::integral_constant<int, 1>{}, tile);
        else prefetch_as(std::integral_constant<int, 0>{}, tile);
    };

    long long tile0 = FFT_BID;
    if (tile0 < n_tiles) prefetch(tile0);
    FFT_SYNC();

    for (long long tile = tile0; tile < n_tiles; tile += tile_step) {
        const TileCoord<T> tc = tile_coord(pb, tile);
        cpx<T> x[H][E][V];
        int r = r_invariant, j = j_invariant, tid = tid_invariant;
        FFT_OPAQUE(r);
        FFT_OPAQUE(j);
        FFT_OPAQUE(tid);

        // ---- the forward transform's last pass: rows through the LDS staging image, radix-8 stages
        FFT_SYNC_LDS();
        FFT_UNROLL
        for (int i = 0; i < E; i++) {
            const int g = tid + i * nthreads;
            *reinterpret_cast<vec16<T>*>(smem + (size_t)(g >> log2CPR) * pitch + (size_t)(g & cpr_mask) * 16) = nxt[0][i];
        }
        FFT_SYNC_LDS();
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            const int l = r + (e << log2TPC);
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++) x[0][e][vv] = *reinterpret_cast<const cpx<T>*>(smem + (size_t)(V * j + vv) * pitch + (size_t)l * SZ);
        }
        // the spectral table's sixteen bytes per slot are requested BEFORE the row stages (their latency hides behind
        // them; the registers are the ones the staged tile just left), the next tile's data after the product (behind
        // the column stages and the store): slot e holds X[k], k = o * out_o + K * out_k + column, K = r + TPC * e
        const bool with_tab = pb.hk.post_mode == HOOK_MUL || pb.hk.post_mode == HOOK_MUL_CONJ;
        vec16<T> tbv[E];
        if (!FFT_CHAIN_ORDER && tile + tile_step < n_tiles) prefetch(tile + tile_step);
        if (FFT_CHAIN_ORDER && with_tab) {
            const cpx<T>* tb = pb.hk.post_tab + tc.b * pb.hk.post_tab_b + tc.oidx + tc.c0 + V * j;
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const long long K = r + ((long long)e << log2TPC);
                tbv[e] = *reinterpret_cast<const vec16<T>*>(tb + K * pb.out_k);  // 16-byte aligned: every term is a multiple of V
            }
        }
        FFT_SCHED_BARRIER();
        FFT_SYNC_LDS();
        stockham_all_stages<T, E, FAM_SR16, V, H>(x, smem, pb.group_bytes, twb, r, j, log2J, log2TPC, log2L, []() {});
        FFT_SCHED_BARRIER();

        if (!FFT_CHAIN_ORDER && with_tab) {
            const cpx<T>* tb = pb.hk.post_tab + tc.b * pb.hk.post_tab_b + tc.oidx + tc.c0 + V * j;
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const long long K = r + ((long long)e << log2TPC);
                tbv[e] = *reinterpret_cast<const vec16<T>*>(tb + K * pb.out_k);
            }
        }
        if (pb.hk.post_mode != HOOK_NONE) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) {
                    cpx<T> v = x[0][e][vv];
                    if (pb.hk.post_mode == HOOK_ABS2) v = mk<T>(v.re * v.re + v.im * v.im, (T)0);
                    else if (pb.hk.post_mode == HOOK_MUL_CONJ) v = cmul_conj(v, tbv[e].c[vv]);
                    else v = cmul(v, tbv[e].c[vv]);
                    x[0][e][vv] = v;
                }
            }
        }
        FFT_SCHED_BARRIER();
        if (FFT_CHAIN_ORDER && tile + tile_step < n_tiles) prefetch(tile + tile_step);
        FFT_SCHED_BARRIER();

        // ---- the inverse transform's first pass on the same tile: swap (inverse = forward between two swaps), radix-4
        // column stages, inter-pass twiddle W_N^(K * column), swap, tile-major store
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++) x[0][e][vv] = cswap(x[0][e][vv]);
        }
        FFT_SYNC_LDS();  // the row part's last exchange is fully consumed
        stockham_all_stages<T, E, FFT_CHAIN_FAM_A, V, H>(x, smem, pa.group_bytes, twa, r, j, log2J, log2TPC, log2L, []() {});
        // the tile's columns in the inverse pass: index of (k1, k2) among the n / L columns = o * out_o + c0
        const long long col0 = tc.oidx + tc.c0;  // b's out_o = L1 = the column count per value of o
        {
            const cpx<T>* t0 = taba + pa.o_t0;
            const cpx<T>* t1 = taba + pa.o_t1;
            const cpx<T>* t2 = taba + pa.o_t2;
            const unsigned m0 = (1u << pa.t0_bits) - 1u, m1 = (1u << pa.t1_bits) - 1u;
            const int sh2 = pa.t0_bits + pa.t1_bits;
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const unsigned K = (unsigned)(r + (e << log2TPC));
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) {
                    const unsigned m = K * (unsigned)(col0 + V * j + vv);
                    cpx<T> w = cmul(t0[m & m0], t1[(m >> pa.t0_bits) & m1]);
                    if (pa.t2_bits) w = cmul(w, t2[m >> sh2]);
                    x[0][e][vv] = cswap(cmul(x[0][e][vv], w));
                }
            }
        }
        cpx<T>* outp = pa.out + tc.b * pa.out_b + col0 * pa.out_c;
        auto store_as = [&](auto nt_tag) __attribute__((always_inline)) {
            constexpr int NTS = decltype(nt_tag)::value;
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const long long K = r + ((long long)e << log2TPC);
                if ((tc.c0 + V * j) < pb.n_cols) {
                    vec16<T> v;
                    FFT_UNROLL
                    for (int vv = 0; vv < V; vv++) v.c[vv] = x[0][e][vv];
                    fft_st16<NTS>(reinterpret_cast<vec16<T>*>(outp + K * pa.out_k + V * j), v);
                }
            }
        };
        if (nt_store) store_as(std::integral_constant<int, 1>{});
        else store_as(std::integral_constant<int, 0>{});
    }
}

}  // namespace fftk
