// fft_kernels.h -- every GPU kernel of the engine (gfx950 / CDNA4).
//
//   tile_fft_kernel         LDS-resident Stockham sub-transforms; the building block of the
//                           one-, two- and three-pass (four-step) power-of-two engine
//   bitrev_kernel           stand-alone bit-reversal permutation (reference radix2_dit.c:70-77)
//   radix2_dit_stage_kernel one in-place radix-2 DIT stage in HBM (reference radix2_dit.c:84-112)
//   blu_*_kernel            Bluestein modulate / pointwise / demodulate (reference bluestein.c:107-141)
//
// None of this is MFMA work: butterflies are 2x2 / 4x4 with per-element
// twiddles, 6 flop/byte at N = 2^20, i.e. HBM-bound.  What
// matters is that every HBM access is a full 16-byte lane access inside a
// >= 64..128-byte contiguous segment, that the data makes exactly one HBM round
// trip per pass, and that twiddles come from LDS.
#pragma once

#include "fft_codelets.h"

namespace fftk {

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
template <typename T>
struct TileParams {
    const cpx<T>* in;
    cpx<T>* out;
    const cpx<T>* tw_stage;  // W_L^m = exp(-2 pi i m / L), m in [0, L)
    const cpx<T>* tw_lo;     // inter-pass twiddle, two-level: W_Ntw^m = tw_lo[m & (LO-1)] * tw_hi[m >> log2LO]
    const cpx<T>* tw_hi;
    int log2L;
    int log2C;
    int n_ct;  // column tiles per (b, o)
    int n_o;
    long long in_b, in_o, in_c, in_l;
    long long out_b, out_o, out_c, out_k;
    int n_cols;  // columns >= n_cols are padding (read as zero, never stored)
    int tw_log2lo;
    int tw_lo_len, tw_hi_len;
    int off_tw_stage, off_tw_lo, off_tw_hi;  // byte offsets of the LDS copies of the tables
    int inverse;                             // 1: inverse transform via the re<->im swap identity
    int ablate;                              // profiling only (FFT_HIP_ABLATE): 1 skip inter-pass twiddle, 2 skip stages, 4 skip stage twiddles
    T scale;                                 // applied at the store (1/N folded into the last pass)
};

template <int X>
struct Log2 {
    static constexpr int value = 1 + Log2<X / 2>::value;
};
template <>
struct Log2<1> {
    static constexpr int value = 0;
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
template <typename T, int E, int R, int V>
FFT_DEVICE void stockham_stage(cpx<T> (&x)[E][V], unsigned char* smem, const cpx<T>* tws, int r, int j, int log2J,
                               int log2TPC, int log2L, int& log2Lprev, int& log2P, bool first, bool last) {
    constexpr int G = E / R;
    constexpr int log2R = Log2<R>::value;
    const int log2Li = log2Lprev - log2R;
    const int Li_mask = (1 << log2Li) - 1;
    vec16<T>* data = reinterpret_cast<vec16<T>*>(smem);

    if (!first) {
        FFT_UNROLL
        for (int m = 0; m < G; m++) {
            const int u = r + (m << log2TPC);
            const int q = u & Li_mask;
            const int kp = u >> log2Li;
            const int base = (kp << log2Lprev) + q;
            FFT_UNROLL
            for (int a = 0; a < R; a++) {
                vec16<T> v = data[((base + (a << log2Li)) << log2J) + j];
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) x[m + G * a][vv] = v.c[vv];
            }
        }
    }

    FFT_UNROLL
    for (int m = 0; m < G; m++) {
        FFT_UNROLL
        for (int vv = 0; vv < V; vv++) {
            cpx<T> t[R];
            FFT_UNROLL
            for (int a = 0; a < R; a++) t[a] = x[m + G * a][vv];
            dft_inplace<T, R>(t);
            FFT_UNROLL
            for (int a = 0; a < R; a++) x[m + G * a][vv] = t[a];
        }
    }

    if (!last) {
        const int Lmask = (1 << log2L) - 1;
        FFT_UNROLL
        for (int m = 0; m < G; m++) {
            const int u = r + (m << log2TPC);
            const int q = u & Li_mask;
            const int tq = q << log2P;
            FFT_UNROLL
            for (int k = 1; k < R; k++) {
                const cpx<T> w = tws[(tq * k) & Lmask];
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) x[m + G * k][vv] = cmul(x[m + G * k][vv], w);
            }
        }
        if (!first) FFT_SYNC();  // everyone has finished reading the previous exchange
        FFT_UNROLL
        for (int m = 0; m < G; m++) {
            const int u = r + (m << log2TPC);
            const int q = u & Li_mask;
            const int kp = u >> log2Li;
            FFT_UNROLL
            for (int k = 0; k < R; k++) {
                vec16<T> v;
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) v.c[vv] = x[m + G * k][vv];
                const int idx = ((kp + (k << log2P)) << log2Li) + q;
                data[(idx << log2J) + j] = v;
            }
        }
        FFT_SYNC();
    }
    log2Lprev = log2Li;
    log2P += log2R;
}

// ---------------------------------------------------------------------------
// The tile kernel.  Thread (j, r): j = tid mod J selects V adjacent columns
// (J = C/V lanes cover one 16*J-byte row segment), r = tid div J in [0, L/E)
// selects which E elements of those columns the thread owns.
// ---------------------------------------------------------------------------
template <typename T, int E, int FAM, int LOADM, int STOREM, bool TWIDDLE>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(512) tile_fft_kernel(TileParams<T> p) {
    constexpr int V = vec16<T>::V;
    constexpr int log2V = Log2<V>::value;
    constexpr int log2E = Log2<E>::value;
    constexpr int RM = (FAM == FAM_SR16) ? E : (FAM == FAM_R4 ? (E < 4 ? E : 4) : 2);
    constexpr int log2RM = Log2<RM>::value;
    constexpr int SZ = (int)sizeof(cpx<T>);
    FFT_DYN_SMEM(smem);

    const int tid = FFT_TID;
    const int nthreads = FFT_NTHREADS;
    const int log2L = p.log2L;
    const int L = 1 << log2L;
    const int log2TPC = log2L - log2E;
    const int log2J = p.log2C - log2V;
    const int J = 1 << log2J;
    const int C = 1 << p.log2C;
    const int j = tid & (J - 1);
    const int r = tid >> log2J;

    long long tile = FFT_BID;
    const int ct = (int)(tile % p.n_ct);
    tile /= p.n_ct;
    const int o = (int)(tile % p.n_o);
    const long long b = tile / p.n_o;
    const int c0 = ct * C;
    const cpx<T>* in = p.in + b * p.in_b + o * p.in_o + (long long)c0 * p.in_c;
    cpx<T>* out = p.out + b * p.out_b + o * p.out_o + (long long)c0 * p.out_c;

    // ---- tables -> LDS ("twiddles staged in LDS")
    cpx<T>* tws = reinterpret_cast<cpx<T>*>(smem + p.off_tw_stage);
    for (int i = tid; i < L; i += nthreads) tws[i] = p.tw_stage[i];
    cpx<T>* tlo = reinterpret_cast<cpx<T>*>(smem + p.off_tw_lo);
    cpx<T>* thi = reinterpret_cast<cpx<T>*>(smem + p.off_tw_hi);
    if (TWIDDLE) {
        for (int i = tid; i < p.tw_lo_len; i += nthreads) tlo[i] = p.tw_lo[i];
        for (int i = tid; i < p.tw_hi_len; i += nthreads) thi[i] = p.tw_hi[i];
    }

    const int pitch = L * SZ + 16;  // raw-row pitch of the l-contiguous staging image
    const int log2CPR = log2L - log2V;  // 16-byte chunks per row (L >= V always)

    // ---- load: slot e <- element l = r + TPC*e of the thread's V columns
    cpx<T> x[E][V];
    if (LOADM == LOAD_CCONTIG) {
        const bool live = (c0 + V * j) < p.n_cols;
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            const long long l = r + ((long long)e << log2TPC);
            vec16<T> v;
            if (live) {
                v = *reinterpret_cast<const vec16<T>*>(in + l * p.in_l + V * j);
            } else {
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) v.c[vv] = mk<T>((T)0, (T)0);
            }
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++) x[e][vv] = v.c[vv];
        }
        FFT_SYNC();  // tables visible
    } else {
        const int total = C << log2CPR;
        for (int g = tid; g < total; g += nthreads) {
            const int t = g >> log2CPR;
            const int pos = g & ((1 << log2CPR) - 1);
            vec16<T> v;
            if (c0 + t < p.n_cols) {
                v = *reinterpret_cast<const vec16<T>*>(in + (long long)t * p.in_c + (long long)pos * V);
            } else {
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) v.c[vv] = mk<T>((T)0, (T)0);
            }
            *reinterpret_cast<vec16<T>*>(smem + (size_t)t * pitch + (size_t)pos * 16) = v;
        }
        FFT_SYNC();
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            const int l = r + (e << log2TPC);
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++)
                x[e][vv] = *reinterpret_cast<const cpx<T>*>(smem + (size_t)(V * j + vv) * pitch + (size_t)l * SZ);
        }
        FFT_SYNC();  // staging image is dead; the exchange area may overwrite it
    }
    if (p.inverse) {
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++) x[e][vv] = cswap(x[e][vv]);
        }
    }

    // ---- stages
    if (!(p.ablate & 2)) {
        int log2Lprev = log2L, log2P = 0;
        const int n_full = log2L / log2RM;
        const int rem = log2L - n_full * log2RM;
        const int total = n_full + (rem ? 1 : 0);
        for (int s = 0; s < n_full; s++)
            stockham_stage<T, E, RM, V>(x, smem, tws, r, j, log2J, log2TPC, log2L, log2Lprev, log2P, s == 0,
                                        s == total - 1);
        if (RM > 2 && rem == 1)
            stockham_stage<T, E, 2, V>(x, smem, tws, r, j, log2J, log2TPC, log2L, log2Lprev, log2P, total == 1, true);
        if (RM > 4 && rem == 2)
            stockham_stage<T, E, (RM > 4 ? 4 : 2), V>(x, smem, tws, r, j, log2J, log2TPC, log2L, log2Lprev, log2P,
                                                      total == 1, true);
        if (RM > 8 && rem == 3)
            stockham_stage<T, E, (RM > 8 ? 8 : 2), V>(x, smem, tws, r, j, log2J, log2TPC, log2L, log2Lprev, log2P,
                                                      total == 1, true);
    }

    // ---- inter-pass twiddle W_Ntw^(K * column), scale, inverse swap
    if (TWIDDLE && !(p.ablate & 1)) {
        const unsigned lo_mask = (1u << p.tw_log2lo) - 1u;
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            const unsigned K = (unsigned)(r + (e << log2TPC));
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++) {
                const unsigned m = K * (unsigned)(c0 + V * j + vv);
                const cpx<T> w = cmul(tlo[m & lo_mask], thi[m >> p.tw_log2lo]);
                x[e][vv] = cmul(x[e][vv], w);
            }
        }
    }
    if (p.inverse) {
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++) x[e][vv] = cswap(x[e][vv]);
        }
    }
    if (p.scale != (T)1) {
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++) x[e][vv] = cscale(x[e][vv], p.scale);
        }
    }

    // ---- store: slot e holds frequency K = r + TPC*e
    if (STOREM == STORE_CCONTIG) {
        if ((c0 + V * j) < p.n_cols) {
            FFT_UNROLL
            for (int e = 0; e < E; e++) {
                const long long K = r + ((long long)e << log2TPC);
                vec16<T> v;
                FFT_UNROLL
                for (int vv = 0; vv < V; vv++) v.c[vv] = x[e][vv];
                *reinterpret_cast<vec16<T>*>(out + K * p.out_k + V * j) = v;
            }
        }
    } else {
        FFT_SYNC();  // last exchange fully consumed
        FFT_UNROLL
        for (int e = 0; e < E; e++) {
            const int K = r + (e << log2TPC);
            FFT_UNROLL
            for (int vv = 0; vv < V; vv++)
                *reinterpret_cast<cpx<T>*>(smem + (size_t)(V * j + vv) * pitch + (size_t)K * SZ) = x[e][vv];
        }
        FFT_SYNC();
        const int total = C << log2CPR;
        for (int g = tid; g < total; g += nthreads) {
            const int t = g >> log2CPR;
            const int pos = g & ((1 << log2CPR) - 1);
            if (c0 + t < p.n_cols)
                *reinterpret_cast<vec16<T>*>(out + (long long)t * p.out_c + (long long)pos * V) =
                    *reinterpret_cast<const vec16<T>*>(smem + (size_t)t * pitch + (size_t)pos * 16);
        }
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

template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) scale_copy_kernel(const cpx<T>* in, cpx<T>* out, long long total, T scale) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) out[i] = cscale(in[i], scale);
}

// ---------------------------------------------------------------------------
// Bluestein (reference algorithms/core/bluestein.c:107-141).  chirp[k] =
// exp(-dir * i*pi*k^2/n)... stored as the reference's chirp: exp(i * (-dir) * pi k^2 / n).
//   modulate:    a[b][k] = x[b][k] * conj(chirp[k]) for k < n, 0 for n <= k < m   (:107-109 + zero fill)
//   pointwise:   a[b][k] *= B[k],  B = FFT_m(b), b[k] = b[m-k] = chirp[k]        (:116-130)
//   demodulate:  y[b][k] = a[b][k] * conj(chirp[k]) * scale                      (:139-148)
// ---------------------------------------------------------------------------
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256)
    blu_modulate_kernel(const cpx<T>* x, const cpx<T>* chirp, cpx<T>* a, int n, int log2m, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    const long long mmask = (1ll << log2m) - 1;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) {
        const long long b = i >> log2m;
        const int k = (int)(i & mmask);
        cpx<T> v = mk<T>((T)0, (T)0);
        if (k < n) v = cmul_conj(x[b * n + k], chirp[k]);
        a[i] = v;
    }
}

template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256)
    blu_pointwise_kernel(cpx<T>* a, const cpx<T>* bfft, int log2m, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    const long long mmask = (1ll << log2m) - 1;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) a[i] = cmul(a[i], bfft[i & mmask]);
}

template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) blu_demodulate_kernel(const cpx<T>* a, const cpx<T>* chirp, cpx<T>* y, int n,
                                                             int log2m, long long total /* batch*n */, T scale) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) {
        const long long b = i / n;
        const int k = (int)(i - b * n);
        y[i] = cscale(cmul_conj(a[(b << log2m) + k], chirp[k]), scale);
    }
}

}  // namespace fftk
