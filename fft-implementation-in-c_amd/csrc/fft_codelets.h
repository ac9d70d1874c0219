// fft_codelets.h -- in-register DFT codelets of length 2, 4, 8, 16.
//
// These are the butterflies of the engine.  A length-N codelet is built by the
// SPLIT-RADIX recursion (the decomposition the reference describes but does not
// implement, algorithms/core/split_radix.c:5-9: "radix-2 on the even indices,
// radix-4 on the odd ones"):
//
//   E  = DFT_{N/2}(x[2n])     O1 = DFT_{N/4}(x[4n+1])     O3 = DFT_{N/4}(x[4n+3])
//   s  = w^k O1[k] + w^{3k} O3[k]          d = -i (w^k O1[k] - w^{3k} O3[k])
//   X[k] = E[k] + s      X[k+N/2]  = E[k] - s
//   X[k+N/4] = E[k+N/4] + d      X[k+3N/4] = E[k+N/4] - d          (L-shaped butterfly)
//
// with w = exp(-2 pi i / N) (forward; the inverse transform is obtained by the
// callers through the re<->im swap identity, so only forward codelets exist).
// For N = 4 the recursion IS the radix-4 butterfly whose matrix the reference
// documents at algorithms/core/radix4.c:19-26; for N = 2 it is the radix-2
// butterfly of algorithms/core/radix2_dit.c:100-106 (without the twiddle).
// All indices are compile-time, so every value lives in a VGPR.
#pragma once

#include "fft_device.h"

namespace fftk {

// multiply by w16^e = exp(-2 pi i e / 16); trivial and 45-degree cases cost no
// or two multiplies instead of four.
template <typename T, int EXP>
FFT_DEVICE cpx<T> mul_w16(cpx<T> a) {
    constexpr int e = EXP & 15;
    constexpr T h = (T)0.70710678118654752440;
    constexpr T c1 = (T)0.92387953251128675613;  // cos(pi/8)
    constexpr T s1 = (T)0.38268343236508977173;  // sin(pi/8)
    if constexpr (e == 0) {
        return a;
    } else if constexpr (e == 4) {
        return mul_neg_i(a);
    } else if constexpr (e == 8) {
        return mk<T>(-a.re, -a.im);
    } else if constexpr (e == 12) {
        return mul_pos_i(a);
    } else if constexpr (e == 2) {
        return mk<T>((a.re + a.im) * h, (a.im - a.re) * h);
    } else if constexpr (e == 6) {
        return mk<T>((a.im - a.re) * h, -(a.re + a.im) * h);
    } else if constexpr (e == 10) {
        return mk<T>(-(a.re + a.im) * h, (a.re - a.im) * h);
    } else if constexpr (e == 14) {
        return mk<T>((a.re - a.im) * h, (a.re + a.im) * h);
    } else {
        // w = (c, s) with c = cos(2 pi e/16), s = -sin(2 pi e/16)
        constexpr T c = (e == 1 || e == 15) ? c1 : (e == 7 || e == 9) ? -c1 : (e == 3 || e == 13) ? s1 : -s1;
        constexpr T s = (e == 1 || e == 7) ? -s1 : (e == 9 || e == 15) ? s1 : (e == 3 || e == 5) ? -c1 : c1;
        return mk<T>(a.re * c - a.im * s, a.re * s + a.im * c);
    }
}

// Split-radix DFT of length N over x[0], x[S], x[2S], ...; natural-order
// result in y[0..N-1].
template <typename T, int N, int S>
struct SplitRadix {
    template <int K>
    static FFT_DEVICE void combine(const cpx<T>* ev, const cpx<T>* o1, const cpx<T>* o3, cpx<T>* y) {
        constexpr int Q = N / 4;
        cpx<T> t1 = mul_w16<T, K * (16 / N)>(o1[K]);
        cpx<T> t3 = mul_w16<T, 3 * K * (16 / N)>(o3[K]);
        cpx<T> s = cadd(t1, t3);
        cpx<T> d = csub(t1, t3);  // the odd outputs take -i d: folded into the add / subtract (cadd_mni / csub_mni)
        y[K] = cadd(ev[K], s);
        y[K + 2 * Q] = csub(ev[K], s);
        y[K + Q] = cadd_mni(ev[K + Q], d);
        y[K + 3 * Q] = csub_mni(ev[K + Q], d);
        if constexpr (K + 1 < Q) combine<K + 1>(ev, o1, o3, y);
    }
    static FFT_DEVICE void run(const cpx<T>* x, cpx<T>* y) {
        cpx<T> ev[N / 2], o1[N / 4], o3[N / 4];
        SplitRadix<T, N / 2, 2 * S>::run(x, ev);
        SplitRadix<T, N / 4, 4 * S>::run(x + S, o1);
        SplitRadix<T, N / 4, 4 * S>::run(x + 3 * S, o3);
        combine<0>(ev, o1, o3, y);
    }
};
template <typename T, int S>
struct SplitRadix<T, 2, S> {
    static FFT_DEVICE void run(const cpx<T>* x, cpx<T>* y) {
        cpx<T> a = x[0], b = x[S];
        y[0] = cadd(a, b);
        y[1] = csub(a, b);
    }
};
template <typename T, int S>
struct SplitRadix<T, 1, S> {
    static FFT_DEVICE void run(const cpx<T>* x, cpx<T>* y) { y[0] = x[0]; }
};

// forward DFT of length R in place on a contiguous register array
template <typename T, int R>
FFT_DEVICE void dft_inplace(cpx<T>* x) {
    cpx<T> y[R];
    SplitRadix<T, R, 1>::run(x, y);
    FFT_UNROLL
    for (int k = 0; k < R; k++) x[k] = y[k];
}

}  // namespace fftk
