// fft_kernels_ext.h -- the small kernels around the batched 1D engine that the "next" rows of the scope table need
// (SURVEY.md 8f): matrix transpose (2D transforms whose column length does not fit one LDS tile), the split /
// merge steps of real-input transforms, one-sided power spectra.  All HBM-bound element-wise or tiled copies.
#pragma once

#include "fft_kernels.h"

namespace fftk {

// out[b][c][r] = in[b][r][c]: 32 x 32 tiles through LDS (padded rows: no bank conflicts), coalesced on both sides.
// grid = batch * tiles_r * tiles_c workgroups of 256 threads (32 x 8).
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) transpose_kernel(const cpx<T>* in, cpx<T>* out, int rows, int cols, long long n_tiles) {
    FFT_DYN_SMEM(smem);
    cpx<T>* tile = reinterpret_cast<cpx<T>*>(smem);  // [32][33]
    const int tr = (rows + 31) / 32, tc = (cols + 31) / 32;
    const int tx = FFT_TID & 31, ty = FFT_TID >> 5;
    for (long long t = FFT_BID; t < n_tiles; t += FFT_NBLOCKS) {
        const long long b = t / ((long long)tr * tc);
        const int rem = (int)(t - b * (long long)tr * tc);
        const int r0 = (rem / tc) * 32, c0 = (rem % tc) * 32;
        const cpx<T>* src = in + b * (long long)rows * cols;
        cpx<T>* dst = out + b * (long long)rows * cols;
        FFT_SYNC();
        FFT_UNROLL
        for (int k = 0; k < 4; k++) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            if (r < rows && c < cols) tile[(ty + 8 * k) * 33 + tx] = src[(long long)r * cols + c];
        }
        FFT_SYNC();
        FFT_UNROLL
        for (int k = 0; k < 4; k++) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < rows && c < cols) dst[(long long)c * rows + r] = tile[tx * 33 + ty + 8 * k];
        }
    }
}

// Real input of even length n = 2 h, transformed as ONE complex transform of length h on z[j] = x[2j] + i x[2j+1]
// (the real array IS that complex array, no copy).  With Z = FFT_h(z):
//   E[k] = (Z[k] + conj(Z[h-k])) / 2          spectrum of the even samples
//   O[k] = (Z[k] - conj(Z[h-k])) / (2i)       spectrum of the odd samples        (Z[h] = Z[0])
//   X[k] = E[k] + W_n^k O[k],  k = 0 .. h      the n/2 + 1 non-redundant bins (reference include/fft_auto.h:88-96)
// w[k] = W_n^k = exp(-2 pi i k / n), k <= h, comes from a plan-time table.  z: [batch][h], x_out: [batch][h + 1].
// One thread per PAIR of bins (k, h - k), k = 0 .. h/2 (round 3: one thread per bin read every Z twice): with t = W_n^k O[k],
//   X[k] = E[k] + t,   X[h-k] = conj(E[k] - t)      (E[h-k] = conj E[k], O[h-k] = conj O[k], W_n^(h-k) = -conj W_n^k);
// k = 0 yields X[0] and X[h] from Z[0] alone, 2k = h a single bin.  `total` = batch * (h/2 + 1).
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) r2c_split_kernel(const cpx<T>* z, cpx<T>* x_out, const cpx<T>* w, int h, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    const int hp = h / 2 + 1;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) {
        const long long b = i / hp;
        const int k = (int)(i - b * hp);
        const cpx<T>* zb = z + b * h;
        cpx<T>* xb = x_out + b * (h + 1);
        const cpx<T> a = zb[k];
        const cpx<T> c0 = zb[k == 0 ? 0 : h - k];
        const cpx<T> c = mk<T>(c0.re, -c0.im);                                   // conj(Z[h-k])
        const cpx<T> e = cscale(cadd(a, c), (T)0.5);
        const cpx<T> o = mul_neg_i(cscale(csub(a, c), (T)0.5));                  // (a - c) / (2i)
        const cpx<T> t = cmul(w[k], o);
        xb[k] = cadd(e, t);
        if (2 * k != h) {
            const cpx<T> m = csub(e, t);
            xb[h - k] = mk<T>(m.re, -m.im);
        }
    }
}

// The inverse of the split: X[0..h] (Hermitian half spectrum of a real signal of even length n = 2h) -> Z[0..h-1] with
//   E[k] = (X[k] + conj(X[h-k])) / 2,  O[k] = (X[k] - conj(X[h-k])) / 2 * conj(W_n^k),  Z[k] = E[k] + i O[k],  Z[h-k] = conj(E[k] - i O[k])
// (one thread per pair, `total` = batch * (h/2 + 1)); the inverse complex transform of length h (scaled by 1/h like every inverse
// here) then returns the n real samples.
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) c2r_merge_kernel(const cpx<T>* x_in, cpx<T>* z, const cpx<T>* w, int h, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    const int hp = h / 2 + 1;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) {
        const long long b = i / hp;
        const int k = (int)(i - b * hp);
        const cpx<T>* xb = x_in + b * (h + 1);
        cpx<T>* zb = z + b * h;
        const cpx<T> a = xb[k];
        const cpx<T> c0 = xb[h - k];
        const cpx<T> c = mk<T>(c0.re, -c0.im);
        const cpx<T> e = cscale(cadd(a, c), (T)0.5);
        const cpx<T> io = mul_pos_i(cmul_conj(cscale(csub(a, c), (T)0.5), w[k]));
        zb[k] = cadd(e, io);
        if (k != 0 && 2 * k != h) {
            const cpx<T> m = csub(e, io);
            zb[h - k] = mk<T>(m.re, -m.im);
        }
    }
}

// odd lengths: promote the real samples to complex / keep the real parts (the transform itself is then a plain
// complex one of length n, Bluestein for a non-power-of-two)
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) real_to_complex_kernel(const T* x, cpx<T>* z, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) z[i] = mk<T>(x[i], (T)0);
}
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) complex_to_real_kernel(const cpx<T>* z, T* x, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) x[i] = z[i].re;
}
// rows of `len_in` -> rows of `len_out` <= len_in (keep the first len_out of every row)
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) copy_rows_kernel(const cpx<T>* in, cpx<T>* out, int len_in, int len_out, long long total_out) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total_out; i += stride) {
        const long long b = i / len_out;
        out[i] = in[b * len_in + (i - b * len_out)];
    }
}
// Hermitian extension: rows of h + 1 bins -> rows of n bins, X[n - k] = conj(X[k])  (odd-length c2r)
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) hermitian_extend_kernel(const cpx<T>* half, cpx<T>* full, int n, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    const int hb = n / 2 + 1;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) {
        const long long b = i / n;
        const int k = (int)(i - b * n);
        cpx<T> v = half[b * hb + (k < hb ? k : n - k)];
        if (k >= hb) v.im = -v.im;
        full[i] = v;
    }
}

// One-sided power spectral density of a (windowed) spectrum: psd[k] = |X[k]|^2 * scale, doubled for 0 < k < n/2
// (reference applications/power_spectrum.c:73-80).  X: [batch][n], psd: [batch][n/2 + 1] real.
template <typename T>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) psd_onesided_kernel(const cpx<T>* X, T* psd, int n, T scale, long long total) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    const int hb = n / 2 + 1;
    for (long long i = FFT_BID * FFT_NTHREADS + FFT_TID; i < total; i += stride) {
        const long long b = i / hb;
        const int k = (int)(i - b * hb);
        const cpx<T> v = X[b * n + k];
        T pw = (v.re * v.re + v.im * v.im) * scale;
        if (k > 0 && k < n / 2) pw *= (T)2;  // the reference's own condition (integer n / 2)
        psd[i] = pw;
    }
}

// Device streams, 16 bytes per lane, `unroll` independent accesses in flight per thread, grid-stride: the practical ceiling
// of the box the benchmark runs on (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).  MODE 0 copy, 1 read only (the
// values are folded into one word that is stored only if it is a NaN pattern the inputs never hold), 2 write only.  NT: the
// non-temporal hint on both sides -- what the FFT kernels' own streams carry.
template <int UNROLL, int MODE, int NT>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) stream16_kernel(const vec16<float>* in, vec16<float>* out, long long n16) {
    const long long stride = FFT_NBLOCKS * FFT_NTHREADS;
    long long i = FFT_BID * FFT_NTHREADS + FFT_TID;
    vec16<float> acc;
    acc.c[0] = mk<float>(0.f, 0.f);
    acc.c[1] = mk<float>(1.f, 2.f);
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        vec16<float> v[UNROLL];
        FFT_UNROLL
        for (int u = 0; u < UNROLL; u++) v[u] = MODE == 2 ? acc : fft_ld16<NT>(in + i + u * stride);
        if (MODE == 1) {
            FFT_UNROLL
            for (int u = 0; u < UNROLL; u++) {
                acc.c[0].re += v[u].c[0].re;
                acc.c[0].im += v[u].c[1].im;
            }
        } else {
            FFT_UNROLL
            for (int u = 0; u < UNROLL; u++) fft_st16<NT>(out + i + u * stride, v[u]);
        }
    }
    for (; i < n16; i += stride) {
        if (MODE == 1) acc.c[0].re += fft_ld16<NT>(in + i).c[0].re;
        else fft_st16<NT>(out + i, MODE == 2 ? acc : fft_ld16<NT>(in + i));
    }
    if (MODE == 1 && acc.c[0].re != acc.c[0].re && acc.c[0].im == 12345.f) out[0] = acc;
}

// The streams in the shape of the single-pass FFT kernels (whose n = 512 instance moves 6.1 TB/s, more than any grid-stride shape above
// reached -- round-3 review): a 256-thread workgroup walks tiles of U x 4 KiB `tile = block, block + grid, ..`; per tile a thread has U
// 16-byte accesses in flight, a wave instruction covers 1 KiB, and the NEXT tile's loads are issued before this tile's stores
// (MODE as above).
template <int MODE, int NT, int U = 8>
FFT_KERNEL void FFT_LAUNCH_BOUNDS(256) stream_tile_kernel(const vec16<float>* in, vec16<float>* out, long long n_tiles) {
    const int tid = FFT_TID;
    vec16<float> acc;
    acc.c[0] = mk<float>(0.f, 0.f);
    acc.c[1] = mk<float>(1.f, 2.f);
    vec16<float> cur[U], nxt[U];  // tiles of U x 4 KiB
    long long t = FFT_BID;
    if (MODE != 2 && t < n_tiles) {
        FFT_UNROLL
        for (int i = 0; i < U; i++) cur[i] = fft_ld16<NT>(in + t * (U * 256) + i * 256 + tid);
    }
    for (; t < n_tiles; t += FFT_NBLOCKS) {
        const long long tn = t + FFT_NBLOCKS;
        if (MODE != 2 && tn < n_tiles) {
            FFT_UNROLL
            for (int i = 0; i < U; i++) nxt[i] = fft_ld16<NT>(in + tn * (U * 256) + i * 256 + tid);
        }
        if (MODE == 1) {
            FFT_UNROLL
            for (int i = 0; i < U; i++) {
                acc.c[0].re += cur[i].c[0].re;
                acc.c[0].im += cur[i].c[1].im;
            }
        } else {
            FFT_UNROLL
            for (int i = 0; i < U; i++) fft_st16<NT>(out + t * (U * 256) + i * 256 + tid, MODE == 2 ? acc : cur[i]);
        }
        if (MODE != 2) {
            FFT_UNROLL
            for (int i = 0; i < U; i++) cur[i] = nxt[i];
        }
    }
    if (MODE == 1 && acc.c[0].re != acc.c[0].re && acc.c[0].im == 12345.f) out[0] = acc;
}

#if !defined(FFT_EMU)
// The copy in the shape of the engine's own kernels: one 512-thread workgroup per CU walks 64 KiB tiles, tile t + 1 lands in
// LDS by LDS-DMA (nt) while tile t leaves it through registers (ds_read_b128 + nt stores): two tiles of reads in flight per CU.
FFT_KERNEL void FFT_LAUNCH_BOUNDS(512) copy_dma_kernel(const vec16<float>* in, vec16<float>* out, long long n_tiles) {
    FFT_DYN_SMEM(smem);  // 2 x 64 KiB
    const int tid = FFT_TID;
    const unsigned lds0 = FFT_LDS_ADDR(smem);
    long long t = FFT_BID;
    auto dma = [&](long long tile, int im) __attribute__((always_inline)) {
        const vec16<float>* src = in + tile * 4096 + tid;
        FFT_UNROLL
        for (int i = 0; i < 8; i++) fft_dma16<2>(src + i * 512, lds0 + (unsigned)im * 65536u + (unsigned)(i * 512 + tid) * 16u);
    };
    if (t < n_tiles) dma(t, 0);
    int im = 0;
    bool first = true;
    for (; t < n_tiles; t += FFT_NBLOCKS, im ^= 1) {
        // this tile has landed: behind its 8 pieces this thread has issued the 8 stores of the previous tile -- except on the first
        // tile, where the pieces are the youngest thing in flight (vmcnt(8) passed at once there and the reads ran early)
        if (first) FFT_WAIT_VM0();
        else FFT_WAIT_VM_LE(8);
        first = false;
        FFT_SYNC_LDS();
        if (t + FFT_NBLOCKS < n_tiles) dma(t + FFT_NBLOCKS, im ^ 1);
        const vec16<float>* img = reinterpret_cast<const vec16<float>*>(smem + im * 65536);
        vec16<float> v[8];
        FFT_UNROLL
        for (int i = 0; i < 8; i++) v[i] = img[i * 512 + tid];
        FFT_UNROLL
        for (int i = 0; i < 8; i++) FFT_STORE16_NT(out + t * 4096 + i * 512 + tid, v[i]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
#endif

}  // namespace fftk
