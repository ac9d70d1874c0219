// fft_plans_ext.h -- plans built ON the batched 1D engine (fft_engine.h) for the "next" rows of the scope table
// (SURVEY.md 8f): 2D complex transforms, real-input / real-output 1D transforms, and the fused consumers of the
// reference's applications/ (FFT convolution, auto- / cross-correlation, periodogram).  Templated on the same runtime
// policy RT as the engine, so the unmodified source also runs in the CPU emulation (tests/emu).
#pragma once

#include "fft_engine.h"
#include "fft_kernels_ext.h"

namespace ffteng {

// ---------------------------------------------------------------------------
// Batched complex 1D transform of ANY length with a fixed direction: the power-of-two engine or Bluestein.
// ---------------------------------------------------------------------------
template <typename T, typename RT>
class AnyPlan {
  public:
    int n = 0, dir = -1;
    Pow2Plan<T, RT>* p2 = nullptr;
    BluesteinPlan<T, RT>* bl = nullptr;
    ~AnyPlan() { delete p2; delete bl; }
    bool build(RT* rt, int n_, int dir_, int batch, int algo = ALGO_AUTO) {
        n = n_;
        dir = dir_ < 0 ? -1 : 1;
        if (n >= 1 && (n & (n - 1)) == 0) {
            p2 = new Pow2Plan<T, RT>();
            return p2->build(rt, ilog2(n), algo, batch);
        }
        if (n < 1 || n > (1 << 29)) return false;
        bl = new BluesteinPlan<T, RT>();
        return bl->build(rt, n, dir, algo, batch);
    }
    void execute(const cpx<T>* in, cpx<T>* out, int nb) {
        if (p2) p2->execute(in, out, nb, dir > 0);
        else if (bl) bl->execute(in, out, nb);
    }
};

template <typename RT, class K, class... A>
static void launch_flat(RT* rt, K kernel, long long total, A... args) {
    long long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    if (g < 1) g = 1;
    rt->launch(kernel, g, 256, (size_t)0, args...);
}

// ---------------------------------------------------------------------------
// 2D complex transform of row-major rows x cols matrices (reference: stubs fft_auto.c:411-415 / fft_gpu.c:377-394; CPU
// model applications/image_fft.c:35-60).  Row-column decomposition on the batched engine:
//   rows    = ONE batched 1D execute (n = cols, batch = rows * matrices);
//   columns = a strided batch: the column pass of the four-step engine in place (rows a power of two that fits one
//             LDS tile with >= 64-byte row segments, cols a multiple of the 16-byte lane access), the same transforms as two
//             strided passes with wide tiles for more rows, otherwise transpose -> batched 1D -> transpose.
// The inverse is scaled ONCE by 1 / (rows * cols): each 1D inverse carries its own 1 / length and nothing is applied on
// top (the reference's image_fft.c:64-71 divides by rows * cols AGAIN after two already scaled 1D inverses).
// ---------------------------------------------------------------------------
template <typename T, typename RT>
class Plan2D {
  public:
    static constexpr int SZ = (int)sizeof(cpx<T>);
    RT* rt = nullptr;
    int rows = 0, cols = 0, dir = -1, max_matrices = 1;
    AnyPlan<T, RT> rowp;
    Pow2Plan<T, RT>* colp = nullptr;      // direct column pass
    AnyPlan<T, RT>* colt = nullptr;       // or: batched 1D on the transposed image
    cpx<T>* tbuf = nullptr;               // transposed image (transpose path)
    bool ok = false;
    ~Plan2D() {
        delete colp;
        delete colt;
        if (rt && tbuf) rt->dfree(tbuf);
    }
    bool build(RT* runtime, int rows_, int cols_, int dir_, int n_matrices) {
        rt = runtime; rows = rows_; cols = cols_; dir = dir_ < 0 ? -1 : 1; max_matrices = n_matrices;
        if (rows < 1 || cols < 1 || n_matrices < 1 || (long long)rows * cols > (1ll << 30) ||
            (long long)rows * n_matrices > 0x7fffffff || (long long)cols * n_matrices > 0x7fffffff) return false;
        if (!rowp.build(rt, cols, dir, rows * n_matrices)) return false;
        if (rows > 1) {
            if ((rows & (rows - 1)) == 0) {
                colp = new Pow2Plan<T, RT>();
                if (!colp->build_columns(rt, ilog2(rows), cols, n_matrices)) { delete colp; colp = nullptr; }
                // many rows: the one-pass column tile (rows x C columns in LDS) gets narrow (C = 2 at 4096 rows fp32: 16-byte row
                // segments) or does not fit at all -- the four-step split of the column transforms keeps whole lines
                // (build_columns2; measured profiles/r2_ext_plans.txt)
                static const int min_seg = FFT_EXP_ENV("FFT_HIP_COL2_MINSEG") ? atoi(FFT_EXP_ENV("FFT_HIP_COL2_MINSEG")) : 64;
                if (!colp || colp->passes[0].seg_bytes < min_seg) {
                    Pow2Plan<T, RT>* two = new Pow2Plan<T, RT>();
                    if (two->build_columns2(rt, ilog2(rows), cols, n_matrices)) {
                        delete colp;
                        colp = two;
                    } else {
                        delete two;
                    }
                }
            }
            if (!colp) {
                colt = new AnyPlan<T, RT>();
                if (!colt->build(rt, rows, dir, cols * n_matrices)) return false;
                tbuf = (cpx<T>*)rt->dmalloc((size_t)rows * cols * n_matrices * SZ);
                if (!tbuf) return false;
            }
        }
        ok = true;
        return true;
    }
    void execute(const cpx<T>* in, cpx<T>* out, int nm) {
        rowp.execute(in, out, rows * nm);
        if (rows == 1) return;
        if (colp) {
            colp->execute(out, out, nm, dir > 0);
            return;
        }
        const long long tiles = (long long)nm * ((rows + 31) / 32) * ((cols + 31) / 32);
        long long grid = tiles > 65536 ? 65536 : tiles;
        rt->launch(fftk::transpose_kernel<T>, grid, 256, (size_t)(32 * 33 * SZ), (const cpx<T>*)out, tbuf, rows, cols, tiles);
        colt->execute(tbuf, tbuf, cols * nm);
        rt->launch(fftk::transpose_kernel<T>, grid, 256, (size_t)(32 * 33 * SZ), (const cpx<T>*)tbuf, out, cols, rows, tiles);
    }
};

// ---------------------------------------------------------------------------
// Real-input forward (r2c) and real-output inverse (c2r) 1D transforms, n/2 + 1 bins (reference include/fft_auto.h:88-106;
// its r2c "implementation" copies into a temporary, plans on it and frees it -- a use-after-free, fft_auto.c:391-402 --
// and c2r returns NULL).  Even n: ONE complex transform of length n/2 on the real array read as complex, plus a split /
// merge kernel (fft_kernels_ext.h); odd n: promoted to a complex transform of length n.  c2r is scaled by 1/n like every
// inverse of the library, so c2r(r2c(x)) = x.
// ---------------------------------------------------------------------------
template <typename T, typename RT>
class RealPlan {
  public:
    static constexpr int SZ = (int)sizeof(cpx<T>);
    RT* rt = nullptr;
    int n = 0, h = 0, max_batch = 1;
    bool forward = true;  // r2c
    AnyPlan<T, RT> core;
    cpx<T>* w = nullptr;     // W_n^k, k <= n/2 (even n)
    cpx<T>* work = nullptr;  // [batch][n/2] (even n) or [batch][n] (odd n)
    bool ok = false;
    ~RealPlan() {
        if (rt && w) rt->dfree(w);
        if (rt && work) rt->dfree(work);
    }
    bool even() const { return (n & 1) == 0; }
    bool build(RT* runtime, int n_, bool r2c, int batch) {
        rt = runtime; n = n_; forward = r2c; max_batch = batch; h = n / 2;
        if (n < 1 || batch < 1) return false;
        if (even()) {
            if (!core.build(rt, h, r2c ? -1 : 1, batch)) return false;
            std::vector<cpx<T>> t;
            make_twiddle_table<T>(t, n, (long long)h + 1, 1);
            w = (cpx<T>*)rt->dmalloc(t.size() * SZ);
            work = (cpx<T>*)rt->dmalloc((size_t)batch * (size_t)h * SZ);
            if (!w || !work) return false;
            rt->h2d(w, t.data(), t.size() * SZ);
        } else {
            if (!core.build(rt, n, r2c ? -1 : 1, batch)) return false;
            work = (cpx<T>*)rt->dmalloc((size_t)batch * (size_t)n * SZ);
            if (!work) return false;
        }
        ok = true;
        return true;
    }
    // r2c: x real [nb][n] -> X complex [nb][n/2 + 1]
    void execute_r2c(const T* x, cpx<T>* X, int nb) {
        if (even()) {
            core.execute(reinterpret_cast<const cpx<T>*>(x), work, nb);
            launch_flat(rt, fftk::r2c_split_kernel<T>, (long long)nb * (h / 2 + 1), (const cpx<T>*)work, X, (const cpx<T>*)w, h, (long long)nb * (h / 2 + 1));
        } else {
            launch_flat(rt, fftk::real_to_complex_kernel<T>, (long long)nb * n, x, work, (long long)nb * n);
            core.execute(work, work, nb);
            launch_flat(rt, fftk::copy_rows_kernel<T>, (long long)nb * (h + 1), (const cpx<T>*)work, X, n, h + 1, (long long)nb * (h + 1));
        }
    }
    // c2r: X complex [nb][n/2 + 1] (Hermitian half) -> x real [nb][n], scaled by 1/n
    void execute_c2r(const cpx<T>* X, T* x, int nb) {
        if (even()) {
            launch_flat(rt, fftk::c2r_merge_kernel<T>, (long long)nb * (h / 2 + 1), X, work, (const cpx<T>*)w, h, (long long)nb * (h / 2 + 1));
            core.execute(work, reinterpret_cast<cpx<T>*>(x), nb);
        } else {
            launch_flat(rt, fftk::hermitian_extend_kernel<T>, (long long)nb * n, X, work, n, (long long)nb * n);
            core.execute(work, work, nb);
            launch_flat(rt, fftk::complex_to_real_kernel<T>, (long long)nb * n, (const cpx<T>*)work, x, (long long)nb * n);
        }
    }
};

// ---------------------------------------------------------------------------
// Fused consumers (reference applications/convolution.c:34-96, applications/power_spectrum.c:58-80, 133-190): every one
// is FFT -> element-wise -> (inverse FFT), the shape of Bluestein's inner loop, and uses the same pass hooks
// (fftk::TileHooks): the zero padding happens in the first load (the padding is never read), the spectral product in
// the forward transform's last store, the truncation in the inverse transform's last store.
//   CONV_LINEAR    y[b] = x[b] * h          x: [batch][nx], h: [nh] fixed at plan time, y: [batch][nx + nh - 1]; m = next_pow2(ny)
//   CONV_CIRCULAR  y[b] = x[b] (*) h        all of length n (a power of two, like the reference's radix-2 path)
//   AUTOCORR       acf[b] = IFFT(|FFT(x[b] zero padded to m)|^2), first n;  m = next_pow2(2 n)
//   XCORR          ccf[b] = IFFT(conj(FFT(x[b])) FFT(y[b])), first n;       m = next_pow2(2 n)
//   PSD            one-sided periodogram of Hann-windowed x[b]: [batch][n/2 + 1] real
// ---------------------------------------------------------------------------
enum FusedKind { FUSED_CONV_LINEAR = 0, FUSED_CONV_CIRCULAR = 1, FUSED_AUTOCORR = 2, FUSED_XCORR = 3, FUSED_PSD = 4 };

template <typename T, typename RT>
class FusedPlan {
  public:
    static constexpr int SZ = (int)sizeof(cpx<T>);
    RT* rt = nullptr;
    int kind = 0, nx = 0, nh = 0, ny = 0, log2m = 0, max_batch = 1;
    Pow2Plan<T, RT> core;
    cpx<T>* H = nullptr;      // FFT_m(h) (convolutions) or the window as complex values (PSD)
    cpx<T>* work = nullptr;   // [batch][m]
    cpx<T>* work2 = nullptr;  // [batch][m] (XCORR: the spectrum of x)
    bool ok = false;
    bool no_fusion = false;   // tests: element-wise steps as kernels of their own
    bool no_chain = false;    // tests: forward-last and inverse-first pass as two kernels
    ~FusedPlan() {
        if (!rt) return;
        rt->dfree(H); rt->dfree(work); rt->dfree(work2);
    }
    long long m() const { return 1ll << log2m; }
    int out_len() const { return kind == FUSED_PSD ? nx / 2 + 1 : ny; }

    // h_host: nh kernel samples (CONV_*); ignored otherwise
    bool build(RT* runtime, int kind_, int nx_, int nh_, const cpx<T>* h_host, int batch) {
        rt = runtime; kind = kind_; nx = nx_; nh = nh_; max_batch = batch;
        if (nx < 1 || batch < 1) return false;
        long long mm = 1;
        switch (kind) {
            case FUSED_CONV_LINEAR:
                if (nh < 1 || !h_host) return false;
                ny = nx + nh - 1;
                while (mm < ny) mm <<= 1;
                break;
            case FUSED_CONV_CIRCULAR:
                if (!h_host || (nx & (nx - 1)) != 0) return false;
                nh = nx; ny = nx; mm = nx;
                break;
            case FUSED_AUTOCORR:
            case FUSED_XCORR:
                ny = nx;
                while (mm < 2ll * nx) mm <<= 1;
                break;
            case FUSED_PSD:
                if ((nx & (nx - 1)) != 0) return false;
                ny = nx; mm = nx;
                break;
            default: return false;
        }
        if (mm > (1ll << 29)) return false;
        log2m = ilog2(mm);
        core.prefer_chain = kind != FUSED_PSD;  // forward + inverse back to back (the periodogram has no inverse)
        core.wants_hooks = true;
        if (!core.build(rt, log2m, ALGO_AUTO, batch)) return false;
        work = (cpx<T>*)rt->dmalloc((size_t)batch * (size_t)mm * SZ);
        if (!work) return false;
        if (kind == FUSED_XCORR) {
            work2 = (cpx<T>*)rt->dmalloc((size_t)batch * (size_t)mm * SZ);
            if (!work2) return false;
        }
        if (kind == FUSED_CONV_LINEAR || kind == FUSED_CONV_CIRCULAR) {
            std::vector<cpx<T>> hp((size_t)mm);
            for (auto& z : hp) { z.re = 0; z.im = 0; }
            for (int i = 0; i < nh; i++) hp[(size_t)i] = h_host[i];
            H = (cpx<T>*)rt->dmalloc((size_t)mm * SZ);
            if (!H) return false;
            rt->h2d(H, hp.data(), (size_t)mm * SZ);
            core.execute(H, H, 1, false);  // the kernel's spectrum, once
        } else if (kind == FUSED_PSD) {
            // Hann window 0.5 (1 - cos(2 pi i / (n - 1))) (power_spectrum.c:5-10), as complex values; + 1 padding entry
            std::vector<cpx<T>> wv((size_t)nx + 1);
            const long double two_pi = 6.283185307179586476925286766559005768L;
            for (int i = 0; i < nx; i++) {
                wv[(size_t)i].re = nx > 1 ? (T)(0.5L * (1.0L - cosl(two_pi * (long double)i / (long double)(nx - 1)))) : (T)1;
                wv[(size_t)i].im = 0;
            }
            wv[(size_t)nx] = wv[0];
            H = (cpx<T>*)rt->dmalloc(wv.size() * SZ);
            if (!H) return false;
            rt->h2d(H, wv.data(), wv.size() * SZ);
        }
        ok = true;
        return true;
    }

    bool fused() const { return core.hook_capable() && !no_fusion; }

    // forward transform of `in` (rows of n_in valid samples, pitch in_pitch), zero padded to m, into dst (pitch m); the
    // spectrum is multiplied by tab / conj(tab) / replaced by |.|^2 on the way out; optional load-side table
    void forward(const cpx<T>* in, long long in_pitch, int n_in, const cpx<T>* pre, cpx<T>* dst, const cpx<T>* tab, long long tab_b,
                 int post_mode, int nb) {
        const long long mm = m();
        if (fused()) {
            ExecHooks<T> f;
            f.pre_tab = pre; f.pre_mode = pre ? fftk::HOOK_MUL : fftk::HOOK_NONE;
            f.n_in = n_in; f.in_pitch = in_pitch;
            f.post_tab = tab; f.post_tab_b = tab_b; f.post_mode = post_mode;
            core.execute_hooked(in, dst, nb, false, f);
            return;
        }
        const unsigned per_block = 256 * BLU_PER_THREAD;
        const unsigned bpr = (unsigned)((mm + per_block - 1) / per_block);
        rt->launch(fftk::pad_mul_kernel<T>, (long long)bpr * nb, 256, (size_t)0, in, in_pitch, n_in, pre,
                   (int)(pre ? fftk::HOOK_MUL : fftk::HOOK_NONE), dst, (int)mm, bpr);
        core.execute(dst, dst, nb, false);
        if (post_mode != fftk::HOOK_NONE)
            rt->launch(fftk::mul_store_kernel<T>, (long long)bpr * nb, 256, (size_t)0, (const cpx<T>*)dst, mm, tab, tab_b, post_mode, dst, mm,
                       (int)mm, (T)1, bpr);
    }
    // inverse transform of `src` (pitch m), first n_out values of every row into out (pitch out_pitch)
    void inverse(cpx<T>* src, cpx<T>* out, long long out_pitch, int n_out, int nb) {
        const long long mm = m();
        if (fused()) {
            ExecHooks<T> g;
            g.n_out = n_out; g.out_pitch = out_pitch;
            core.execute_hooked(src, out, nb, true, g);
            return;
        }
        core.execute(src, src, nb, true);
        const unsigned per_block = 256 * BLU_PER_THREAD;
        const unsigned bpr = (unsigned)(((long long)n_out + per_block - 1) / per_block);
        rt->launch(fftk::mul_store_kernel<T>, (long long)bpr * nb, 256, (size_t)0, (const cpx<T>*)src, mm, (const cpx<T>*)nullptr, 0ll,
                   (int)fftk::HOOK_NONE, out, out_pitch, n_out, (T)1, bpr);
    }

    // forward + spectral product + inverse in one go; the middle of it as ONE kernel where the plan allows (execute_chain)
    void forward_inverse(const cpx<T>* in, int n_in, const cpx<T>* tab, long long tab_b, int post_mode, cpx<T>* out, int n_out, int nb) {
        if (fused() && !no_chain && core.round_capable() && tab_b == 0) {  // the padded transform fits one tile: ONE kernel
            ExecHooks<T> rr;
            rr.n_in = n_in; rr.in_pitch = n_in;
            rr.mid_tab = tab; rr.mid_mode = post_mode;
            rr.n_out = n_out; rr.out_pitch = n_out;
            core.execute_round(in, out, nb, rr);
            return;
        }
        if (fused() && !no_chain && core.chain_capable()) {
            ExecHooks<T> f, g;
            f.n_in = n_in; f.in_pitch = n_in;
            f.post_tab = tab; f.post_tab_b = tab_b; f.post_mode = post_mode;
            g.n_out = n_out; g.out_pitch = n_out;
            if (core.execute_chain(in, out, nb, f, g)) return;
        }
        forward(in, n_in, n_in, nullptr, work, tab, tab_b, post_mode, nb);
        inverse(work, out, n_out, n_out, nb);
    }

    // x: [nb][nx]; y: second input of XCORR ([nb][nx]), else unused; out: [nb][out_len()] complex
    // (PSD: out is [nb][nx/2 + 1] REAL values of type T, sample_rate scales it)
    void execute(const cpx<T>* x, const cpx<T>* y, void* out, int nb, T sample_rate = (T)1) {
        switch (kind) {
            case FUSED_CONV_LINEAR:
            case FUSED_CONV_CIRCULAR:
                forward_inverse(x, nx, H, 0, fftk::HOOK_MUL, (cpx<T>*)out, ny, nb);
                break;
            case FUSED_AUTOCORR:
                forward_inverse(x, nx, nullptr, 0, fftk::HOOK_ABS2, (cpx<T>*)out, ny, nb);
                break;
            case FUSED_XCORR:
                forward(x, nx, nx, nullptr, work2, nullptr, 0, fftk::HOOK_NONE, nb);                    // X
                forward_inverse(y, nx, work2, m(), fftk::HOOK_MUL_CONJ, (cpx<T>*)out, ny, nb);           // IFFT(Y conj(X))
                break;
            case FUSED_PSD: {
                forward(x, nx, nx, H, work, nullptr, 0, fftk::HOOK_NONE, nb);
                const T scale = (T)(1.0L / ((long double)sample_rate * 0.375L * (long double)nx));  // Hann window power (power_spectrum.c:70-71)
                launch_flat(rt, fftk::psd_onesided_kernel<T>, (long long)nb * (nx / 2 + 1), (const cpx<T>*)work, (T*)out, nx, scale,
                            (long long)nb * (nx / 2 + 1));
                break;
            }
            default: break;
        }
    }
};

}  // namespace ffteng
