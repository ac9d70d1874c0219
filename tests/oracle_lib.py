"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, where present,
the real reference build (oracle/_ref/libref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

FFT_FORWARD = -1
FFT_INVERSE = 1

ALGO = {"dit": 0, "dif": 1, "split_radix": 2, "radix4": 3, "bluestein": 4, "exact": 5, "naive": 6}

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


def _build(target_file, make_target):
    path = os.path.join(ORACLE_DIR, target_file)
    src = os.path.join(ORACLE_DIR, "oracle_fft.c")
    if not os.path.exists(path) or (os.path.exists(src) and target_file.startswith("liboracle")
                                    and os.path.getmtime(path) < os.path.getmtime(src)):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, make_target], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return path


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(_build("liboracle.so", "liboracle.so"))
        lib.oracle_fft_batch.argtypes = [_dp, C.c_int, C.c_long, C.c_int, C.c_int]
        lib.oracle_fft_batch.restype = C.c_int
        lib.oracle_fft_batch_f32.argtypes = [_fp, C.c_int, C.c_long, C.c_int]
        lib.oracle_fft_batch_f32.restype = C.c_int
        lib.oracle_bit_reverse.argtypes = [C.c_uint, C.c_int]
        lib.oracle_bit_reverse.restype = C.c_uint
        lib.oracle_bit_reverse_asref.argtypes = [C.c_uint, C.c_int]
        lib.oracle_bit_reverse_asref.restype = C.c_uint
        lib.oracle_bit_reverse_table.argtypes = [C.POINTER(C.c_uint32), C.c_int]
        lib.oracle_bit_reverse_table.restype = None
        lib.oracle_next_power_of_two.argtypes = [C.c_int]
        lib.oracle_next_power_of_two.restype = C.c_int
        lib.oracle_twiddle_factor.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp]
        lib.oracle_twiddle_factor.restype = None
        for name, ptr in (("oracle_gen_two_tone", _dp), ("oracle_gen_two_tone_f32", _fp), ("oracle_gen_lcg", _dp)):
            f = getattr(lib, name)
            f.argtypes = [ptr, C.c_long, C.c_long, C.c_long]
            f.restype = None
        lib.oracle_two_tone_bins.argtypes = [C.c_long, C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long)]
        lib.oracle_two_tone_bins.restype = None
        _oracle = lib
    return _oracle


def oracle_fft(x, direction=FFT_FORWARD, algo="dit"):
    """x: complex128 array [..., n] (any leading batch dims).  Returns a new array."""
    x = np.ascontiguousarray(x, dtype=np.complex128).copy()
    n = x.shape[-1]
    batch = x.size // n if n else 0
    rc = oracle().oracle_fft_batch(x.ctypes.data_as(_dp), n, batch, int(direction), ALGO[algo])
    if rc != 0:
        raise ValueError("oracle rejected n=%d for algo %s" % (n, algo))
    return x


def oracle_fft_f32(x, direction=FFT_FORWARD):
    x = np.ascontiguousarray(x, dtype=np.complex64).copy()
    n = x.shape[-1]
    rc = oracle().oracle_fft_batch_f32(x.ctypes.data_as(_fp), n, x.size // n, int(direction))
    if rc != 0:
        raise ValueError("oracle rejected n=%d" % n)
    return x


def bit_reverse_table(log2n):
    t = np.empty(1 << log2n, dtype=np.uint32)
    oracle().oracle_bit_reverse_table(t.ctypes.data_as(C.POINTER(C.c_uint32)), log2n)
    return t


def gen_two_tone(n, b0, batch, dtype=np.complex128):
    if dtype == np.complex64:
        x = np.empty((batch, n), dtype=np.complex64)
        oracle().oracle_gen_two_tone_f32(x.ctypes.data_as(_fp), n, b0, batch)
    else:
        x = np.empty((batch, n), dtype=np.complex128)
        oracle().oracle_gen_two_tone(x.ctypes.data_as(_dp), n, b0, batch)
    return x


def two_tone_bins(n, b):
    f, g = C.c_long(), C.c_long()
    oracle().oracle_two_tone_bins(n, b, C.byref(f), C.byref(g))
    return f.value, g.value


def gen_lcg(n, b0, batch):
    x = np.empty((batch, n), dtype=np.complex128)
    oracle().oracle_gen_lcg(x.ctypes.data_as(_dp), n, b0, batch)
    return x


# ---- the real reference (only where oracle/_ref/libref.so has been built) ----
_ref = None
REF_FUNCS = {"dit": "radix2_dit_fft", "dif": "radix2_dif_fft", "split_radix": "split_radix_fft",
             "radix4": "radix4_fft", "bluestein": "bluestein_fft", "naive": "naive_dft",
             "recursive": "fft_recursive"}


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref.so"))


def ref(fast=False):
    global _ref
    if fast:
        return C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref_fast.so"))
    if _ref is None:
        _ref = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref.so"))
    return _ref


def ref_fft(x, direction=FFT_FORWARD, algo="dit", lib=None):
    """Run the REAL reference routine (in place semantic, returns a copy)."""
    lib = lib or ref()
    x = np.ascontiguousarray(x, dtype=np.complex128).copy()
    n = x.shape[-1]
    f = getattr(lib, REF_FUNCS[algo])
    f.argtypes = [C.c_void_p, C.c_int, C.c_int]
    f.restype = None
    flat = x.reshape(-1, n)
    for b in range(flat.shape[0]):
        f(flat[b].ctypes.data, n, int(direction))
    return x


# ---------------------------------------------------------------------------
# Restatements of the reference's consumers of the 1D transform ("next" rows), composed from the oracle's
# 1D FFT (the transform they call is radix2_dit_fft = ALGO "dit"; "bluestein" for other lengths).  Checker only.
# ---------------------------------------------------------------------------
def _algo_for(n):
    return "dit" if n >= 1 and (n & (n - 1)) == 0 else "bluestein"


def oracle_fft2d(x, direction=FFT_FORWARD):
    """Row-column 2D transform, reference applications/image_fft.c:35-60: every row, then every column (extract,
    transform, put back).  The inverse is scaled ONCE by 1/(rows*cols) -- the 1D inverses carry 1/cols and 1/rows; the
    reference's extra division (:64-71) would scale twice and is the documented defect this library does not copy."""
    x = np.asarray(x, dtype=np.complex128)
    rows, cols = x.shape[-2:]
    y = oracle_fft(x, direction, _algo_for(cols))
    yt = np.ascontiguousarray(np.swapaxes(y, -1, -2))
    yt = oracle_fft(yt, direction, _algo_for(rows))
    return np.ascontiguousarray(np.swapaxes(yt, -1, -2))


def oracle_r2c(x):
    """Real input -> the n/2 + 1 non-redundant bins (reference include/fft_auto.h:88-96; fft_auto.c:391-402 converts
    the real samples to complex and runs the ordinary transform)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[-1]
    return oracle_fft(x.astype(np.complex128), FFT_FORWARD, _algo_for(n))[..., : n // 2 + 1]


def oracle_c2r(X, n):
    """Hermitian half spectrum -> n real samples, inverse scaled by 1/n (include/fft_auto.h:98-106)."""
    X = np.asarray(X, dtype=np.complex128)
    full = np.empty(X.shape[:-1] + (n,), dtype=np.complex128)
    hb = n // 2 + 1
    full[..., :hb] = X
    k = np.arange(hb, n)
    full[..., hb:] = np.conj(X[..., n - k])
    return oracle_fft(full, FFT_INVERSE, _algo_for(n)).real


def oracle_next_pow2(v):
    m = 1
    while m < v:
        m <<= 1
    return m


def oracle_conv_linear(x, h):
    """fft_convolution, reference applications/convolution.c:34-69: zero-pad both to next_power_of_two(nx + nh - 1),
    transform, multiply, inverse transform, keep the first nx + nh - 1."""
    x = np.asarray(x, dtype=np.complex128)
    h = np.asarray(h, dtype=np.complex128)
    nx, nh = x.shape[-1], h.shape[-1]
    ny = nx + nh - 1
    m = oracle_next_pow2(ny)
    xp = np.zeros(x.shape[:-1] + (m,), dtype=np.complex128)
    xp[..., :nx] = x
    hp = np.zeros(m, dtype=np.complex128)
    hp[:nh] = h
    X = oracle_fft(xp, FFT_FORWARD) * oracle_fft(hp, FFT_FORWARD)
    return oracle_fft(X, FFT_INVERSE)[..., :ny]


def oracle_conv_circular(x, h):
    """circular_convolution, reference applications/convolution.c:72-96 (n a power of two)."""
    X = oracle_fft(np.asarray(x, dtype=np.complex128), FFT_FORWARD) * oracle_fft(np.asarray(h, dtype=np.complex128), FFT_FORWARD)
    return oracle_fft(X, FFT_INVERSE)


def oracle_autocorr(x):
    """autocorrelation_fft, reference applications/power_spectrum.c:133-158: zero-pad to next_power_of_two(2 n)."""
    x = np.asarray(x, dtype=np.complex128)
    n = x.shape[-1]
    m = oracle_next_pow2(2 * n)
    xp = np.zeros(x.shape[:-1] + (m,), dtype=np.complex128)
    xp[..., :n] = x
    X = oracle_fft(xp, FFT_FORWARD)
    return oracle_fft(X * np.conj(X), FFT_INVERSE)[..., :n]


def oracle_xcorr(x, y):
    """cross_correlation_fft, reference applications/power_spectrum.c:161-190: IFFT(conj(X) Y), first n."""
    x = np.asarray(x, dtype=np.complex128)
    y = np.asarray(y, dtype=np.complex128)
    n = x.shape[-1]
    m = oracle_next_pow2(2 * n)
    xp = np.zeros(x.shape[:-1] + (m,), dtype=np.complex128)
    yp = np.zeros_like(xp)
    xp[..., :n] = x
    yp[..., :n] = y
    return oracle_fft(np.conj(oracle_fft(xp, FFT_FORWARD)) * oracle_fft(yp, FFT_FORWARD), FFT_INVERSE)[..., :n]


def oracle_periodogram(x, sample_rate=1.0):
    """compute_periodogram, reference applications/power_spectrum.c:58-86: Hann window 0.5 (1 - cos(2 pi i / (n - 1)))
    (:5-10), transform, |X[k]|^2 / (sample_rate * 0.375 n), doubled for 0 < k < n/2."""
    x = np.asarray(x, dtype=np.complex128)
    n = x.shape[-1]
    w = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(n) / (n - 1))) if n > 1 else np.ones(1)
    X = oracle_fft(x * w, FFT_FORWARD, _algo_for(n))
    psd = np.abs(X[..., : n // 2 + 1]) ** 2 / (sample_rate * 0.375 * n)
    psd[..., 1:n // 2] *= 2.0
    return psd
