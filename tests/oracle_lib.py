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
