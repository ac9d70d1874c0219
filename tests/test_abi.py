"""The C-ABI library loads and exports every symbol include/*.h declares; without a GPU every entry point
fails loudly (NULL / -1 / "No GPU") instead of computing anything on the CPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("fft_gpu.h", "fft_hip.h", "fft_auto.h", "fft_algorithms.h", "fft_apps.h", "fft_utils.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b((?:fft|radix|split|bluestein|circular|compute_periodogram|autocorrelation|cross_correlation|save_complex|load_complex)[a-z0-9_]*)\s*\(", text):
            names.add(m.group(1))
    return names


def test_every_declared_symbol_is_exported_and_bound():
    import fftlib
    lib = fftlib.load()
    declared = declared_symbols()
    assert len(declared) >= 80
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(fftlib.SIGNATURES), declared ^ set(fftlib.SIGNATURES)


def test_enum_values_match_the_reference_headers():
    h = open(os.path.join(ROOT, "include", "fft_gpu.h")).read()
    for name, val in (("FFT_GPU_NONE", 0), ("FFT_GPU_CUDA", 1), ("FFT_GPU_METAL", 2), ("FFT_GPU_OPENCL", 3),
                      ("FFT_GPU_HIP", 4), ("FFT_GPU_AUTO", -1)):
        assert re.search(r"%s\s*=\s*%d\b" % (name, val), h), name
    c = open(os.path.join(ROOT, "include", "fft_common.h")).read()
    assert re.search(r"FFT_FORWARD\s*=\s*-1", c) and re.search(r"FFT_INVERSE\s*=\s*1", c)
    a = open(os.path.join(ROOT, "include", "fft_auto.h")).read()
    assert re.search(r"FFT_PREFER_GPU\s*=\s*1\s*<<\s*9", a) and re.search(r"FFT_HW_GPU_HIP\s*=\s*1\s*<<\s*8", a)


def _no_gpu():
    import fftlib
    return fftlib.load().fft_gpu_available() == 0


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present: the no-device behaviour cannot be observed")
def test_no_device_means_loud_failure_not_cpu_fallback(capfd):
    import fftlib
    lib = fftlib.load()
    assert lib.fft_gpu_init(fftlib.FFT_GPU_AUTO) == -1
    assert lib.fft_gpu_get_backend() == 0
    assert lib.fft_gpu_get_device_name() == b"No GPU"
    assert lib.fft_gpu_alloc(16) is None and lib.fft_gpu_plan_1d(16, 1, -1) is None
    x = np.ones(16, dtype=np.complex128)
    y = np.full(16, 7.0 + 0j)
    assert lib.fft_auto(x.ctypes.data, y.ctypes.data, 16, -1) == -1
    assert lib.fft_gpu_dft_1d(x.ctypes.data, y.ctypes.data, 16, -1) == -1
    assert lib.radix4_fft_gpu(x.ctypes.data, 16, -1) == -1
    assert lib.fft_plan_dft_1d(16, x.ctypes.data, y.ctypes.data, -1, 0) is None
    assert np.all(y == 7.0) and np.all(x == 1.0)  # nothing was computed anywhere
    assert "no" in capfd.readouterr().err.lower()  # and it said so on stderr
    with pytest.raises(RuntimeError):
        fftlib.init()
    # host-only helpers still behave
    assert lib.fft_version().startswith(b"2.")
    p = lib.fft_alloc_complex(100)
    assert p and p % 64 == 0
    lib.fft_free(p)
    assert lib.fft_plan_r2c_1d(8, None, None, 0) is None and lib.fft_gpu_plan_2d(4, 4, -1) is None


def test_host_planner_logic_matches_reference_constants():
    """Host-side helpers of include/fft_common.h agree with the oracle (and so with the reference)."""
    import ctypes as C
    import subprocess
    import tempfile
    import oracle_lib as O
    src = r'''
    #include "fft_common.h"
    unsigned t_bitrev(unsigned x, int l) { return bit_reverse(x, l); }
    int t_np2(int n) { return next_power_of_two(n); }
    int t_log2(int n) { return log2_int(n); }
    int t_ispow2(int n) { return is_power_of_two(n); }
    void t_tw(int k, int n, int d, double* re, double* im) { complex_t w = twiddle_factor(k, n, (fft_direction)d); *re = creal(w); *im = cimag(w); }
    '''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        so = os.path.join(td, "t.so")
        subprocess.run(["gcc", "-std=c99", "-O1", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), c, "-o", so, "-lm"], check=True)
        lib = C.CDLL(so)
        lib.t_bitrev.restype = C.c_uint
        lib.t_bitrev.argtypes = [C.c_uint, C.c_int]
        for log2n in range(0, 14):
            t = O.bit_reverse_table(log2n) if log2n else np.zeros(1, np.uint32)
            got = np.array([lib.t_bitrev(i, log2n) for i in range(1 << log2n)], dtype=np.uint32)
            assert np.array_equal(got, t), log2n
        for n, w in ((1, 1), (2, 2), (3, 4), (1000, 1024), (1024, 1024), (2000005, 2097152)):
            assert lib.t_np2(n) == w == O.oracle().oracle_next_power_of_two(n)
        assert lib.t_log2(1 << 20) == 20 and lib.t_ispow2(4096) == 1 and lib.t_ispow2(12) == 0 and lib.t_ispow2(0) == 0
        lib.t_tw.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        for k, n, d in ((0, 8, -1), (2, 8, -1), (2, 8, 1), (4, 8, -1), (6, 8, -1), (1, 1024, -1), (3, 64, 1)):
            a, b, a2, b2 = C.c_double(), C.c_double(), C.c_double(), C.c_double()
            lib.t_tw(k, n, d, C.byref(a), C.byref(b))
            O.oracle().oracle_twiddle_factor(k, n, d, C.byref(a2), C.byref(b2))
            assert abs(a.value - a2.value) < 1e-16 and abs(b.value - b2.value) < 1e-16
