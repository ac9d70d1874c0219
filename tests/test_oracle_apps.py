"""The oracle's restatements of the reference's consumers of the transform (tests/oracle_lib.py: 2D, convolution,
correlation, periodogram) pinned to golden vectors that tests/golden/make_golden_apps.py generated from the REAL
reference compiled here (applications/convolution.c, power_spectrum.c, image_fft.c, utils/fft_utils.c); plus the text
interchange format of include/fft_utils.h against a file the reference itself wrote.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "reference_apps_vectors.npz"))


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - b) / np.linalg.norm(b))


@pytest.mark.parametrize("nx,nh", [(100, 17), (256, 256), (1, 1), (33, 5)])
def test_linear_convolution_restatement(nx, nh):
    y = O.oracle_conv_linear(G["conv_%d_%d_x" % (nx, nh)], G["conv_%d_%d_h" % (nx, nh)])
    ref = G["conv_%d_%d_y" % (nx, nh)]
    # same transforms in the same order; the spectral product is numpy's here and C99's __muldc3 there (last-bit differences)
    assert rel(y, ref) < 1e-14


@pytest.mark.parametrize("n", [64, 1024])
def test_circular_convolution_restatement(n):
    assert rel(O.oracle_conv_circular(G["circ_%d_x" % n], G["circ_%d_h" % n]), G["circ_%d_y" % n]) < 1e-14


@pytest.mark.parametrize("n", [256, 4096])
def test_periodogram_restatement(n):
    assert rel(O.oracle_periodogram(G["psd_%d_x" % n], 48000.0), G["psd_%d_out" % n]) < 1e-14


@pytest.mark.parametrize("n", [100, 1000])
def test_correlation_restatements(n):
    assert rel(O.oracle_autocorr(G["corr_%d_x" % n]), G["autocorrelation_fft_%d" % n]) < 1e-14
    assert rel(O.oracle_xcorr(G["corr_%d_x" % n], G["corr_%d_y" % n]), G["cross_correlation_fft_%d" % n]) < 1e-14


@pytest.mark.parametrize("rows,cols", [(32, 64), (128, 32)])
def test_2d_restatement_and_the_reference_double_scaling(rows, cols):
    x = G["fft2d_%dx%d_in" % (rows, cols)]
    assert rel(O.oracle_fft2d(x, -1), G["fft2d_%dx%d_fwd" % (rows, cols)]) < 1e-15
    # the reference's inverse divides by rows * cols AGAIN after two scaled 1D inverses (image_fft.c:64-71): its result
    # is ours / (rows * cols) -- the defect this library documents and does not copy
    assert rel(O.oracle_fft2d(x, 1) / (rows * cols), G["fft2d_%dx%d_inv" % (rows, cols)]) < 1e-15


def _lib():
    import fftlib
    return fftlib.load()


def test_load_reads_the_file_the_reference_wrote():
    lib = _lib()
    data = C.c_void_p()
    n = C.c_int()
    path = os.path.join(ROOT, "tests", "golden", "reference_saved_array.txt")
    assert lib.load_complex_array(path.encode(), C.byref(data), C.byref(n)) == 0
    assert n.value == 12
    got = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_double)), shape=(24,)).view(np.complex128).copy()
    lib.fft_free(data)
    want = G["saved_array_values"]
    assert np.allclose(got, want, rtol=1e-6, atol=0)  # "%e" keeps 7 significant digits


def test_save_writes_the_reference_format_byte_for_byte(tmp_path):
    lib = _lib()
    x = np.ascontiguousarray(G["saved_array_values"])
    out = tmp_path / "ours.txt"
    assert lib.save_complex_array(str(out).encode(), x.ctypes.data, len(x)) == 0
    ref = open(os.path.join(ROOT, "tests", "golden", "reference_saved_array.txt"), "rb").read()
    assert out.read_bytes() == ref


def test_load_edge_cases(tmp_path):
    lib = _lib()
    data = C.c_void_p()
    n = C.c_int()
    assert lib.load_complex_array(str(tmp_path / "missing.txt").encode(), C.byref(data), C.byref(n)) == -1
    p = tmp_path / "nohdr.txt"
    p.write_text("0 1.5 -2.5 0 0\n1 3.0 4.0 5 0.9\n# a comment\n2 0.0 1.0 1 1.57\n")
    assert lib.load_complex_array(str(p).encode(), C.byref(data), C.byref(n)) == 0 and n.value == 3
    got = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_double)), shape=(6,)).copy()
    lib.fft_free(data)
    assert np.array_equal(got, [1.5, -2.5, 3.0, 4.0, 0.0, 1.0])
    e = tmp_path / "empty.txt"
    e.write_text("# FFT Data File\n")
    assert lib.load_complex_array(str(e).encode(), C.byref(data), C.byref(n)) == -1
