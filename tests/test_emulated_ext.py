"""CPU emulation (tests/emu) of the plans built on the batched engine for the "next" rows of the scope table
(SURVEY.md 8f): 2D complex transforms, r2c / c2r, fused convolution / correlation / periodogram -- the unmodified
planner and kernel source against the oracle's restatement of the reference's applications."""
import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.complex128) - b) / np.linalg.norm(b))


TOL = {np.dtype(np.complex64): 3e-5, np.dtype(np.complex128): 1e-12, np.dtype(np.float32): 3e-5, np.dtype(np.float64): 1e-12}


def lcg(shape, seed, dtype):
    n = int(np.prod(shape))
    return O.gen_lcg(n, seed, 1).reshape(shape).astype(dtype)


@pytest.mark.parametrize("rows,cols,nm,dtype,lds,path", [
    (32, 64, 2, np.complex64, 0, 1),      # direct column pass
    (64, 32, 1, np.complex128, 0, 1),
    (16, 24, 3, np.complex64, 0, 1),      # cols not a power of two (rows: Bluestein over cols), still a multiple of the lane access
    (8, 5, 2, np.complex64, 0, 2),        # odd cols: 16-byte accesses impossible -> transpose path
    (12, 32, 2, np.complex128, 0, 2),     # rows not a power of two -> transpose + Bluestein
    (256, 8, 1, np.complex64, 1024, 3),   # rows do not fit one LDS tile of this budget -> two strided passes (16 x 16)
    (64, 32, 2, np.complex128, 4096, 3),  # the one-pass tile would be narrow (< 64-byte segments) -> two strided passes (8 x 8)
    (512, 8, 1, np.complex64, 2048, 3),   # 32 x 16: unequal factors, the twiddle index is the row offset
    (64, 100, 2, np.complex64, 4096, 3),  # column count not a power of two: the last column tile is partly padding
    (128, 6, 1, np.complex64, 1024, 3),
    (64, 5, 1, np.complex64, 0, 2),       # odd column count -> transpose path
    (1, 64, 2, np.complex64, 0, 0),       # a single row: rows only
])
def test_fft2d_emulated(rows, cols, nm, dtype, lds, path):
    x = lcg((nm, rows, cols), rows * 131 + cols, dtype)
    for d in (-1, 1):
        for inplace in (False, True):
            y, info = E.emu_fft2d(x, d, lds_budget=lds, inplace=inplace)
            assert info[0] == path, info[0]
            assert rel(y, O.oracle_fft2d(x.astype(np.complex128), d)) < TOL[np.dtype(dtype)], (d, inplace)
    # round trip: the inverse is scaled once
    y, _ = E.emu_fft2d(x, -1, lds_budget=lds)
    z, _ = E.emu_fft2d(y, 1, lds_budget=lds)
    assert rel(z, x.astype(np.complex128)) < TOL[np.dtype(dtype)]


@pytest.mark.parametrize("n", [2, 4, 8, 64, 1024, 6, 100, 1, 3, 9, 31, 101])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_r2c_c2r_emulated(n, dtype):
    batch = 3
    x = lcg((batch, n), n + 5, np.complex128).real.astype(dtype)
    X = E.emu_r2c(x)
    ref = O.oracle_r2c(x.astype(np.float64))
    assert X.shape == (batch, n // 2 + 1)
    assert rel(X, ref) < TOL[np.dtype(dtype)], n
    back = E.emu_c2r(X, n)
    assert np.linalg.norm(back - x) / np.linalg.norm(x) < TOL[np.dtype(dtype)] * 2, n
    assert rel(E.emu_c2r(ref.astype(X.dtype), n), O.oracle_c2r(ref, n)) < TOL[np.dtype(dtype)] * 2


@pytest.mark.parametrize("nx,nh,dtype,lds", [(100, 17, np.complex128, 0), (100, 17, np.complex64, 0), (101, 8, np.complex64, 0),
                                             (700, 401, np.complex128, 8192), (1, 1, np.complex128, 0), (5, 1, np.complex64, 0),
                                             (1500, 600, np.complex128, 40000), (2000, 148, np.complex64, 4096)])
def test_fused_linear_convolution_emulated(nx, nh, dtype, lds):
    x = lcg((4, nx), nx, dtype)
    h = lcg((nh,), nh + 3, dtype)
    ref = O.oracle_conv_linear(x.astype(np.complex128), h.astype(np.complex128))
    for no_fusion in (False, True):
        y, info = E.emu_fused("conv", x, h=h, lds_budget=lds, no_fusion=no_fusion)
        # info[1]: 1 fused ends, 2 also forward-last + inverse-first pass as one kernel (the last two cases: 64 x 64, 16 x 16 x 16)
        # ... 3: the padded transform is a single pass: FFT -> product -> inverse FFT as ONE kernel (TileHooks::mid_tab)
        # (two-pass plans with unequal factors chain through the mirrored split where the tiles agree: 1 or 2 there)
        want = (0,) if (no_fusion or info[2] == 0) else (2,) if nx >= 1500 else (3,) if info[0] == 1 else (1, 2)
        assert info[1] in want, info
        assert rel(y, ref) < TOL[np.dtype(dtype)] * 4, (nx, nh, no_fusion)
    # and against the defining sum (reference direct_convolution, convolution.c:19-31)
    direct = np.stack([np.convolve(r, h.astype(np.complex128)) for r in x.astype(np.complex128)])
    assert rel(ref, direct) < 1e-11


CHAINED_CASES = {("xcorr", 1500, 40000), ("autocorr", 1500, 4096), ("circ", 4096, 40000)}


@pytest.mark.parametrize("kind,n,dtype,lds", [("circ", 64, np.complex128, 0), ("circ", 2048, np.complex64, 4096),
                                              ("autocorr", 100, np.complex128, 0), ("autocorr", 1000, np.complex64, 4096),
                                              ("xcorr", 100, np.complex128, 0), ("xcorr", 333, np.complex64, 0),
                                              ("xcorr", 1500, np.complex128, 4096), ("xcorr", 1500, np.complex128, 40000),
                                              ("autocorr", 1500, np.complex64, 4096), ("circ", 4096, np.complex128, 40000)])
def test_fused_correlations_emulated(kind, n, dtype, lds):
    """info[1] == 2: the middle of FFT -> product -> IFFT ran as ONE kernel (fft_kernels_chain.h), see CHAINED_CASES"""
    x = lcg((3, n), n, dtype)
    y = lcg((3, n), n + 1, dtype)
    h = lcg((n,), n + 2, dtype)
    x64, y64, h64 = (a.astype(np.complex128) for a in (x, y, h))
    ref = {"circ": lambda: O.oracle_conv_circular(x64, h64), "autocorr": lambda: O.oracle_autocorr(x64),
           "xcorr": lambda: O.oracle_xcorr(x64, y64)}[kind]()
    chained = False
    for no_fusion in (False, True):
        out, info = E.emu_fused(kind, x, y=y if kind == "xcorr" else None, h=h if kind == "circ" else None, lds_budget=lds,
                                no_fusion=no_fusion)
        assert info[1] in ((0,) if no_fusion else (1, 2, 3))
        if not no_fusion and info[0] == 1:
            assert info[1] == (1 if kind == "xcorr" else 3), (kind, info)  # per-transform products keep two kernels
        chained = chained or info[1] == 2
        assert rel(out, ref) < TOL[np.dtype(dtype)] * 4, (kind, n, no_fusion)
    assert chained or (kind, n, lds) not in CHAINED_CASES, (kind, n, lds, info)


@pytest.mark.parametrize("n,dtype,lds", [(64, np.complex128, 0), (1024, np.complex64, 0), (4096, np.complex128, 4096)])
def test_fused_periodogram_emulated(n, dtype, lds):
    x = lcg((3, n), n, dtype)
    ref = O.oracle_periodogram(x.astype(np.complex128), 48000.0)
    for no_fusion in (False, True):
        psd, info = E.emu_fused("psd", x, lds_budget=lds, no_fusion=no_fusion, fs=48000.0)
        assert psd.shape == (3, n // 2 + 1)
        assert np.linalg.norm(psd - ref) / np.linalg.norm(ref) < TOL[np.dtype(dtype)] * 4


@pytest.mark.parametrize("kind", ["xcorr", "circ"])
def test_chained_pass_over_several_launch_groups(kind, monkeypatch):
    """Forward-last + inverse-first pass as ONE kernel (fft_kernels_chain.h) with the batch cut into launch groups
    (FFT_HIP_CHUNK_MB = 1: 16 transforms of m = 4096 fp64 per group, 40 = 16 + 16 + 8): the per-transform spectral table of the
    cross-correlation must follow the group offset."""
    monkeypatch.setenv("FFT_HIP_CHUNK_MB", "1")
    n, batch = (1500, 40) if kind == "xcorr" else (4096, 40)
    x = lcg((batch, n), n, np.complex128)
    y = lcg((batch, n), n + 1, np.complex128)
    h = lcg((n,), n + 2, np.complex128)
    out, info = E.emu_fused(kind, x, y=y if kind == "xcorr" else None, h=h if kind == "circ" else None, lds_budget=40000)
    assert info[1] == 2 and info[3] == 16, info  # chained, launch groups of 16
    ref = O.oracle_xcorr(x, y) if kind == "xcorr" else O.oracle_conv_circular(x, h)
    assert rel(out, ref) < TOL[np.dtype(np.complex128)] * 4
