"""GPU parity of the "next" rows of the scope table, all through the C ABI:
  f1  FFT_MEASURE really measures; fft_auto() keeps its plans (no hipMalloc / stream on a repeated call); pinned arrays
  f2  2D complex transforms, r2c / c2r 1D
  f3  fused consumers: convolution, correlations, periodogram; Bluestein with fused ends
against the oracle's restatement of the reference's applications (tests/oracle_lib.py, pinned to the real reference by
tests/test_oracle_apps.py) and against the golden vectors the compiled reference produced."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "reference_apps_vectors.npz"))
TOL = {np.dtype(np.complex64): 1e-4, np.dtype(np.complex128): 1e-6, np.dtype(np.float32): 1e-4, np.dtype(np.float64): 1e-6}      # north_star
TIGHT = {np.dtype(np.complex64): 3e-6, np.dtype(np.complex128): 1e-11, np.dtype(np.float32): 3e-6, np.dtype(np.float64): 1e-11}  # what it really achieves


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.complex128) - b) / np.linalg.norm(b))


def lcg(shape, seed, dtype):
    n = int(np.prod(shape))
    return O.gen_lcg(n, seed, 1).reshape(shape).astype(dtype)


# ------------------------------------------------------------------ f2: 2D
@pytest.mark.parametrize("rows,cols,nm,dtype", [
    (32, 64, 1, np.complex128), (128, 32, 2, np.complex64),
    (1024, 1024, 1, np.complex64),      # direct column pass, 8 MiB image
    (4096, 256, 1, np.complex64),       # the longest column that fits one LDS tile: narrow there -> two strided passes (64 x 64)
    (2048, 64, 2, np.complex128),       # two strided passes, two matrices
    (8192, 16, 1, np.complex64),        # columns longer than a tile: two strided passes (128 x 64)
    (4096, 24, 1, np.complex64),        # ... with a column count that is not a power of two (the rows go through Bluestein)
    (65536, 8, 1, np.complex64),        # 256 x 256 strided
    (2048, 3, 1, np.complex64),         # odd column count: no 16-byte lane access -> transpose path
    (17, 33, 3, np.complex128),         # nothing is a power of two: Bluestein both ways, odd pitch
    (60, 100, 2, np.complex64),
    (1, 4096, 2, np.complex64), (256, 2, 1, np.complex128), (2, 2, 5, np.complex64),
])
def test_fft2d_vs_oracle(gpu_lib, rows, cols, nm, dtype):
    import fftlib
    x = lcg((nm, rows, cols), rows * 31 + cols, dtype)
    for d in (-1, 1):
        y = fftlib.fft2d(x, d)
        r = rel(y, O.oracle_fft2d(x.astype(np.complex128), d))
        assert r <= TOL[np.dtype(dtype)] and r <= TIGHT[np.dtype(dtype)] * 4, (rows, cols, d, r)
    assert rel(fftlib.fft2d(fftlib.fft2d(x, -1), 1), x.astype(np.complex128)) <= TIGHT[np.dtype(dtype)] * 4  # scaled ONCE


@pytest.mark.parametrize("rows,cols", [(32, 64), (128, 32)])
def test_fft2d_vs_reference_golden_and_host_apis(gpu_lib, rows, cols):
    """fft_gpu_dft_2d and fft_plan_dft_2d / fft_execute on host arrays against what the reference's own fft_2d
    (applications/image_fft.c:35-72) produced; its inverse is ours / (rows * cols) (double scaling, documented)."""
    lib = gpu_lib
    x = np.ascontiguousarray(G["fft2d_%dx%d_in" % (rows, cols)])
    y = np.empty_like(x)
    assert lib.fft_gpu_dft_2d(x.ctypes.data, y.ctypes.data, rows, cols, -1) == 0
    assert rel(y, G["fft2d_%dx%d_fwd" % (rows, cols)]) < 1e-12
    plan = lib.fft_plan_dft_2d(rows, cols, x.ctypes.data, y.ctypes.data, +1, 0)
    assert plan
    lib.fft_execute(plan)
    lib.fft_destroy_plan(plan)
    assert rel(y / (rows * cols), G["fft2d_%dx%d_inv" % (rows, cols)]) < 1e-12
    z = x.copy()  # in place through the device-handle API
    mem = lib.fft_gpu_alloc(rows * cols)
    p2 = lib.fft_gpu_plan_2d(rows, cols, -1)
    assert mem and p2
    lib.fft_gpu_copy_h2d(mem, z.ctypes.data, rows * cols)
    lib.fft_gpu_execute(p2, mem, mem)
    lib.fft_gpu_copy_d2h(z.ctypes.data, mem, rows * cols)
    lib.fft_gpu_destroy_plan(p2)
    lib.fft_gpu_free(mem)
    assert rel(z, G["fft2d_%dx%d_fwd" % (rows, cols)]) < 1e-12


def test_fft2d_rejects_bad_arguments(gpu_lib):
    lib = gpu_lib
    assert lib.fft_gpu_plan_2d(0, 8, -1) is None and lib.fft_gpu_plan_2d(8, -1, -1) is None
    x = np.zeros(4, dtype=np.complex128)
    assert lib.fft_gpu_dft_2d(None, x.ctypes.data, 2, 2, -1) == -1
    assert lib.fft_plan_dft_2d(2, 2, None, x.ctypes.data, -1, 0) is None


# ------------------------------------------------------------------ f2: real transforms
@pytest.mark.parametrize("n,batch,dtype", [(1024, 3, np.float64), (1 << 16, 4, np.float32), (1 << 20, 2, np.float32),
                                           (1 << 18, 2, np.float64), (1000, 5, np.float64), (6, 2, np.float32),
                                           (2, 3, np.float64), (1, 2, np.float64), (1009, 3, np.float32), (99999, 1, np.float64)])
def test_r2c_c2r_vs_oracle(gpu_lib, n, batch, dtype):
    import fftlib
    x = lcg((batch, n), n + 17, np.complex128).real.astype(dtype)
    X = fftlib.rfft(x)
    ref = O.oracle_r2c(x.astype(np.float64))
    assert X.shape == (batch, n // 2 + 1)
    r = rel(X, ref)
    assert r <= TOL[np.dtype(dtype)] and r <= TIGHT[np.dtype(dtype)] * 4, (n, r)
    back = fftlib.irfft(X, n)
    assert np.linalg.norm(back - x) / np.linalg.norm(x) <= TIGHT[np.dtype(dtype)] * 8
    assert rel(fftlib.irfft(ref.astype(X.dtype), n), O.oracle_c2r(ref, n)) <= TIGHT[np.dtype(dtype)] * 8


def test_r2c_c2r_host_planner_api(gpu_lib):
    """fft_plan_r2c_1d / fft_plan_c2r_1d + fft_execute (reference fft_auto.h:88-106; NULL / use-after-free there)."""
    lib = gpu_lib
    n = 4096
    x = lcg((n,), 3, np.complex128).real.copy()
    X = np.zeros(n // 2 + 1, dtype=np.complex128)
    p = lib.fft_plan_r2c_1d(n, x.ctypes.data, X.ctypes.data, 0)
    assert p
    lib.fft_execute(p)
    lib.fft_destroy_plan(p)
    assert rel(X, O.oracle_r2c(x)) < 1e-12
    back = np.zeros(n)
    q = lib.fft_plan_c2r_1d(n, X.ctypes.data, back.ctypes.data, 0)
    assert q
    lib.fft_execute(q)
    lib.fft_destroy_plan(q)
    assert np.linalg.norm(back - x) / np.linalg.norm(x) < 1e-12
    assert lib.fft_plan_r2c_1d(0, x.ctypes.data, X.ctypes.data, 0) is None


def test_host_planner_in_place_r2c_and_plans_sharing_an_array(gpu_lib):
    """(1) an in-place r2c plan (in == out) writes n/2 + 1 complex values = 8 n + 16 bytes over an array it read 8 n
    bytes from: the page-locked range must cover the larger use; (2) a second, LARGER plan on an array that a smaller plan has
    already page-locked must still move its whole input and output (bounce buffer), never leave stale output; both report
    through fft_plan_last_error."""
    lib = gpu_lib
    n = 2048
    buf = np.zeros(n // 2 + 1, dtype=np.complex128)  # room for the in-place result
    x = lcg((n,), 5, np.complex128).real.copy()
    buf.view(np.float64)[:n] = x
    p = lib.fft_plan_r2c_1d(n, buf.ctypes.data, buf.ctypes.data, 0)
    assert p
    lib.fft_execute(p)
    assert lib.fft_plan_last_error(p) == 0
    lib.fft_destroy_plan(p)
    assert rel(buf, O.oracle_r2c(x)) < 1e-12
    # two plans of different length on one scratch array, the small one first (it page-locks the prefix)
    big = 8192
    a = lcg((big,), 7, np.complex128)
    y = np.zeros(big, dtype=np.complex128)
    small = lib.fft_plan_dft_1d(1024, a.ctypes.data, y.ctypes.data, -1, 0)
    large = lib.fft_plan_dft_1d(big, a.ctypes.data, y.ctypes.data, -1, 0)
    assert small and large
    lib.fft_execute(large)
    assert lib.fft_plan_last_error(large) == 0
    assert rel(y, O.oracle_fft(a[None, :], -1, "dit")[0]) < 1e-12
    lib.fft_execute(small)
    assert lib.fft_plan_last_error(small) == 0
    assert rel(y[:1024], O.oracle_fft(a[None, :1024], -1, "dit")[0]) < 1e-12
    lib.fft_destroy_plan(small)
    lib.fft_destroy_plan(large)


# ------------------------------------------------------------------ f3: fused consumers
@pytest.mark.parametrize("nx,nh", [(100, 17), (256, 256), (1, 1), (33, 5)])
def test_convolution_vs_reference_golden(gpu_lib, nx, nh):
    lib = gpu_lib
    x = np.ascontiguousarray(G["conv_%d_%d_x" % (nx, nh)])
    h = np.ascontiguousarray(G["conv_%d_%d_h" % (nx, nh)])
    y = np.zeros(nx + nh - 1, dtype=np.complex128)
    assert lib.fft_convolution_gpu(x.ctypes.data, nx, h.ctypes.data, nh, y.ctypes.data) == 0
    assert rel(y, G["conv_%d_%d_y" % (nx, nh)]) < 1e-12


@pytest.mark.parametrize("n", [64, 1024])
def test_circular_convolution_vs_reference_golden(gpu_lib, n):
    lib = gpu_lib
    x, h = (np.ascontiguousarray(G["circ_%d_%s" % (n, k)]) for k in "xh")
    y = np.zeros(n, dtype=np.complex128)
    assert lib.circular_convolution_gpu(x.ctypes.data, h.ctypes.data, n, y.ctypes.data) == 0
    assert rel(y, G["circ_%d_y" % n]) < 1e-12
    assert lib.circular_convolution_gpu(x.ctypes.data, h.ctypes.data, 48, y.ctypes.data) == -1  # not a power of two: refused, no exit()


@pytest.mark.parametrize("n", [100, 1000])
def test_correlations_and_periodogram_vs_reference_golden(gpu_lib, n):
    lib = gpu_lib
    x, y = (np.ascontiguousarray(G["corr_%d_%s" % (n, k)]) for k in "xy")
    for fn, args, key in (("autocorrelation_fft_gpu", (x,), "autocorrelation_fft_%d"), ("cross_correlation_fft_gpu", (x, y), "cross_correlation_fft_%d")):
        p = getattr(lib, fn)(*[a.ctypes.data for a in args], n)
        assert p
        got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(2 * n,)).view(np.complex128).copy()
        lib.fft_free(p)
        assert rel(got, G[key % n]) < 1e-12
    m = {100: 256, 1000: 4096}[n]
    sig = np.ascontiguousarray(G["psd_%d_x" % m])
    p = lib.compute_periodogram_gpu(sig.ctypes.data, m, 48000.0)
    assert p
    psd = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(m // 2 + 1,)).copy()
    lib.fft_free(p)
    assert np.linalg.norm(psd - G["psd_%d_out" % m]) / np.linalg.norm(G["psd_%d_out" % m]) < 1e-12
    assert lib.compute_periodogram_gpu(sig.ctypes.data, 100, 1.0) is None


@pytest.mark.parametrize("kind,nx,nh,batch,dtype", [
    ("conv", 1000, 25, 64, np.complex64),            # m = 1024: single-pass hooks
    ("conv", 60000, 5537, 8, np.complex64),          # m = 65536: two passes
    ("conv", 1 << 20, 1 << 20, 2, np.complex64),     # m = 2^21: the largest two-pass size class
    ("conv", 3000000, 1000, 1, np.complex64),        # m = 2^22: three passes
    ("conv", 50000, 5000, 4, np.complex128),
    ("circ", 1 << 16, 0, 8, np.complex64), ("circ", 4096, 0, 16, np.complex128),
    ("autocorr", 20000, 0, 6, np.complex64), ("autocorr", 3333, 0, 6, np.complex128),
    ("xcorr", 20001, 0, 6, np.complex64), ("xcorr", 70000, 0, 3, np.complex128),
    ("psd", 1 << 16, 0, 8, np.complex64), ("psd", 4096, 0, 8, np.complex128),
    # transforms of 2^20 / 2^21 points: forward-last + inverse-first pass chained (csrc/fft_kernels_chain.h), two- and three-pass
    ("conv", 1 << 19, 1000, 3, np.complex64), ("circ", 1 << 20, 0, 2, np.complex64), ("circ", 1 << 21, 0, 2, np.complex128),
    ("autocorr", 1 << 19, 0, 2, np.complex128), ("xcorr", 1 << 19, 0, 2, np.complex64),
])
def test_fused_consumers_batched_vs_oracle(gpu_lib, kind, nx, nh, batch, dtype):
    """Device-resident, batched plans (fft_gpu_plan_fused_hip): fused and with the element-wise steps as kernels of
    their own (FFT_GPU_OPT_NO_FUSION), against the oracle's restatement on a few transforms of the batch."""
    import fftlib
    x = lcg((batch, nx), nx, dtype)
    y = lcg((batch, nx), nx + 1, dtype) if kind == "xcorr" else None
    h = lcg((nh if kind == "conv" else nx,), nh + 9, dtype) if kind in ("conv", "circ") else None
    plan = fftlib.ExtPlan.fused(kind, nx, batch, h, dtype)
    rdt = dtype if kind != "psd" else (np.float32 if dtype == np.complex64 else np.float64)
    bufs = [fftlib.DeviceBuffer(x.nbytes)]
    bufs[0].upload(x)
    if y is not None:
        bufs.append(fftlib.DeviceBuffer(y.nbytes))
        bufs[1].upload(y)
    out = fftlib.DeviceBuffer(batch * plan.out_len * np.dtype(rdt).itemsize)
    results = []
    for no_fusion, no_chain in ((0, 0), (1, 0), (0, 1)):
        plan.set_option(fftlib.OPT_NO_FUSION, no_fusion)
        plan.set_option(fftlib.OPT_NO_CHAIN, no_chain)
        fused = plan.info().fused  # 2: the middle two passes as one kernel (transforms of >= 2^19 points whose end tiles agree)
        assert fused == 0 if no_fusion else (fused == 1 if no_chain else fused in (1, 2, 3)), (kind, nx, fused)
        plan.execute_fused(bufs[0].ptr, bufs[1].ptr if y is not None else None, out.ptr, 48000.0)
        assert plan.sync() == 0
        results.append(out.download((batch, plan.out_len), rdt))
    pick = sorted({0, batch // 2, batch - 1})
    x64 = x[pick].astype(np.complex128)
    ref = {"conv": lambda: O.oracle_conv_linear(x64, h.astype(np.complex128)),
           "circ": lambda: O.oracle_conv_circular(x64, h.astype(np.complex128)),
           "autocorr": lambda: O.oracle_autocorr(x64),
           "xcorr": lambda: O.oracle_xcorr(x64, y[pick].astype(np.complex128)),
           "psd": lambda: O.oracle_periodogram(x64, 48000.0)}[kind]()
    for res in results:
        r = rel(res[pick], ref)
        assert r <= TOL[np.dtype(dtype)] and r <= TIGHT[np.dtype(dtype)] * 16, (kind, nx, r)
    for other in results[1:]:
        assert rel(results[0], other.astype(np.complex128)) <= TIGHT[np.dtype(dtype)] * 16  # every transform of the batch
    plan.destroy()
    for b in bufs + [out]:
        b.free()


@pytest.mark.parametrize("n,batch,dtype", [(1009, 7, np.complex128), (1009, 7, np.complex64), (100003, 3, np.complex128),
                                           (100003, 3, np.complex64), (30011, 5, np.complex64), (30011, 5, np.complex128),
                                           (1000003, 2, np.complex128), (3000017, 1, np.complex128), (1000, 33, np.complex64)])
def test_bluestein_fused_equals_unfused_and_oracle(gpu_lib, n, batch, dtype):
    import fftlib
    x = lcg((batch, n), n, dtype)
    buf = fftlib.DeviceBuffer(x.nbytes)
    for d in (-1, 1):
        plan = fftlib.Plan(n, batch, d, dtype)
        res = []
        for no_fusion, no_chain in ((0, 0), (1, 0), (0, 1)):
            plan.set_option(fftlib.OPT_NO_FUSION, no_fusion)
            plan.set_option(fftlib.OPT_NO_CHAIN, no_chain)
            fused = plan.info().fused
            # m = 2^16 = 256 x 256, 2^18 = 512 x 512, 2^21 = 128^3, 2^23 = 256 x 128 x 256: the planner picks splits whose end tiles agree, those chain;
            # m = 2048 is a single pass
            assert fused == (0 if no_fusion else 1 if no_chain else 3 if n < 3000 else 2), (n, fused)  # 3: m fits one tile, ONE kernel; 2: every other m here chains (equal ends, or the mirrored split)
            buf.upload(x)
            plan.execute_ptr(buf.ptr, buf.ptr)  # in place: the user's array is both the first load and the last store
            assert plan.sync() == 0
            res.append(buf.download(x.shape, dtype))
        ref = O.oracle_fft(x[:1].astype(np.complex128), d, "bluestein")
        r = rel(res[0][:1], ref)
        # the reference's own error grows ~ n eps (twiddle recurrence, SURVEY.md fact 8: 2e-10 at n = 10^6): the oracle
        # restates it, so the fp64 bound against it is loosened with n; fused vs unfused below stays tight
        tight = max(TIGHT[np.dtype(dtype)] * 16, 1e-15 * n)
        assert r <= TOL[np.dtype(dtype)] and r <= tight, (n, d, r)
        for other in res[1:]:
            assert rel(res[0], other.astype(np.complex128)) <= TIGHT[np.dtype(dtype)] * 16
        plan.destroy()
    buf.free()


# ------------------------------------------------------------------ f1: planner
def test_fft_measure_times_the_candidates(gpu_lib):
    lib = gpu_lib
    FFT_MEASURE = 1
    for n in (1024, 1 << 16, 1000):
        x = lcg((n,), n, np.complex128)
        y = np.empty_like(x)
        plan = lib.fft_plan_dft_1d(n, x.ctypes.data, y.ctypes.data, -1, FFT_MEASURE)
        assert plan
        algo = lib.fft_plan_measured_algo(plan)
        assert algo in ((0, 1, 2) if (n & (n - 1)) == 0 else (0,)), algo
        lib.fft_execute(plan)
        lib.fft_destroy_plan(plan)
        assert rel(y, O.oracle_fft(x, -1, "dit" if (n & (n - 1)) == 0 else "bluestein")) < 1e-11
    est = lib.fft_plan_dft_1d(64, x.ctypes.data, y.ctypes.data, -1, 0)
    assert lib.fft_plan_measured_algo(est) == -1
    lib.fft_destroy_plan(est)


def test_device_level_measure_picks_a_schedule(gpu_lib):
    """fft_gpu_plan_measure_hip: team kernel vs multi-pass timed for THIS plan (size and batch) instead of the static
    table; whatever it keeps must compute the same spectrum."""
    import fftlib
    fftlib.set_policy(team=2)
    for n, batch in ((1 << 18, 16), (1 << 20, 256)):
        x = lcg((batch, n), 5, np.complex64)
        plan = fftlib.Plan(n, batch, -1, np.complex64)
        assert plan.info().team_tiles == 4
        kept = plan.lib.fft_gpu_plan_measure_hip(plan.handle, 3)
        assert kept in (0, 1)
        buf = fftlib.DeviceBuffer(x.nbytes)
        buf.upload(x)
        plan.execute_ptr(buf.ptr, buf.ptr)
        st = plan.team_status()
        assert st == (0 if kept == 1 else st)
        assert rel(buf.download(x.shape, np.complex64)[:1], O.oracle_fft(x[:1].astype(np.complex128), -1, "dit")) < 3e-6
        plan.destroy()
        buf.free()
    p2 = fftlib.Plan(4096, 8, -1, np.complex64)  # single pass: nothing to choose
    assert p2.lib.fft_gpu_plan_measure_hip(p2.handle, 2) == 0
    p2.destroy()


def test_fft_auto_keeps_its_plans(gpu_lib):
    """A second fft_auto() of the same (n, direction) allocates nothing and creates no stream."""
    lib = gpu_lib
    a, s = C.c_longlong(), C.c_longlong()
    n = 1 << 15
    x = lcg((n,), 1, np.complex128)
    y = np.empty_like(x)
    assert lib.fft_auto(x.ctypes.data, y.ctypes.data, n, -1) == 0
    lib.fft_gpu_debug_counters_hip(C.byref(a), C.byref(s))
    a0, s0 = a.value, s.value
    for _ in range(5):
        assert lib.fft_auto(x.ctypes.data, y.ctypes.data, n, -1) == 0
    lib.fft_gpu_debug_counters_hip(C.byref(a), C.byref(s))
    assert (a.value, s.value) == (a0, s0), "repeated fft_auto calls must reuse plan, stream and device buffer"
    assert rel(y, O.oracle_fft(x, -1, "dit")) < 1e-11
    assert lib.fft_auto(y.ctypes.data, y.ctypes.data, n, +1) == 0  # the inverse is a different plan: allocates once
    assert rel(y, x) < 1e-11
    for k in range(12):  # more sizes than slots: the oldest plans are replaced, results stay right
        m = 64 << (k % 6)
        u = lcg((m,), k, np.complex128)
        v = np.empty_like(u)
        assert lib.fft_auto(u.ctypes.data, v.ctypes.data, m, -1) == 0
        assert rel(v, O.oracle_fft(u, -1, "dit")) < 1e-11
    lib.fft_auto_cleanup()
    assert lib.fft_auto(x.ctypes.data, y.ctypes.data, n, -1) == 0


def test_planner_pins_the_borrowed_arrays(gpu_lib):
    lib = gpu_lib
    n = 1 << 16
    x = lcg((n,), 2, np.complex128)
    y = np.empty_like(x)
    assert lib.fft_gpu_host_is_registered_hip(x.ctypes.data) == 0
    plan = lib.fft_plan_dft_1d(n, x.ctypes.data, y.ctypes.data, -1, 0)
    assert plan
    assert lib.fft_gpu_host_is_registered_hip(x.ctypes.data) == 1 and lib.fft_gpu_host_is_registered_hip(y.ctypes.data) == 1
    lib.fft_execute(plan)
    lib.fft_destroy_plan(plan)
    assert rel(y, O.oracle_fft(x, -1, "dit")) < 1e-11
    assert lib.fft_gpu_host_is_registered_hip(x.ctypes.data) == 0 and lib.fft_gpu_host_is_registered_hip(y.ctypes.data) == 0
    FFT_CONSERVE_MEMORY = 1 << 8
    plan = lib.fft_plan_dft_1d(n, x.ctypes.data, y.ctypes.data, -1, FFT_CONSERVE_MEMORY)
    assert plan and lib.fft_gpu_host_is_registered_hip(x.ctypes.data) == 0
    lib.fft_destroy_plan(plan)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nx,batch,dtype", [("xcorr", 1 << 17, 11, np.complex64), ("circ", 1 << 18, 7, np.complex128),
                                                 ("conv", 60000, 9, np.complex64),
                                                 ("circ", 1 << 17, 9, np.complex64), ("xcorr", 1 << 18, 5, np.complex128)])  # odd powers: mirrored split
def test_chained_fused_plans_over_several_launch_groups(gpu_lib, kind, nx, batch, dtype):
    """The chained middle pass (csrc/fft_kernels_chain.h) with the batch cut into launch groups of 2-4 transforms
    (chunk_mb policy): per-transform spectral tables (cross-correlation) must follow the group offset; against the oracle
    and against the same plan with the two kernels."""
    import fftlib
    fftlib.set_policy(team=1, min_batch=0, chunk_mb=4)  # m = 2^18 fp32 (2 MiB): groups of 2; 2^18 fp64 (4 MiB): groups of 1; 2^16: 8
    x = lcg((batch, nx), nx, dtype)
    y = lcg((batch, nx), nx + 1, dtype) if kind == "xcorr" else None
    nh = 5000 if kind == "conv" else nx
    h = lcg((nh,), nh + 9, dtype) if kind in ("conv", "circ") else None
    plan = fftlib.ExtPlan.fused(kind, nx, batch, h, dtype)
    info = plan.info()
    assert info.fused == 2 and info.chunk_batch < batch, (info.fused, info.chunk_batch)
    bufs = [fftlib.DeviceBuffer(x.nbytes)]
    bufs[0].upload(x)
    if y is not None:
        bufs.append(fftlib.DeviceBuffer(y.nbytes))
        bufs[1].upload(y)
    out = fftlib.DeviceBuffer(batch * plan.out_len * np.dtype(dtype).itemsize)
    res = []
    for no_chain in (0, 1):
        plan.set_option(fftlib.OPT_NO_CHAIN, no_chain)
        plan.execute_fused(bufs[0].ptr, bufs[1].ptr if y is not None else None, out.ptr, 1.0)
        assert plan.sync() == 0
        res.append(out.download((batch, plan.out_len), dtype))
    x64 = x.astype(np.complex128)
    ref = {"conv": lambda: O.oracle_conv_linear(x64, h.astype(np.complex128)),
           "circ": lambda: O.oracle_conv_circular(x64, h.astype(np.complex128)),
           "xcorr": lambda: O.oracle_xcorr(x64, y.astype(np.complex128))}[kind]()
    r = rel(res[0], ref)
    assert r <= TOL[np.dtype(dtype)] and r <= TIGHT[np.dtype(dtype)] * 16, (kind, nx, r)
    assert rel(res[0], res[1].astype(np.complex128)) <= TIGHT[np.dtype(dtype)] * 16
    plan.destroy()
    for b in bufs + [out]:
        b.free()


@pytest.mark.gpu
def test_random_sizes_roundtrip_parseval_and_numpy(gpu_lib):
    """Size-independent properties over a seeded mix of lengths (powers of two of every pass count, Bluestein lengths whose
    padded transform is single-pass, two-pass chained / unchained, three-pass) and batches: IFFT(FFT(x)) = x, Parseval, and the
    first transform against numpy's pocketfft in double precision."""
    import fftlib
    rng = np.random.default_rng(20260101)
    sizes = [1 << k for k in (6, 11, 13, 14, 15, 16, 17, 19, 21)] + [int(v) for v in rng.integers(1000, 300000, 14)] + [999983, 1048573, 2 ** 20 + 1]
    for n in sizes:
        for dtype in (np.complex64, np.complex128):
            batch = int(max(1, min(64, (1 << 22) // n)))
            batch = int(rng.integers(1, batch + 1))
            x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(dtype)
            a, b = fftlib.DeviceBuffer(x.nbytes), fftlib.DeviceBuffer(x.nbytes)
            a.upload(x)
            fwd, inv = fftlib.Plan(n, batch, -1, dtype), fftlib.Plan(n, batch, 1, dtype)
            fwd.execute_ptr(a.ptr, b.ptr)
            assert fwd.sync() == 0
            X = b.download(x.shape, dtype)
            inv.execute_ptr(b.ptr, b.ptr)
            assert inv.sync() == 0
            back = b.download(x.shape, dtype)
            tol = TOL[np.dtype(dtype)]
            assert rel(back, x.astype(np.complex128)) < tol, (n, batch, dtype)
            e_t = np.sum(np.abs(x.astype(np.complex128)) ** 2, axis=1)
            e_f = np.sum(np.abs(X.astype(np.complex128)) ** 2, axis=1) / n
            assert np.max(np.abs(e_f - e_t) / e_t) < tol * 4, (n, batch, dtype)
            assert rel(X[:1], np.fft.fft(x[:1].astype(np.complex128), axis=1)) < tol, (n, batch, dtype)
            fwd.destroy(); inv.destroy(); a.free(); b.free()


@pytest.mark.gpu
def test_host_batch_api_pipelines_page_locked_arrays(gpu_lib):
    """fft_gpu_dft_1d_batch (reference gpu/fft_gpu.c:344-375) with page-locked host arrays of >= 1 GiB: the batch goes through
    the device in 128 MiB groups on three streams (copy in / transform / copy out overlap); same results as with pageable
    arrays, in place too, with a last group that is not full."""
    import time
    lib = gpu_lib
    n, batch = 4096, 8 * 2048 + 77  # eight groups of 2048 transforms + a short one
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex128)
    pick = [0, 2047, 2048, 3 * 2048 + 5, batch - 1]
    ref = np.fft.fft(x[pick], axis=1)
    out_pageable = np.empty_like(x)
    t0 = time.perf_counter()
    assert lib.fft_gpu_dft_1d_batch(x.ctypes.data, out_pageable.ctypes.data, n, batch, -1) == 0
    t_pageable = time.perf_counter() - t0
    assert rel(out_pageable[pick], ref) < 1e-12
    xp, out_pinned = x.copy(), np.empty_like(x)
    assert lib.fft_gpu_host_register_hip(xp.ctypes.data, xp.nbytes) == 0 and lib.fft_gpu_host_register_hip(out_pinned.ctypes.data, out_pinned.nbytes) == 0
    try:
        t0 = time.perf_counter()
        assert lib.fft_gpu_dft_1d_batch(xp.ctypes.data, out_pinned.ctypes.data, n, batch, -1) == 0
        t_pinned = time.perf_counter() - t0
        assert np.array_equal(out_pinned, out_pageable)  # the same single-pass kernel on the same data, group by group
        assert lib.fft_gpu_dft_1d_batch(xp.ctypes.data, xp.ctypes.data, n, batch, -1) == 0  # in place
        assert np.array_equal(xp, out_pageable)
    finally:
        lib.fft_gpu_host_unregister_hip(xp.ctypes.data)
        lib.fft_gpu_host_unregister_hip(out_pinned.ctypes.data)
    print("host batch API, %d MiB each way: pageable %.3f s, page-locked + pipelined %.3f s" % (x.nbytes >> 20, t_pageable, t_pinned))
