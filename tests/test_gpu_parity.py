"""GPU parity tests: the HIP engine, called through the C ABI, against the CPU
oracle (oracle/oracle_fft.c = restatement of the reference's radix-2 path,
pinned bit-exact to the real reference in tests/test_oracle.py) and
against the golden vectors generated from the real reference.

Tolerances (BASELINE.json north_star): rel-L2 <= 1e-6 for fp64, <= 1e-4 for fp32.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.complex128): 1e-6, np.dtype(np.complex64): 1e-4}
# what the engine actually achieves; a regression guard much tighter than the contractual gate
TIGHT = {np.dtype(np.complex128): 2e-11, np.dtype(np.complex64): 2e-6}


def rel(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def lcg(n, batch, dtype, seed=0):
    return O.gen_lcg(n, seed, batch).astype(dtype)


ALGOS = ["auto", "radix2", "radix4", "split_radix", "radix2_global", "radix2_shfl"]


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_small_pow2_vs_oracle(gpu_lib, dtype, algo, n):
    import fftlib
    x = lcg(n, 5, dtype, seed=n)
    for d in (-1, 1):
        y = fftlib.fft(x, d, fftlib.ALGO_NAMES[algo])
        # oracle = reference radix-2 DIT restatement; for n in {4, 8, 16} the reference itself is wrong
        # (bit_reverse bug) so the exact-DFT oracle is used there
        ref = O.oracle_fft(x.astype(np.complex128), d, "dit" if n >= 32 or n == 2 else "naive")
        r = rel(y, ref)
        assert r <= TOL[np.dtype(dtype)], (n, d, algo, r)
        assert r <= TIGHT[np.dtype(dtype)], (n, d, algo, r)


@pytest.mark.parametrize("log2n,dtype", [(13, np.complex64), (14, np.complex64), (13, np.complex128)])
def test_wide_row_kernel_vs_oracle(gpu_lib, log2n, dtype):
    """n = 8192 and 16384 fp32 and n = 8192 fp64 run ONE HBM round trip (csrc/fft_wide_row.h: 16 values per thread, radix-16 x 16 x 16
    x 2 / x 4, the 128 KiB images run in place) instead of the two-pass schedule -- VERDICT r2 item 2.  A batch that is ragged over the 256
    workgroups (some walk two rows, the next row's DMA is issued under the stages / the stores), both directions, in place and out of
    place, every transform against the oracle; and the plan says one pass."""
    import fftlib
    n = 1 << log2n
    batch = 256 + 37
    x = lcg(n, batch, dtype, seed=log2n)
    plan = fftlib.Plan(n, batch, -1, dtype)
    assert plan.info().n_passes == 1
    plan.destroy()
    for d, inplace in ((-1, False), (-1, True), (1, False), (1, True)):
        y = fftlib.fft(x, d, inplace=inplace)
        ref = O.oracle_fft(x.astype(np.complex128), d, "dit")
        for b in range(batch):
            r = rel(y[b:b + 1], ref[b:b + 1])
            assert r <= TOL[np.dtype(dtype)] and r <= TIGHT[np.dtype(dtype)], (log2n, d, inplace, b, r)


@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
@pytest.mark.parametrize("log2n", [13, 14, 15, 16, 17, 18, 20])
def test_multipass_vs_oracle(gpu_lib, dtype, log2n):
    import fftlib
    n = 1 << log2n
    batch = 3
    x = lcg(n, batch, dtype, seed=log2n)
    for d, inplace in ((-1, True), (1, False)):
        y = fftlib.fft(x, d, inplace=inplace)
        ref = O.oracle_fft(x.astype(np.complex128), d, "dit")
        r = rel(y, ref)
        assert r <= TOL[np.dtype(dtype)], (n, d, r)
        assert r <= TIGHT[np.dtype(dtype)], (n, d, r)
        exact = O.oracle_fft(x.astype(np.complex128), d, "exact")  # same schedule, exact twiddles: the truer DFT
        assert rel(y, exact) <= (1e-14 if dtype == np.complex128 else 2e-6), (n, d, rel(y, exact))


@pytest.mark.parametrize("algo", ["radix2", "radix4", "radix2_global"])
def test_multipass_other_families(gpu_lib, algo):
    import fftlib
    n = 1 << 16
    x = lcg(n, 2, np.complex64, seed=3)
    y = fftlib.fft(x, -1, fftlib.ALGO_NAMES[algo])
    ref = O.oracle_fft(x.astype(np.complex128), -1, "dit")
    assert rel(y, ref) <= 2e-6


def test_golden_n1024_all_reference_algorithms(gpu_lib, golden):
    """BASELINE config 1: N=1024 fp64 against the real reference's own outputs."""
    import fftlib
    x = golden["n1024_in"]
    for ref_algo in ("dit", "dif", "radix4", "split_radix"):
        for d, tag in ((-1, "fwd"), (1, "inv")):
            want = golden["n1024_%s_%s" % (ref_algo, tag)]
            for algo in ALGOS:
                y = fftlib.fft(x, d, fftlib.ALGO_NAMES[algo])
                assert rel(y, want) <= 1e-12, (ref_algo, tag, algo)


def test_golden_small_sizes(gpu_lib, golden):
    import fftlib
    for n in (32, 64, 128, 256, 512):
        x = golden["n%d_in" % n]
        assert rel(fftlib.fft(x, -1), golden["n%d_dit_fwd" % n]) <= 1e-13
        assert rel(fftlib.fft(x, 1), golden["n%d_dit_inv" % n]) <= 1e-13


def test_known_answer_n8(gpu_lib, golden):
    """The only known-answer vector in the reference tree (fft/fft.c:75)."""
    import fftlib
    y = fftlib.fft(golden["n8_known_in"], -1)
    assert np.max(np.abs(y - golden["n8_known_out_3dp"])) < 1e-3


@pytest.mark.parametrize("n", [65536, 1 << 18, 1 << 20])
def test_golden_large_sampled_bins(gpu_lib, golden, n):
    """Reference outputs at 256 sampled bins + norm, inputs regenerated from the closed forms."""
    import fftlib
    bins = golden["n%d_bins" % n]
    for name, gen in (("tone", lambda: O.gen_two_tone(n, 3, 1)), ("lcg", lambda: O.gen_lcg(n, 3, 1))):
        x = gen()
        assert np.array_equal(x[0, :16], golden["n%d_%s_in_head" % (n, name)])
        for dtype in (np.complex128, np.complex64):
            y = fftlib.fft(x.astype(dtype), -1)[0]
            want = golden["n%d_%s_fwd_bins" % (n, name)]
            scale = golden["n%d_%s_fwd_norm" % (n, name)] / np.sqrt(n)
            err = np.max(np.abs(y[bins] - want)) / scale
            assert err <= (1e-9 if dtype == np.complex128 else 3e-5), (n, name, dtype, err)
            assert abs(np.linalg.norm(y.astype(np.complex128)) / golden["n%d_%s_fwd_norm" % (n, name)] - 1) < TOL[np.dtype(dtype)]
            yi = fftlib.fft(x.astype(dtype), 1)[0]
            wanti = golden["n%d_%s_inv_bins" % (n, name)]
            erri = np.max(np.abs(yi[bins] - wanti)) / (scale / n)
            assert erri <= (1e-9 if dtype == np.complex128 else 3e-5), (n, name, dtype, erri)


@pytest.mark.parametrize("n", [31, 97, 1009])
def test_bluestein_golden(gpu_lib, golden, n):
    import fftlib
    x = golden["blu%d_in" % n]
    assert rel(fftlib.fft(x, -1), golden["blu%d_fwd" % n]) <= 1e-12
    assert rel(fftlib.fft(x, 1), golden["blu%d_inv" % n]) <= 1e-12
    assert rel(fftlib.fft(x.astype(np.complex64), -1), golden["blu%d_fwd" % n]) <= 1e-4


@pytest.mark.parametrize("n", [3, 5, 6, 12, 15, 20, 24, 30, 100, 1000, 4097, 65537])
def test_bluestein_any_n_vs_numpy(gpu_lib, n):
    """Composite sizes of the reference's test sweep (tests/test_all.c:415) and some primes."""
    import fftlib
    x = lcg(n, 3, np.complex128, seed=n)
    assert rel(fftlib.fft(x, -1), np.fft.fft(x, axis=-1)) <= 1e-12
    assert rel(fftlib.fft(x, 1), np.fft.ifft(x, axis=-1)) <= 1e-12


def test_bluestein_baseline_prime(gpu_lib, golden):
    """BASELINE config 5 shape: N=1000003 fp64 (batch reduced to 2 for the parity check)."""
    import fftlib
    n = 1000003
    x = O.gen_lcg(n, 5, 2)
    assert np.array_equal(x[0, :16], golden["blu1000003_in_head"])
    y = fftlib.fft(x, -1)
    bins = golden["blu1000003_bins"]
    scale = golden["blu1000003_fwd_norm"] / np.sqrt(n)
    assert np.max(np.abs(y[0][bins] - golden["blu1000003_fwd_bins"])) / scale <= 1e-6
    assert rel(y, np.fft.fft(x, axis=-1)) <= 1e-6
    back = fftlib.fft(y, 1)
    assert rel(back, x) <= 1e-6


def test_bit_reversal_kernel_index_exact(gpu_lib, golden):
    """BASELINE config 1's 'bit-exact index check': the permutation kernel against the
    reference's bit_reverse() table for log2n = 10 (and every log2n the reference gets right)."""
    import fftlib
    for log2n in range(5, 13):
        n = 1 << log2n
        table = golden["bitrev_ref_log2n_%d" % log2n]
        assert np.array_equal(table, O.bit_reverse_table(log2n))
        x = (np.arange(2 * n, dtype=np.float64).reshape(2, n) + 0j)
        a = fftlib.DeviceBuffer(x.nbytes)
        b = fftlib.DeviceBuffer(x.nbytes)
        a.upload(x)
        assert gpu_lib.fft_gpu_bit_reverse(a.handle, b.handle, n, 2, fftlib.PREC_F64) == 0
        y = b.download(x.shape, x.dtype)
        want = np.empty_like(x)
        want[:, table] = x
        assert np.array_equal(y, want)
        assert gpu_lib.fft_gpu_bit_reverse(a.handle, a.handle, n, 2, fftlib.PREC_F64) == 0  # in place
        assert np.array_equal(a.download(x.shape, x.dtype), want)
        a.free()
        b.free()


def test_c_api_fft_auto_and_plan(gpu_lib):
    """fft_auto / fft_plan_dft_1d / fft_execute / fft_execute_dft through the C ABI (config 1 plumbing)."""
    import fftlib
    n = 1024
    x = O.gen_two_tone(n, 0, 1)[0]
    out = np.zeros_like(x)
    assert gpu_lib.fft_auto(x.ctypes.data, out.ctypes.data, n, -1) == 0
    ref = O.oracle_fft(x, -1, "dit")
    assert rel(out, ref) <= 1e-12
    f, g = O.two_tone_bins(n, 0)
    assert abs(out[f] - n) < 1e-9 and abs(out[g] - n / 2) < 1e-9
    plan = gpu_lib.fft_plan_dft_1d(n, x.ctypes.data, out.ctypes.data, +1, fftlib.FFT_PREFER_GPU)
    assert plan
    gpu_lib.fft_execute(plan)
    assert rel(out, O.oracle_fft(x, 1, "dit")) <= 1e-12
    x2 = O.gen_lcg(n, 9, 1)[0]
    y2 = x2.copy()
    gpu_lib.fft_execute_dft(plan, y2.ctypes.data, y2.ctypes.data)  # in place with a plan made out of place
    assert rel(y2, O.oracle_fft(x2, 1, "dit")) <= 1e-12
    gpu_lib.fft_destroy_plan(plan)
    assert gpu_lib.fft_plan_dft_1d(0, x.ctypes.data, out.ctypes.data, -1, 0) is None
    assert gpu_lib.fft_plan_dft_1d(n, None, out.ctypes.data, -1, 0) is None
    assert gpu_lib.fft_auto(x.ctypes.data, out.ctypes.data, -5, -1) == -1
    z = x.copy()
    assert gpu_lib.fft_gpu_dft_1d(z.ctypes.data, z.ctypes.data, n, -1) == 0
    assert rel(z, ref) <= 1e-12
    xb = O.gen_lcg(256, 1, 7)
    yb = np.zeros_like(xb)
    assert gpu_lib.fft_gpu_dft_1d_batch(xb.ctypes.data, yb.ctypes.data, 256, 7, -1) == 0
    assert rel(yb, O.oracle_fft(xb, -1, "dit")) <= 1e-12
    for name in ("radix2_dit_fft_gpu", "radix2_fft_gpu", "radix4_fft_gpu", "split_radix_fft_gpu", "bluestein_fft_gpu"):
        w = x.copy()
        assert getattr(gpu_lib, name)(w.ctypes.data, n, -1) == 0
        assert rel(w, ref) <= 1e-12, name
    w = x[:100].copy()
    assert gpu_lib.radix4_fft_gpu(w.ctypes.data, 100, -1) == -1  # not a power of two: error, no exit()
    assert gpu_lib.fft_gpu_get_device_name() != b"No GPU"
    import ctypes as C
    tot, av = C.c_size_t(), C.c_size_t()
    gpu_lib.fft_gpu_get_memory_info(C.byref(tot), C.byref(av))
    assert tot.value > (100 << 30) and 0 < av.value <= tot.value
    p2 = gpu_lib.fft_gpu_plan_2d(4, 4, -1)  # a stub (NULL) in the reference and in round 1; real since round 2 (tests/test_gpu_ext.py)
    assert p2 is not None
    gpu_lib.fft_gpu_destroy_plan(p2)
    assert gpu_lib.fft_gpu_dft_2d(None, None, 4, 4, -1) == -1


def test_reference_property_tests(gpu_lib):
    """Properties 1-6 of the reference's tests/test_all.c:64-351 at n >= 32, its tolerances."""
    import fftlib
    rng = np.random.default_rng(7)
    for n in (32, 64, 128, 256, 512, 1024):
        tol = 1e-10
        imp = np.zeros(n, dtype=np.complex128); imp[0] = 1
        assert np.max(np.abs(np.abs(fftlib.fft(imp, -1)) - 1)) < tol                      # 1 impulse
        dc = np.ones(n, dtype=np.complex128)
        X = fftlib.fft(dc, -1)
        assert abs(X[0] - n) < tol * n and np.max(np.abs(X[1:])) < tol * n                # 2 DC
        a = rng.random(n) - 0.5 + 1j * (rng.random(n) - 0.5)
        b = rng.random(n) - 0.5 + 1j * (rng.random(n) - 0.5)
        lhs = fftlib.fft(2 * a + 3 * b, -1)
        assert np.max(np.abs(lhs - (2 * fftlib.fft(a, -1) + 3 * fftlib.fft(b, -1)))) < tol * n   # 3 linearity
        A = fftlib.fft(a, -1)
        assert abs(np.sum(np.abs(a) ** 2) - np.sum(np.abs(A) ** 2) / n) < tol * n         # 4 Parseval
        i = np.arange(n)
        s = (np.sin(2 * np.pi * 3 * i / n) + 0.5 * np.cos(2 * np.pi * 7 * i / n)).astype(np.complex128)
        assert np.max(np.abs(fftlib.fft(fftlib.fft(s, -1), 1) - s)) < tol                 # 5 round trip
        f = 5
        c = np.cos(2 * np.pi * f * i / n).astype(np.complex128)
        C_ = np.abs(fftlib.fft(c, -1))
        assert abs(C_[f] - n / 2) < tol * n and abs(C_[n - f] - n / 2) < tol * n          # 6 known transform
        mask = np.ones(n, bool); mask[[f, n - f]] = False
        assert np.max(C_[mask]) < tol * n


def test_full_size_analytic_two_tone_config2(gpu_lib):
    """BASELINE config 2 at FULL size (N=65536 fp32, batch=4096): every transform checked against the
    analytic spectrum X[f_b] = N, X[g_b] = N/2, 0 elsewhere (size-independent property; test_all.c:290-351)."""
    import fftlib
    n, batch = 65536, 4096
    step = 256
    plan = fftlib.Plan(n, step, -1, np.complex64)
    buf = fftlib.DeviceBuffer(n * step * 8)
    for b0 in range(0, batch, step):
        x = O.gen_two_tone(n, b0, step, np.complex64)
        buf.upload(x)
        plan.execute(buf, buf)
        y = buf.download((step, n), np.complex64)
        for i in (0, step - 1) if b0 in (0, 512, 2048, 3584, batch - step) else ():  # the oracle at the shard boundaries of an 8-way split
            ref = O.oracle_fft(x[i].astype(np.complex128), -1, "dit")
            assert rel(y[i], ref) <= 1e-4 and rel(y[i], ref) <= 2e-6, (b0, i)
        for i in range(0, step, 37):
            f, g = O.two_tone_bins(n, b0 + i)
            yy = y[i].astype(np.complex128)
            assert abs(yy[f] - n) / n < 1e-4 and abs(yy[g] - n / 2) / n < 1e-4
            yy[f] = 0
            yy[g] = 0
            assert np.linalg.norm(yy) / (n * np.sqrt(1.25)) < 1e-4
    plan.destroy()
    buf.free()


def _full_batch_through_shipped_schedule(n, batch, team_kernel=3):
    """ONE execute of the whole batch on the default plan -- the schedule bench.py times: the plan must have chosen the one-round-trip
    team kernel and the team kernel must have done the work (status 0, no fallback).  Every transform against its analytic spectrum
    (X[f_b] = N, X[g_b] = N/2, all the energy in those two bins), the oracle on both sides of the boundaries of an 8-way batch shard."""
    import fftlib
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    info = plan.info()
    assert info.team_kernel == team_kernel, (info.team_kernel, info.team_tiles)
    buf = fftlib.DeviceBuffer(n * batch * 8)
    x = O.gen_two_tone(n, 0, batch, np.complex64)
    buf.upload(x)
    plan.execute(buf, buf)
    assert plan.team_status() == 0  # the team kernel ran (1: its fallback did)
    y = buf.download((batch, n), np.complex64)
    idx = np.arange(batch)
    fg = np.array([O.two_tone_bins(n, b) for b in range(batch)])
    assert np.max(np.abs(y[idx, fg[:, 0]] - n)) / n < 1e-4
    assert np.max(np.abs(y[idx, fg[:, 1]] - n / 2)) / n < 1e-4
    for b0 in range(0, batch, 256):  # (in slices: a complex128 copy of the whole batch would be 4 GiB)
        tot = np.linalg.norm(y[b0:b0 + 256].astype(np.complex128), axis=1)
        assert np.max(np.abs(tot / (n * np.sqrt(1.25)) - 1)) < 1e-5, b0
    shard = batch // 8
    for b in sorted({0, batch - 1} | {g * shard - 1 for g in range(1, 8)} | {g * shard for g in range(1, 8)}):
        ref = O.oracle_fft(x[b].astype(np.complex128), -1, "dit")
        assert rel(y[b], ref) <= 1e-4 and rel(y[b], ref) <= 2e-6, b
    plan.destroy()
    buf.free()


def test_full_batch_config2_on_the_shipped_schedule(gpu_lib):
    """BASELINE config 2 (N=65536 fp32, batch=4096) as bench.py runs it: one execute, team_quad_kernel on teams of 2 claiming their
    transforms from the device counter (the chunked test above stays below the team kernel's crossover and runs the multi-pass plan)."""
    _full_batch_through_shipped_schedule(65536, 4096)


def test_full_batch_config4_shard_on_the_shipped_schedule(gpu_lib):
    """BASELINE config 4's per-GPU shard (N=262144 fp32, 1024 of the 8192 transforms) in one execute: team_quad_kernel on teams of 8."""
    _full_batch_through_shipped_schedule(1 << 18, 1024)


def test_roundtrip_full_size_config3(gpu_lib):
    """BASELINE config 3 at FULL size (N=2^20 fp32, batch=512): forward then inverse on device returns the
    input (idempotence-style property), plus oracle parity on the first and last transform."""
    import fftlib
    n, batch = 1 << 20, 512
    fwd = fftlib.Plan(n, batch, -1, np.complex64)
    inv = fftlib.Plan(n, batch, 1, np.complex64)
    buf = fftlib.DeviceBuffer(n * batch * 8)
    x = O.gen_two_tone(n, 0, batch, np.complex64)
    buf.upload(x)
    fwd.execute(buf, buf)
    y = buf.download((batch, n), np.complex64)
    # the oracle on nine transforms: first, last, and both sides of the boundaries of an 8-way batch shard
    for b in (0, 1, 63, 64, 255, 256, 447, 448, batch - 1):
        ref = O.oracle_fft(x[b].astype(np.complex128), -1, "dit")
        assert rel(y[b], ref) <= 1e-4 and rel(y[b], ref) <= 2e-6
        f, g = O.two_tone_bins(n, b)
        assert abs(y[b][f] - n) / n < 1e-4 and abs(y[b][g] - n / 2) / n < 1e-4
    inv.execute(buf, buf)
    z = buf.download((batch, n), np.complex64)
    assert rel(z, x) <= 1e-4
    assert rel(z, x) <= 5e-6
    fwd.destroy(); inv.destroy(); buf.free()


def test_length_one_and_single_transform(gpu_lib):
    import fftlib
    x = lcg(1, 5, np.complex128, seed=2)
    assert np.array_equal(fftlib.fft(x, -1), x) and np.array_equal(fftlib.fft(x, 1, inplace=False), x)
    for n in (2, 1024, 1 << 20):  # batch = 1: one (mostly padded) tile per pass
        x1 = lcg(n, 1, np.complex128, seed=n)[0]
        assert rel(fftlib.fft(x1, -1), O.oracle_fft(x1, -1, "exact")) <= 1e-14


def test_additive_api_plan_info_streams_timing(gpu_lib):
    """Additive entry points of include/fft_hip.h: device count / set_device, plan info, async execute + sync,
    HIP-event timing, per-pass profiling, raw-pointer execute on a caller-provided stream handle (NULL = own)."""
    import ctypes as C
    import fftlib
    assert gpu_lib.fft_gpu_device_count() >= 1
    assert gpu_lib.fft_gpu_set_device(0) == 0 and gpu_lib.fft_gpu_set_device(9999) == -1
    assert gpu_lib.fft_gpu_get_device_hip() == 0
    n, batch = 1 << 16, 6
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    info = plan.info()
    assert (info.n, info.batch, info.direction, info.precision) == (n, batch, -1, fftlib.PREC_F32)
    assert info.n_passes == 2 and info.factors[0] * info.factors[1] == n and info.bluestein_m == 0
    assert info.workspace_bytes >= n * 8
    x = lcg(n, batch, np.complex64, seed=5)
    a = fftlib.DeviceBuffer(x.nbytes)
    b = fftlib.DeviceBuffer(x.nbytes)
    a.upload(x)
    assert gpu_lib.fft_gpu_execute_async(plan.handle, a.handle, b.handle) == 0
    assert gpu_lib.fft_gpu_plan_sync(plan.handle) == 0
    ref = O.oracle_fft(x.astype(np.complex128), -1, "dit")
    assert rel(b.download(x.shape, x.dtype), ref) <= 2e-6
    assert gpu_lib.fft_gpu_plan_set_stream(plan.handle, None) == 0
    ms = plan.timed(a.ptr, b.ptr, 3)
    assert ms > 0
    prof = plan.profile_passes(a.ptr, b.ptr)
    assert len(prof) == 2 and all(m > 0 and c >= 1 for m, c in prof)
    assert rel(b.download(x.shape, x.dtype), ref) <= 2e-6
    # fp32 host-pointer conveniences
    y = np.zeros_like(x)
    assert gpu_lib.fft_gpu_dft_1d_batch_f32(x.ctypes.data, y.ctypes.data, n, batch, -1) == 0
    assert rel(y, ref) <= 2e-6
    y1 = x[0].copy()
    assert gpu_lib.fft_gpu_dft_1d_f32(y1.ctypes.data, y1.ctypes.data, n, -1) == 0
    assert rel(y1, ref[0]) <= 2e-6
    # undersized buffers are refused, not overrun
    small = fftlib.DeviceBuffer(64)
    gpu_lib.fft_gpu_execute(plan.handle, small.handle, small.handle)
    assert gpu_lib.fft_gpu_copy_h2d_bytes_hip(small.handle, x.ctypes.data, x.nbytes) == -1
    bi = fftlib.Plan(1000, 2, 1, np.complex128).info()
    assert bi.bluestein_m == 2048 and bi.direction == 1
    for buf in (a, b, small):
        buf.free()
    plan.destroy()


def test_batch_larger_than_one_launch_group(gpu_lib):
    """The batch is processed in launch groups (FFT_HIP_CHUNK_MB); 40 transforms in groups of 8 must equal the
    oracle for every transform, in place and out of place, forward and inverse."""
    import os
    import fftlib
    n, batch = 1 << 16, 40
    x = lcg(n, batch, np.complex64, seed=11)
    fftlib.set_policy(chunk_mb=4)
    try:
        plan = fftlib.Plan(n, batch, -1, np.complex64)
        assert plan.info().chunk_batch == 8
        plan.destroy()
        for d, inplace in ((-1, True), (1, False)):
            y = fftlib.fft(x, d, inplace=inplace)
            ref = O.oracle_fft(x.astype(np.complex128), d, "dit")
            for b in range(batch):
                assert rel(y[b], ref[b]) <= 2e-6, (d, b)
    finally:
        fftlib.set_policy(chunk_mb=0)


@pytest.mark.parametrize("log2n", [21, 22, 24])
def test_three_pass_sizes(gpu_lib, log2n):
    """Sizes past the two-pass range use three passes; checked against the analytic two-tone spectrum (fp32)."""
    import fftlib
    n = 1 << log2n
    x = O.gen_two_tone(n, 1, 2, np.complex64)
    plan = fftlib.Plan(n, 2, -1, np.complex64)
    assert plan.info().n_passes == 3
    plan.destroy()
    y = fftlib.fft(x, -1).astype(np.complex128)
    for b in range(2):
        f, g = O.two_tone_bins(n, 1 + b)
        assert abs(y[b, f] - n) / n < 1e-5 and abs(y[b, g] - n / 2) / n < 1e-5
        y[b, f] = 0
        y[b, g] = 0
        assert np.linalg.norm(y[b]) / (n * np.sqrt(1.25)) < 1e-5
    back = fftlib.fft(fftlib.fft(x, -1), 1)
    assert rel(back, x) <= 5e-6


def test_no_device_memory_leak_and_reinit(gpu_lib):
    """Plans and buffers give their device memory back; cleanup + init cycles keep working."""
    import ctypes as C
    import fftlib

    def avail():
        tot, av = C.c_size_t(), C.c_size_t()
        gpu_lib.fft_gpu_get_memory_info(C.byref(tot), C.byref(av))
        return av.value

    x = lcg(4096, 3, np.complex64, seed=1)
    fftlib.fft(x, -1)
    before = avail()
    for i in range(60):
        for n, dt in ((1 << 16, np.complex64), (1000, np.complex128), (1 << 12, np.complex128)):
            p = fftlib.Plan(n, 4, -1 if i % 2 else 1, dt)
            b = fftlib.DeviceBuffer(n * 4 * np.dtype(dt).itemsize)
            p.execute(b, b)
            p.destroy()
            b.free()
    after = avail()
    assert before - after < (64 << 20), (before, after)  # nothing accumulates (allow allocator slack)
    gpu_lib.fft_gpu_cleanup()
    assert gpu_lib.fft_gpu_get_backend() == 0 and gpu_lib.fft_gpu_get_device_name() == b"No GPU"
    assert gpu_lib.fft_gpu_alloc(16) is None  # not initialised: refused
    assert gpu_lib.fft_gpu_init(fftlib.FFT_GPU_AUTO) == 0 and gpu_lib.fft_gpu_get_backend() == fftlib.FFT_GPU_HIP
    y = fftlib.fft(x, -1)
    assert rel(y, O.oracle_fft(x.astype(np.complex128), -1, "dit")) <= 2e-6


def test_full_size_config4_shard_and_config5(gpu_lib):
    """BASELINE config 4's per-GPU shard at 8 GPUs (N=262144 fp32, 1024 transforms) against the analytic spectrum
    of EVERY transform it owns (ranks 0 and 7), and config 5 at full batch (N=1000003 fp64 x 64): forward then
    inverse returns the input."""
    import fftlib
    n, per_gpu = 1 << 18, 1024
    plan = fftlib.Plan(n, 256, -1, np.complex64)
    buf = fftlib.DeviceBuffer(n * 256 * 8)
    for rank in (0, 7):
        for b0 in range(rank * per_gpu, (rank + 1) * per_gpu, 256):
            x = O.gen_two_tone(n, b0, 256, np.complex64)
            buf.upload(x)
            plan.execute(buf, buf)
            y = buf.download((256, n), np.complex64)
            idx = np.arange(256)
            fg = np.array([O.two_tone_bins(n, b0 + i) for i in range(256)])
            assert np.max(np.abs(y[idx, fg[:, 0]] - n)) / n < 1e-4
            assert np.max(np.abs(y[idx, fg[:, 1]] - n / 2)) / n < 1e-4
            tot = np.linalg.norm(y.astype(np.complex128), axis=1)
            assert np.max(np.abs(tot / (n * np.sqrt(1.25)) - 1)) < 1e-5  # all energy sits in the two bins
            for i in (0, 255):  # and the oracle on the first / last transform of every quarter of the two shards (16 transforms)
                ref = O.oracle_fft(x[i].astype(np.complex128), -1, "dit")
                assert rel(y[i], ref) <= 1e-4 and rel(y[i], ref) <= 2e-6, (rank, b0, i)
    plan.destroy()
    buf.free()
    n, batch = 1000003, 64
    x = O.gen_lcg(n, 0, batch)
    fwd = fftlib.Plan(n, batch, -1, np.complex128)
    inv = fftlib.Plan(n, batch, 1, np.complex128)
    d = fftlib.DeviceBuffer(x.nbytes)
    d.upload(x)
    fwd.execute(d, d)
    y = d.download(x.shape, x.dtype)
    assert rel(y[0], np.fft.fft(x[0])) <= 1e-6 and rel(y[-1], np.fft.fft(x[-1])) <= 1e-6
    # the oracle (the reference's own chirp-z, restated) on nine transforms: first, last, 8-way shard boundaries
    for b in (0, 7, 8, 15, 16, 31, 32, 56, batch - 1):
        ref = O.oracle_fft(x[b], -1, "bluestein")
        assert rel(y[b], ref) <= 1e-6 and rel(y[b], ref) <= 2e-9, b
    inv.execute(d, d)
    assert rel(d.download(x.shape, x.dtype), x) <= 1e-6
    fwd.destroy(); inv.destroy(); d.free()


# ---------------------------------------------------------------------------
# team kernel (csrc/fft_team.h): a whole transform per team of CUs of one XCD, one HBM round trip.  FFT_HIP_TEAM=2
# plans it for every size it is built for and any batch (the default policy only uses it where it measured faster and
# from 2 GiB per execute up), so that all ten instantiations are parity-checked on the device.
# ---------------------------------------------------------------------------
def test_first_in_place_team_execute_runs_from_a_staged_copy(gpu_lib):
    """Round-3 review item 6: the FIRST in-place execute of a plan (its team kernel not yet seen to end well on this device) reads a
    staged copy of its input, so that a team-kernel timeout could be repaired like an out-of-place one; once a sync has seen status 0
    the staging stops.  Both executes must give the same, correct spectrum, and the second must not be slower for a copy."""
    import fftlib
    fftlib.set_policy(team=2)
    n, batch = 1 << 18, 64
    x = lcg(n, batch, np.complex64, seed=5)
    ref = np.fft.fft(x.astype(np.complex128), axis=1)
    buf = fftlib.DeviceBuffer(x.nbytes)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    assert plan.info().team_kernel == 3
    buf.upload(x)
    plan.execute_ptr(buf.ptr, buf.ptr)  # staged
    assert plan.team_status() == 0
    y1 = buf.download(x.shape, np.complex64)
    assert rel(y1, ref) <= 2e-6
    buf.upload(x)
    plan.execute_ptr(buf.ptr, buf.ptr)  # proven: unstaged
    assert plan.team_status() == 0
    assert np.array_equal(buf.download(x.shape, np.complex64), y1)
    plan.set_option(fftlib.OPT_TEAM_NO_REPLAY, 1)  # (accepted by plain 1D plans)
    plan.destroy()
    # two in-place executes of a FRESH plan on two buffers before any sync: the first is staged, the second finds the staging area in use
    # and runs unstaged -- both must be right
    buf2 = fftlib.DeviceBuffer(x.nbytes)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    buf.upload(x)
    buf2.upload(x[::-1].copy())
    plan.execute_ptr(buf.ptr, buf.ptr)
    plan.execute_ptr(buf2.ptr, buf2.ptr)
    assert plan.team_status() == 0
    assert np.array_equal(buf.download(x.shape, np.complex64), y1)
    assert np.array_equal(buf2.download(x.shape, np.complex64), y1[::-1])
    plan.destroy()
    buf.free()
    buf2.free()


def _team_plan(monkeypatch, n, batch, direction, dtype, mode="2"):
    import fftlib
    fftlib.set_policy(team=int(mode))  # tests/conftest.py puts the default policy back after every test
    plan = fftlib.Plan(n, batch, direction, dtype)
    return plan


@pytest.mark.parametrize("log2n,dtype", [(20, np.complex64), (19, np.complex64), (18, np.complex64), (17, np.complex64), (15, np.complex64),
                                         (16, np.complex64), (19, np.complex128), (18, np.complex128),
                                         (17, np.complex128), (16, np.complex128), (15, np.complex128), (14, np.complex128)])
def test_team_kernel_vs_oracle(gpu_lib, monkeypatch, log2n, dtype):
    """Every device instantiation: teams of 32 (a whole XCD), 16, 8, 4 and 2 CUs -- and of one (n = 2^15 fp32, team_quad_kernel)."""
    import fftlib
    tiles = 4
    n = 1 << log2n
    n_teams = 8 << (20 - log2n - (1 if dtype == np.complex128 else 0))
    batch = 2 * n_teams + 3  # ragged over the teams
    x = lcg(n, batch, dtype, seed=log2n)
    buf = fftlib.DeviceBuffer(x.nbytes)
    out = fftlib.DeviceBuffer(x.nbytes)
    for d in (-1, 1):
        plan = _team_plan(monkeypatch, n, batch, d, dtype)
        assert plan.info().team_tiles == tiles
        # team_quad_kernel at every built size but fp64 2^17 ... 2^19, which keep round 2's team_defer_kernel (fft_team_quad_decl.h)
        assert plan.info().team_kernel == (2 if (dtype == np.complex128 and log2n >= 17) else 3)
        buf.upload(x)
        out.upload(np.full_like(x, np.nan))
        plan.execute_ptr(buf.ptr, out.ptr)
        assert plan.team_status() == 0, "the team kernel must have done the work (teams formed, no timeout)"
        y = out.download(x.shape, dtype)
        # the oracle on the whole first round of teams (one transform per team), the team boundaries and the ragged tail
        for b in sorted(set(list(range(0, n_teams, max(1, n_teams // 8))) + [n_teams - 1, n_teams, 2 * n_teams - 1, 2 * n_teams, batch - 1])):
            ref = O.oracle_fft(x[b:b + 1].astype(np.complex128), d, "dit")
            r = rel(y[b:b + 1], ref)
            assert r <= TOL[np.dtype(dtype)] and r <= TIGHT[np.dtype(dtype)], (log2n, d, b, r)
        # in place, and bit-identical to out of place
        plan.execute_ptr(buf.ptr, buf.ptr)
        assert plan.team_status() == 0
        assert np.array_equal(buf.download(x.shape, dtype), y)
        # the multi-pass schedule computes the same spectrum
        fftlib.set_policy(team=0)
        plan2 = fftlib.Plan(n, batch, d, dtype)
        assert plan2.info().team_tiles == 0
        buf.upload(x)
        plan2.execute_ptr(buf.ptr, out.ptr)
        plan2.sync()
        assert rel(out.download(x.shape, dtype), y) <= (1e-6 if dtype == np.complex64 else 1e-14)
        plan.destroy()
        plan2.destroy()
    buf.free()
    out.free()


@pytest.mark.parametrize("log2n,dtype", [(20, np.complex64), (19, np.complex64), (18, np.complex64), (17, np.complex64), (16, np.complex64),
                                         (15, np.complex64), (16, np.complex128), (15, np.complex128), (14, np.complex128)])
def test_default_policy_switches_schedule_at_the_measured_crossover(gpu_lib, log2n, dtype):
    """Default policy (round 3: profiles/r3_batch_crossover.txt): the one-round-trip kernel runs from 128 MiB per execute (n = 2^20, 2^19) /
    256 MiB (the other sizes), the multi-pass schedule below; one transform short of the crossover and one past it, out of place and in
    place, EVERY transform against numpy (the team kernels take their transforms by claiming: no fixed transform -> team map)."""
    import fftlib
    fftlib.set_policy(team=1, min_batch=0)
    n = 1 << log2n
    esz = np.dtype(dtype).itemsize
    mib = 128 if (dtype == np.complex64 and log2n >= 19) else 256
    mb = (mib << 20) // (n * esz)
    x8 = lcg(n, 8, dtype, seed=40 + log2n)
    ref = np.fft.fft(x8.astype(np.complex128), axis=1)
    tol = TIGHT[np.dtype(dtype)]  # (and with it the contractual TOL)
    for batch, want in ((mb - 1, -1), (mb + 1, 0)):
        x = x8[np.arange(batch) % 8]
        buf, out = fftlib.DeviceBuffer(x.nbytes), fftlib.DeviceBuffer(x.nbytes)
        plan = fftlib.Plan(n, batch, -1, dtype)
        for inplace in (False, True):
            buf.upload(x)
            plan.execute_ptr(buf.ptr, buf.ptr if inplace else out.ptr)
            assert plan.team_status() == want, (log2n, batch, plan.team_status())
            y = (buf if inplace else out).download(x.shape, dtype)
            err = max(float(np.linalg.norm(y[i] - ref[i % 8]) / np.linalg.norm(ref[i % 8])) for i in range(batch))
            assert err <= tol, (log2n, batch, inplace, err)
        plan.destroy()
        buf.free()
        out.free()


def test_team_kernel_default_policy_and_full_size(gpu_lib, monkeypatch):
    """BASELINE configs[2] as bench.py runs it: N = 2^20 fp32 x 512 takes the team kernel by default; every one of
    the 512 spectra is checked against the analytic two-tone answer; a batch of 8 (below the measured crossover of 16
    transforms, profiles/r3_batch_crossover.txt) keeps the two-pass schedule."""
    import fftlib
    fftlib.set_policy(team=1, min_batch=0)
    n, batch = 1 << 20, 512
    small = fftlib.Plan(n, 8, -1, np.complex64)
    assert small.info().team_tiles == 4  # planned ...
    xs = lcg(n, 8, np.complex64, seed=5)
    bs = fftlib.DeviceBuffer(xs.nbytes)
    bs.upload(xs)
    small.execute_ptr(bs.ptr, bs.ptr)
    assert small.team_status() == -1  # ... but not launched below the crossover batch
    small.destroy()
    bs.free()
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    x = O.gen_two_tone(n, 0, batch, np.complex64)
    buf = fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan.execute_ptr(buf.ptr, buf.ptr)
    assert plan.team_status() == 0
    y = buf.download(x.shape, np.complex64)
    bb = np.arange(batch, dtype=np.int64)
    f = (1 + 7 * bb) % n
    g = (n // 3 + 13 * bb) % n
    g = np.where(g == f, (g + 1) % n, g)
    rows = np.arange(batch)
    assert np.abs(y[rows, f] - n).max() / n < 1e-4
    assert np.abs(y[rows, g] - n / 2).max() / n < 1e-4
    want = n * 1.25 ** 0.5
    for s in range(0, batch, 32):  # Parseval, 32 transforms at a time (fp64 accumulation without a 9 GB temporary)
        tot = np.linalg.norm(y[s:s + 32].astype(np.complex128), axis=1)
        assert np.abs(tot - want).max() / want < 1e-4
    # inverse of the forward result returns the input (round trip at full size)
    inv = fftlib.Plan(n, batch, 1, np.complex64)
    inv.execute_ptr(buf.ptr, buf.ptr)
    assert inv.team_status() == 0
    assert rel(buf.download(x.shape, np.complex64), x) < 1e-5
    plan.destroy()
    inv.destroy()
    buf.free()


def test_team_kernel_fallback_on_device(gpu_lib, monkeypatch):
    """FFT_GPU_OPT_TEAM_FORCE_FALLBACK makes the kernel's placement check fail on a healthy device: status 1, nothing
    touched by the team kernel, the two-pass launches queued behind it deliver the (in-place) result; after three
    fallbacks in a row the plan stops launching the team kernel."""
    import fftlib
    n, batch = 1 << 20, 16
    plan = _team_plan(monkeypatch, n, batch, -1, np.complex64)
    plan.set_option(fftlib.OPT_TEAM_FORCE_FALLBACK, 1)
    x = lcg(n, batch, np.complex64, seed=9)
    buf = fftlib.DeviceBuffer(x.nbytes)
    ref = O.oracle_fft(x[:2].astype(np.complex128), -1, "dit")
    for it in range(4):
        buf.upload(x)
        plan.execute_ptr(buf.ptr, buf.ptr)
        st = plan.team_status()
        assert st == 1, (it, st)  # the fourth execute no longer launches the team kernel: last known status stays 1
        assert rel(buf.download(x.shape, np.complex64)[:2], ref) <= TIGHT[np.dtype(np.complex64)]
    # round 4: the kernel is paused, not retired -- 64 executes on the multi-pass schedule (the fourth above was the first of them), then it
    # is tried again; the device is healthy now, so it does the work
    plan.set_option(fftlib.OPT_TEAM_FORCE_FALLBACK, 0)
    for it in range(63):
        plan.execute_ptr(buf.ptr, buf.ptr)
    assert plan.team_status() == 1  # still the last known status: nothing launched
    buf.upload(x)
    plan.execute_ptr(buf.ptr, buf.ptr)
    assert plan.team_status() == 0
    assert rel(buf.download(x.shape, np.complex64)[:2], ref) <= TIGHT[np.dtype(np.complex64)]
    plan.destroy()
    buf.free()


def test_team_kernel_repeatable_bit_for_bit(gpu_lib, monkeypatch):
    """The hand-over between the workgroups of a team relies on same-XCD visibility through the shared L2 (plain
    stores, L1-bypassing loads, generation flags).  A stale or torn read anywhere would change some output word:
    24 back-to-back executes of two team geometries must reproduce the first result bit for bit (tools/team_stress.py
    is the long version: 300 executes of up to 4 GiB each)."""
    import fftlib
    fftlib.set_policy(team=2)
    for log2n, batch in ((20, 96), (18, 259)):
        n = 1 << log2n
        x = lcg(n, batch, np.complex64, seed=batch)
        buf = fftlib.DeviceBuffer(x.nbytes)
        out = fftlib.DeviceBuffer(x.nbytes)
        buf.upload(x)
        plan = fftlib.Plan(n, batch, -1, np.complex64)
        assert plan.info().team_tiles == 4
        plan.execute_ptr(buf.ptr, out.ptr)
        assert plan.team_status() == 0
        first = out.download(x.shape, np.complex64)
        for it in range(6):
            for _ in range(4):
                plan.execute_ptr(buf.ptr, out.ptr)
            assert plan.team_status() == 0
            assert np.array_equal(out.download(x.shape, np.complex64).view(np.uint8), first.view(np.uint8)), (log2n, it)
        plan.destroy()
        buf.free()
        out.free()


def test_team_kernels_of_two_plans_on_two_streams(gpu_lib, monkeypatch):
    """Two plans own two streams: their team kernels may be dispatched at the same time, and each needs every CU.
    Whatever the hardware does with them (one after the other; or both partially resident, in which case the team
    formation of both gives up after its bounded wait and the multi-pass fallbacks run), both results must be right
    and nothing may hang."""
    import fftlib
    fftlib.set_policy(team=2)
    n, batch = 1 << 18, 64
    xs = [lcg(n, batch, np.complex64, seed=s) for s in (21, 22)]
    plans = [fftlib.Plan(n, batch, -1, np.complex64) for _ in xs]
    bufs = []
    for x in xs:
        b = fftlib.DeviceBuffer(x.nbytes)
        b.upload(x)
        bufs.append(b)
    for rep in range(3):
        for p, b in zip(plans, bufs):  # no sync in between: the two streams run concurrently
            p.execute_ptr(b.ptr, b.ptr)
        for p, b, x in zip(plans, bufs, xs):
            assert p.team_status() in (0, 1)
            y = b.download(x.shape, np.complex64)
            ref = O.oracle_fft(x[:2].astype(np.complex128), -1, "dit")
            assert rel(y[:2], ref) <= TIGHT[np.dtype(np.complex64)]
            b.upload(x)
    for p in plans:
        p.destroy()
    for b in bufs:
        b.free()


@pytest.mark.parametrize("defer,nt,pair", [("0", "0", "1"), ("1", "0", "1"), ("1", "7", "1"), ("0", "7", "1"), ("1", "0", "0"), ("1", "3", "0")])
def test_team_kernel_forced_variants(gpu_lib, defer, nt, pair):
    """Both team kernels at every kind of size, whatever the per-size default is: FFT_HIP_TEAM_DEFER=0 team_fft_kernel,
    =1 team_defer_kernel (the deferred row phase), with the default cache-policy bits off (FFT_HIP_TEAM_NT=0) and all
    on (7: non-temporal column-tile DMA, result stores and window loads), with and without the paired row tiles of
    fp32 n = 2^19, 2^20 (FFT_HIP_TEAM_PAIR; the default is on).  The variables are read once per process, hence the fresh one."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import fftlib, oracle_lib as O\n"
        "fftlib.init()\n"
        "for log2n, dt in ((20, np.complex64), (19, np.complex64), (18, np.complex64), (17, np.complex128), (18, np.complex128)):\n"
        "    n, batch = 1 << log2n, 70\n"
        "    x = O.gen_lcg(n, 33, batch).astype(dt)\n"
        "    buf = fftlib.DeviceBuffer(x.nbytes)\n"
        "    for d in (-1, 1):\n"
        "        p = fftlib.Plan(n, batch, d, dt)\n"
        "        buf.upload(x); p.execute_ptr(buf.ptr, buf.ptr)\n"
        "        assert p.team_status() == 0\n"
        "        y = buf.download(x.shape, dt)\n"
        "        for b in (0, 33, 69):\n"
        "            ref = O.oracle_fft(x[b:b+1].astype(np.complex128), d, 'dit')\n"
        "            r = float(np.linalg.norm(y[b:b+1] - ref) / np.linalg.norm(ref))\n"
        "            assert r < (2e-6 if dt == np.complex64 else 2e-11), (log2n, d, b, r)\n"
        "        p.destroy()\n"
        "    buf.free()\n"
        "print('ok')\n"
    ) % (os.path.join(ROOT, "fft-implementation-in-c_amd"), os.path.join(ROOT, "tests"))
    # kernel-variant switches exist only in the -DFFT_EXPERIMENTS build of the library
    env = dict(os.environ, FFT_HIP_TEAM="2", FFT_HIP_TEAM_DEFER=defer, FFT_HIP_TEAM_NT=nt, FFT_HIP_TEAM_PAIR=pair,
               FFT_LIB_PATH=os.path.join(ROOT, "fft-implementation-in-c_amd", "libfft_mi355x_exp.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_team_kernel_even_odd_row_split(gpu_lib, monkeypatch):
    """FFT_HIP_TEAM_ASPLIT=1: the column step of n = 2^20 fp32 on half-height, double-width tiles (128-byte row
    segments) joined by a radix-2 butterfly in registers -- an experiment (slower than the plain tiles), kept
    parity-green."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import fftlib, oracle_lib as O\n"
        "fftlib.init()\n"
        "n, batch = 1 << 20, 24\n"
        "x = O.gen_lcg(n, 31, batch).astype(np.complex64)\n"
        "buf = fftlib.DeviceBuffer(x.nbytes); buf.upload(x)\n"
        "for d in (-1, 1):\n"
        "    p = fftlib.Plan(n, batch, d, np.complex64)\n"
        "    buf.upload(x); p.execute_ptr(buf.ptr, buf.ptr)\n"
        "    assert p.team_status() == 0\n"
        "    y = buf.download(x.shape, np.complex64)\n"
        "    for b in (0, 8, 23):\n"
        "        ref = O.oracle_fft(x[b:b+1].astype(np.complex128), d, 'dit')\n"
        "        r = float(np.linalg.norm(y[b:b+1] - ref) / np.linalg.norm(ref))\n"
        "        assert r < 2e-6, (d, b, r)\n"
        "print('ok')\n"
    ) % (os.path.join(ROOT, "fft-implementation-in-c_amd"), os.path.join(ROOT, "tests"))
    env = dict(os.environ, FFT_HIP_TEAM="2", FFT_HIP_TEAM_ASPLIT="1",  # read once per process: a fresh one
               FFT_LIB_PATH=os.path.join(ROOT, "fft-implementation-in-c_amd", "libfft_mi355x_exp.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
