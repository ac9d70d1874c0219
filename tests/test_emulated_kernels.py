"""The UNMODIFIED kernel and planner source (fft_kernels.h / fft_engine.h), compiled for the CPU with a
workgroup = host threads (tests/emu), checked against the oracle.  This validates the index algebra (Stockham
digit order, four-step strides, LDS layouts, persistent tile walk + prefetch) in a container without a GPU.
It is NOT a product path: the shipped library has no CPU implementation."""
import os

import numpy as np
import pytest

import emu_lib as E
import oracle_lib as O


def rel(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


TOL = {np.complex64: 3e-7, np.complex128: 2e-14}
# team kernels, fp32: stage and inter-step twiddles are built as powers (products up to 4 deep, FFT_TEAM_TW_TREE) instead
# of being read from tables one by one -- 14 of 15 LDS reads per stage traded for rounding: ~5e-7 instead of ~1.5e-7
TEAM_TOL = {np.complex64: 1.5e-6, np.complex128: 2e-14}


def oracle(x, d):
    n = x.shape[-1]
    pow2 = (n & (n - 1)) == 0
    return O.oracle_fft(x.astype(np.complex128), d, ("exact" if n != 4 and n != 8 and n != 16 else "naive") if pow2 else "naive")


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 512, 1024, 2048])
def test_single_pass(dtype, n):
    x = O.gen_lcg(n, n, 19).astype(dtype)  # 19 transforms: ragged last tile
    for d in (-1, 1):
        y, info = E.emu_fft(x, d)
        assert info[0] == 1
        assert rel(y, oracle(x, d)) < TOL[dtype]


@pytest.mark.parametrize("algo", [1, 2, 3, 4])
@pytest.mark.parametrize("n", [8, 64, 1024])
def test_families(algo, n):
    """radix-2 / radix-4 / split-radix LDS families and the bit-reversal + global radix-2 DIT path."""
    x = O.gen_lcg(n, 2, 5).astype(np.complex64)
    y, _ = E.emu_fft(x, -1, algo)
    assert rel(y, oracle(x, -1)) < 3e-7
    xi = O.gen_lcg(n, 3, 3)
    yi, _ = E.emu_fft(xi, 1, algo, inplace=True)
    assert rel(yi, oracle(xi, 1)) < 2e-14


@pytest.mark.parametrize("dtype,n,batch", [(np.complex64, 8192, 3), (np.complex64, 65536, 2), (np.complex128, 16384, 2)])
def test_two_pass(dtype, n, batch):
    x = O.gen_lcg(n, 1, batch).astype(dtype)
    y, info = E.emu_fft(x, -1)
    assert info[0] == 2
    assert rel(y, oracle(x, -1)) < TOL[dtype]
    y, _ = E.emu_fft(x, 1, inplace=True)
    assert rel(y, oracle(x, 1)) < TOL[dtype]


def test_three_pass_forced_by_small_lds():
    x = O.gen_lcg(65536, 4, 1).astype(np.complex64)
    y, info = E.emu_fft(x, -1, 0, lds_budget=12000)
    assert info[0] == 3
    assert rel(y, oracle(x, -1)) < 3e-7
    xd = O.gen_lcg(65536, 5, 1)
    y, info = E.emu_fft(xd, 1, 0, lds_budget=12000, inplace=True)
    assert info[0] == 3
    assert rel(y, oracle(xd, 1)) < 2e-14


@pytest.mark.parametrize("n", [128, 256, 512, 1024])
def test_one_wave_radix2_dit_with_shuffles(n):
    """wave_dit_kernel: LDS bit-reversal permutation + in-register stages + __shfl_xor cross-lane stages."""
    for dtype in (np.complex64, np.complex128):
        x = O.gen_lcg(n, 9, 6).astype(dtype)  # 6 transforms: one and a half workgroups of 4 waves
        for d in (-1, 1):
            y, _ = E.emu_fft(x, d, 6)
            assert rel(y, oracle(x, d)) < TOL[dtype]
            assert rel(y, O.oracle_fft(x.astype(np.complex128), d, "dit")) < 1e-6  # the reference's own radix-2 path


def test_length_one_is_a_scaled_copy():
    x = O.gen_lcg(1, 0, 7)
    for d in (-1, 1):
        y, _ = E.emu_fft(x, d)
        assert np.array_equal(y, x)  # n = 1: DFT is the identity, the inverse scale 1/n is 1


def test_bluestein_emulated():
    for n in (3, 31, 100, 1009):
        x = O.gen_lcg(n, n, 2)
        for d in (-1, 1):
            y, info = E.emu_fft(x, d)
            assert info[0] >= 10
            assert rel(y, O.oracle_fft(x, d, "bluestein")) < 1e-12


@pytest.mark.parametrize("n,dtype,lds,passes,chained", [
    (100, np.complex64, 0, 1, 0),        # single-pass hooks, fp32 pairs, even pitch
    (101, np.complex64, 0, 1, 0),        # odd pitch: the user's rows are not 16-byte aligned -> value-by-value accesses
    (1009, np.complex64, 0, 1, 0),       # m = 2048, odd n: the last pair straddles n
    (1009, np.complex128, 4096, 2, 0),   # m = 2048 in two passes: load hook on the column pass, store hooks on the row pass
    (1009, np.complex64, 4096, 2, 0),
    (1500, np.complex64, 4096, 3, 1),    # m = 4096 = 16 x 16 x 16: the forward's last pass and the inverse's first share a tile
    (3001, np.complex128, 2048, 3, 0),   # m = 8192 in three passes
    (3001, np.complex64, 2048, 3, 1),    # 16 x 32 x 16: a split with equal ends is preferred (Pow2Plan::prefer_chain)
    (2000, np.complex128, 40000, 2, 1),  # m = 4096 = 64 x 64, two passes, chained (tile-major scratch of the inverse)
    (2000, np.complex64, 40000, 2, 1),
    (5000, np.complex128, 4096, 3, 0),   # m = 16384 = 16 x 32 x 32: first and last tiles differ, no chaining
])
def test_bluestein_fused_ends(n, dtype, lds, passes, chained, monkeypatch):
    """Bluestein with its modulate / pointwise / demodulate steps fused into the first load and last store of the two
    power-of-two transforms (fftk::TileHooks) against the oracle, and bit-for-bit ... no: to rounding ... against the
    same plan run with the three steps as kernels of their own."""
    batch = 5
    x = O.gen_lcg(n, n, batch).astype(dtype)
    tol = 1e-12 if dtype == np.complex128 else 2e-5
    for d in (-1, 1):
        for inplace in (False, True):
            y, info = E.emu_fft(x, d, lds_budget=lds, inplace=inplace)
            # info[4]: 1 fused ends, 2 also the forward's last and the inverse's first pass as ONE kernel (fft_kernels_chain.h),
            # 3 single-pass m: modulate -> FFT -> product -> inverse FFT -> demodulate as ONE kernel (TileHooks::mid_tab)
            # (chained = 0: not required -- two-pass plans with unequal factors chain through the mirrored split where the tiles agree)
            assert info[0] == 10 + passes and (info[4] == 3 if passes == 1 else info[4] == 2 if chained else info[4] in (1, 2)), info[:5]
            was_chained = info[4] >= 2
            assert rel(y, O.oracle_fft(x.astype(np.complex128), d, "bluestein")) < tol, (n, d, inplace)
    if was_chained:
        monkeypatch.setenv("FFT_EMU_NO_CHAIN", "1")
        y1, info = E.emu_fft(x, -1, lds_budget=lds)
        assert info[4] == 1
        assert rel(y1, y.astype(np.complex128) if d == -1 else E.emu_fft(x, -1, lds_budget=lds)[0].astype(np.complex128)) < tol
        monkeypatch.delenv("FFT_EMU_NO_CHAIN")
    monkeypatch.setenv("FFT_EMU_NO_FUSION", "1")
    y2, info = E.emu_fft(x, -1, lds_budget=lds)
    assert info[4] == 0
    assert rel(y2, O.oracle_fft(x.astype(np.complex128), -1, "bluestein")) < tol


def test_bit_reversal_kernel_emulated(golden):
    import ctypes as C
    n = 1024
    x = (np.arange(2 * n, dtype=np.float64).reshape(2, n) + 0j)
    out = np.zeros_like(x)
    E.lib().emu_bitrev(x.ctypes.data, out.ctypes.data, n, 2, 0)
    want = np.empty_like(x)
    want[:, golden["bitrev_ref_log2n_10"]] = x
    assert np.array_equal(out, want)
    E.lib().emu_bitrev(x.ctypes.data, x.ctypes.data, n, 2, 0)  # in place
    assert np.array_equal(x, want)


# ---------------------------------------------------------------------------
# team kernel (fft_team.h): a whole transform per "XCD", one HBM round trip.  The emulation runs EVERY workgroup of
# the launch concurrently (one host thread per GPU thread, sequentially consistent memory), so it checks the index
# algebra of the register/L2 hand-over AND the arrival/wait protocol between workgroups (a missing wait shows up as
# a data race => wrong spectrum).  Geometries are scaled down: teams of 4 or 8 workgroups of 8..128 threads.
# ---------------------------------------------------------------------------
TEAM_CASES = [
    # n, batch, dtype, log2seats, n_xcc, threads, lds_budget, tiles -> team size
    (4096, 5, np.complex64, 2, 2, 16, 16384, 4),    # 64 x 64, the whole "XCD" is one team of 4 (two phases wait in registers)
    (2048, 9, np.complex64, 2, 2, 16, 8192, 4),     # 32 x 64, TWO teams of 2 per "XCD" (sub-XCD teams: n = 2^16..2^19 on the device)
    (1024, 11, np.complex64, 3, 2, 8, 4096, 4),     # 32 x 32, four teams of 2 per "XCD", ragged batch over 8 teams
    (2048, 4, np.complex64, 2, 2, 16, 8192, 2),     # two tiles per workgroup: both phases handed over during the column step
    (1024, 7, np.complex64, 3, 2, 8, 4096, 1),      # one tile, one phase, teams of 8
    (2048, 5, np.complex128, 2, 3, 16, 16384, 4),   # fp64, three "XCDs"
    (1024, 5, np.complex128, 2, 2, 16, 8192, 4),    # fp64, teams of 2
]


@pytest.mark.parametrize("plain", [False, True])
@pytest.mark.parametrize("n,batch,dtype,log2seats,n_xcc,threads,lds,tiles", TEAM_CASES)
def test_team_kernel(n, batch, dtype, log2seats, n_xcc, threads, lds, tiles, plain, monkeypatch):
    """plain=False: team_defer_kernel where it applies (four tiles: the shipped default; the last row phase of a transform
    runs after the next one's column step), plain=True: team_fft_kernel."""
    if plain:
        monkeypatch.setenv("FFT_EMU_TEAM_PLAIN", "1")
    x = O.gen_lcg(n, 1, batch).astype(dtype)
    for d in (-1, 1):
        for inplace in (False, True):
            y, info = E.emu_fft_team(x, d, log2seats=log2seats, n_xcc=n_xcc, threads=threads, lds_budget=lds,
                                     inplace=inplace, tiles=tiles)
            assert info[0] // 100 == tiles, "team kernel was not planned"
            assert rel(y, oracle(x, d)) < TEAM_TOL[dtype], (n, d, inplace)


def test_wide_row_kernel(monkeypatch):
    """wide_row_kernel (fft_wide_row.h; the device runs it at n = 8192 fp32: 512 threads x 16 values) in its emulated shape: n = 512,
    32 threads, radix-16 x 16 x 2, one row per workgroup step with the next row's LDS-DMA issued under the stages; ragged batches over
    the 3 emulated CUs, both directions, in place and out of place."""
    monkeypatch.setenv("FFT_EMU_WIDE", "1")
    for batch in (1, 3, 7):
        x = O.gen_lcg(512, 37, batch).astype(np.complex64)
        for d in (-1, 1):
            for inplace in (False, True):
                y, info = E.emu_fft(x, d, inplace=inplace)
                assert info[6] & 16, "wide_row_kernel was not planned"
                assert rel(y, oracle(x, d)) < TEAM_TOL[np.complex64], (batch, d, inplace)


QUAD_CASES = [  # n, batch, log2seats, "XCDs", threads, LDS bytes
    (4096, 2, 2, 2, 64, 8192),    # E = 4, M = 4 x 4, teams of 4 seats of 64 threads (NC = 16): every seat in a row block of its own
    (4096, 9, 2, 1, 64, 8192),    # ragged batch on one "XCD"
    (4096, 16, 2, 3, 64, 8192),
    (1024, 5, 1, 2, 32, 4096),    # M = 4 x 2 (two radix-2 butterflies per thread in stage 2), teams of TWO: a seat's rows span two row blocks
    (1024, 7, 2, 2, 16, 2048),    # the same transform on teams of 4 (NC = 8: two columns per class)
    (1024, 4, 2, 1, 32, 4096),    # two teams of 2 on one "XCD"
    (2048, 5, 1, 2, 64, 8192),    # n = 64 x 32 (the device's 2^19 = 1024 x 512 and 2^17 = 512 x 256): column step 4 x 4, row step 4 x 2; teams of 2, two window slots
    (2048, 7, 2, 2, 32, 4096),    # ... teams of 4 (NC = 8 columns, NR = 16 rows per seat), one window slot
    (2048, 6, 2, 1, 64, 8192),    # two teams of 2 on one "XCD"
    (1024, 9, 2, 2, 64, 8192),    # a "team" of ONE (the device's n = 2^15 = 256 x 128): the seat's rows span all four row blocks
]


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
@pytest.mark.parametrize("n,batch,log2seats,n_xcc,threads,lds", QUAD_CASES)
def test_team_quad_kernel(n, batch, log2seats, n_xcc, threads, lds, dtype, monkeypatch):
    """team_quad_kernel (fft_team_quad.h) in its emulated shapes: E = 4 values per thread and chunk, the length-L/4 transforms as
    4 x 4 (the device's n = 2^20 is 16 x 16) and as 4 x 2 (the device's 16 x 8 at 2^18 and 16 x 4 at 2^16: several radix-R2
    butterflies per thread in stage 2, adjacent rows in ONE thread at the hand-over), teams of 4 and of 2 seats, ragged batches,
    both directions, in place and out of place.  The device shapes are the same source with other constants;
    tests/test_gpu_parity.py covers them."""
    monkeypatch.setenv("FFT_EMU_TEAM_QUAD", "1")
    x = O.gen_lcg(n, 23, batch).astype(dtype)
    if dtype == np.complex128:
        lds *= 2  # (fp64: one value per 16-byte access -- the same shapes with images twice the bytes; the device's n = 2^14 ... 2^16)
    for d in (-1, 1):
        for inplace in (False, True):
            y, info = E.emu_fft_team(x, d, log2seats=log2seats, n_xcc=n_xcc, threads=threads, lds_budget=lds, inplace=inplace)
            assert info[0] // 100 == 4 and info[6] & 8, "team_quad_kernel was not planned"
            assert info[5] == 1, "status / fallback / timeout counters: %d" % info[5]
            assert rel(y, oracle(x, d)) < TEAM_TOL[dtype], (n, batch, d, inplace)


def test_team_quad_kernel_with_deferred_result_stores(tmp_path):
    """Round 4: on teams of 32 (the device's n = 2^20) a transform's result stores go out in four parts from the column step of the team's NEXT
    transform (QUAD_DEFER_STORES; loop-carried results, the last transform's in an epilogue, dynamic claims, in place).  The emulated shapes
    have small teams, where the schedule is off by default: this test builds the emulation with the schedule forced on and runs the quad cases
    in a process of its own (the library is chosen when it is first loaded)."""
    import subprocess
    import sys
    so = str(tmp_path / "libfft_emu_defer.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-DFFT_EMU", "-DFFT_EXPERIMENTS", "-DQUAD_DEFER_STORES=1", "-fPIC", "-shared", "-pthread",
                    "-I" + E.CSRC, os.path.join(E.EMU_DIR, "emu_fft.cpp"), "-o", so], check=True)
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import emu_lib as E, oracle_lib as O\n"
        "import os; os.environ['FFT_EMU_TEAM_QUAD'] = '1'\n"
        "for n, batch, log2seats, n_xcc, threads, lds in %r:\n"
        "    x = O.gen_lcg(n, 23, batch).astype(np.complex64)\n"
        "    ref = np.fft.fft(x.astype(np.complex128), axis=1)\n"
        "    for dyn in ('0', '1'):\n"
        "        os.environ['FFT_HIP_TEAM_DYNAMIC'] = dyn\n"
        "        for inplace in (False, True):\n"
        "            y, info = E.emu_fft_team(x, -1, log2seats=log2seats, n_xcc=n_xcc, threads=threads, lds_budget=lds, inplace=inplace)\n"
        "            assert info[6] & 8 and info[5] == 1, (n, info[5])\n"
        "            err = np.linalg.norm(y - ref) / np.linalg.norm(ref)\n"
        "            assert err < 2e-6, (n, batch, dyn, inplace, err)\n"
        "print('ok')\n" % (os.path.dirname(os.path.abspath(__file__)), QUAD_CASES))
    env = dict(os.environ, FFT_EMU_SO=so)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_team_quad_kernel_static_and_dynamic_split_of_the_batch(monkeypatch):
    """team_quad_kernel takes its transforms either by the static rule team + it * n_teams or -- the default -- by claiming the next
    unclaimed one from a device-wide counter (the first seat claims, the team learns the index through its L2 line).  Both give the
    same spectra, bit for bit; ragged batches, more teams than transforms, one team per "XCD" and two."""
    monkeypatch.setenv("FFT_EMU_TEAM_QUAD", "1")
    for n, batch, log2seats, n_xcc, threads, lds in ((4096, 11, 2, 3, 64, 8192), (2048, 9, 2, 2, 32, 4096), (1024, 3, 2, 2, 16, 2048), (1024, 13, 2, 1, 32, 4096)):
        x = O.gen_lcg(n, 29, batch).astype(np.complex64)
        res = []
        for dyn in ("0", "1"):
            monkeypatch.setenv("FFT_HIP_TEAM_DYNAMIC", dyn)
            y, info = E.emu_fft_team(x, -1, log2seats=log2seats, n_xcc=n_xcc, threads=threads, lds_budget=lds)
            assert info[6] & 8 and info[5] == 1, (n, dyn, info[5])
            assert rel(y, oracle(x, -1)) < TEAM_TOL[np.complex64], (n, batch, dyn)
            res.append(y)
        assert np.array_equal(res[0], res[1])


@pytest.mark.parametrize("n,batch,log2seats,n_xcc,threads,lds", [
    (4096, 9, 2, 2, 64, 8192),    # teams of 4, ragged batch, two "XCDs"
    (2048, 5, 1, 2, 64, 8192),    # teams of 2 (a seat's rows span two row blocks)
    (2048, 7, 2, 2, 32, 4096),    # teams of 4, several butterflies per thread in stage 2
    (1024, 7, 2, 2, 16, 2048),
])
def test_team_quad_kernel_pair_protocol(n, batch, log2seats, n_xcc, threads, lds, monkeypatch):
    """Round 4: the exchange with ONE image per seat in the window and per-seat counters instead of the team's (SLOTS = 3, fft_team_quad.h
    `pair_guard` / `pair_signal` / `pair_wait`; the device's schedule at n = 2^20 and 2^19, profiles/r4_ab_pair_protocol_sizes.txt).  Here a sender unit is a host thread: every thread signals and guards for itself.  Static and
    dynamic split of the batch (the next transform reaches the team through the tagged 8-byte word), both directions, in place; and a
    member that never arrives ends in a TIMEOUT that the host repairs, with no workgroup stuck at a barrier."""
    monkeypatch.setenv("FFT_EMU_TEAM_QUAD", "1")
    monkeypatch.setenv("FFT_EMU_QUAD_SLOTS", "3")
    x = O.gen_lcg(n, 41, batch).astype(np.complex64)
    res = []
    for dyn in ("0", "1"):
        monkeypatch.setenv("FFT_HIP_TEAM_DYNAMIC", dyn)
        for d in (-1, 1):
            for inplace in (False, True):
                y, info = E.emu_fft_team(x, d, log2seats=log2seats, n_xcc=n_xcc, threads=threads, lds_budget=lds, inplace=inplace)
                assert info[0] // 100 == 4 and info[6] & 8 and info[5] == 1, (n, dyn, info[5])
                assert rel(y, oracle(x, d)) < TEAM_TOL[np.complex64], (n, batch, dyn, d, inplace)
                if d == -1 and not inplace:
                    res.append(y)
    assert np.array_equal(res[0], res[1])
    if n == 4096:
        monkeypatch.setenv("FFT_EMU_DROP_BLOCK", "1")
        monkeypatch.setenv("FFT_EMU_TEAM_TIMEOUT_MS", "300")
        monkeypatch.setenv("FFT_EMU_RECOVER", "1")
        y, info = E.emu_fft_team(x, -1, log2seats=log2seats, n_xcc=n_xcc, threads=threads, lds_budget=lds)
        assert info[3] == 1100, "timeout seen, nothing lost: %d" % info[3]
        assert rel(y, oracle(x, -1)) < TEAM_TOL[np.complex64]


@pytest.mark.parametrize("quad", [False, True])
def test_team_kernel_timeout_is_repaired_on_the_multi_pass_schedule(quad, monkeypatch):
    """VERDICT r2 item 6.  A member of a formed team never arrives (FFT_EMU_DROP_BLOCK leaves right after formation): the team's
    waits run into their bound, the kernel reports TIMEOUT and ends.  What the host does next (HIP: team_status_of inside
    fft_gpu_plan_sync; here FFT_EMU_RECOVER): the out-of-place execute is REPEATED on the multi-pass schedule and the result is
    correct.  Round 4: an IN-PLACE execute of a plan whose team kernel has not yet been seen to end well
    runs from a staged copy of its input and is repeated from that copy (fft_gpu_execute has no failure mode, reference
    include/fft_gpu.h:102); a ping-pong A -> B, B -> A between two syncs is NOT repeated (the later launch has overwritten A, and B was
    written by a void launch): both executes are reported lost instead of being "repaired" from garbage."""
    monkeypatch.setenv("FFT_EMU_DROP_BLOCK", "1")
    monkeypatch.setenv("FFT_EMU_TEAM_TIMEOUT_MS", "300")
    monkeypatch.setenv("FFT_EMU_RECOVER", "1")
    if quad:
        monkeypatch.setenv("FFT_EMU_TEAM_QUAD", "1")
        geo = dict(log2seats=2, n_xcc=2, threads=64, lds_budget=8192)
    else:
        geo = dict(log2seats=2, n_xcc=2, threads=16, lds_budget=16384)
    x = O.gen_lcg(4096, 31, 6).astype(np.complex64)
    y, info = E.emu_fft_team(x, -1, **geo)
    assert info[0] // 100 == 4 and bool(info[6] & 8) == quad
    assert info[3] == 1100, "timeout seen, nothing lost: %d" % info[3]
    assert rel(y, oracle(x, -1)) < TEAM_TOL[np.complex64]
    y, info = E.emu_fft_team(x, -1, inplace=True, **geo)
    assert info[3] == 1100, "timeout seen, the in-place execute repeated from its staged input: %d" % info[3]
    assert rel(y, oracle(x, -1)) < TEAM_TOL[np.complex64]
    monkeypatch.setenv("FFT_EMU_PINGPONG", "1")
    y, info = E.emu_fft_team(x, -1, **geo)
    assert info[3] == 1102, "timeout seen, neither execute of the ping-pong may be repeated: %d" % info[3]


def test_team_quad_kernel_falls_back_when_teams_cannot_form(monkeypatch):
    """Workgroup 0 reports the wrong XCD: the quad kernel must leave before touching anything (status 1) and the multi-pass
    plan queued behind it must produce the result."""
    monkeypatch.setenv("FFT_EMU_TEAM_QUAD", "1")
    x = O.gen_lcg(4096, 29, 4).astype(np.complex64)
    y, info = E.emu_fft_team(x, -1, log2seats=2, n_xcc=2, threads=64, lds_budget=8192, skew=True)
    assert info[6] & 8 and info[5] % 10 == 2, info  # status NO_TEAMS
    assert rel(y, oracle(x, -1)) < TEAM_TOL[np.complex64]


@pytest.mark.parametrize("n,batch,log2seats,n_xcc,threads,lds,l1", [
    (4096, 5, 2, 2, 16, 16384, None),     # 64 x 64, CB = 4: a team of 4; block parity = a bit of the row-in-thread index
    (2048, 9, 2, 2, 16, 8192, None),      # 32 x 64, two teams of 2 per "XCD"
    (1024, 11, 3, 2, 8, 4096, None),      # 32 x 32, teams of 2, ragged batch
    (1 << 14, 5, 2, 2, 64, 65536, 5),     # 32 x 512: three-stage rows
    (1 << 16, 3, 3, 2, 128, 1 << 17, 6),  # 64 x 1024: the production stage sequence 16 x 16 x 4 with the bank swizzle
    (1 << 14, 6, 2, 2, 64, 65536, 8),     # 256 x 64: TPCA = 16 > CB = 4 ... every kind of slot <-> phase relation
])
@pytest.mark.parametrize("nodefer", [False, True, "alll2"])
def test_team_kernel_paired_row_tiles(n, batch, log2seats, n_xcc, threads, lds, l1, nodefer, monkeypatch):
    """team_defer_kernel PAIR: a seat's row tiles of phases (0, 1) and (2, 3) are adjacent blocks of rows, the even phase's
    results wait in registers (phase 2's across the next transform's column step) and both are written as double-width
    segments by the odd phase."""
    monkeypatch.setenv("FFT_EMU_TEAM_PAIR", "1")
    if nodefer:  # NODEFER: phase 3 right after phase 2, handed over into S1: two live windows, five arrivals
        monkeypatch.setenv("FFT_EMU_TEAM_NODEFER", "1")
    if nodefer == "alll2":  # ALLL2: every phase handed over during the column step, four windows, two arrivals
        monkeypatch.setenv("FFT_EMU_TEAM_ALLL2", "1")
    if l1:
        monkeypatch.setenv("FFT_HIP_TEAM_L1", str(l1))
    x = O.gen_lcg(n, 17, batch).astype(np.complex64)
    for d in (-1, 1):
        for inplace in (False, True):
            y, info = E.emu_fft_team(x, d, log2seats=log2seats, n_xcc=n_xcc, threads=threads, lds_budget=lds, inplace=inplace)
            assert info[0] // 100 == 4 and info[6] & 4, "the paired kernel was not planned"
            assert rel(y, oracle(x, d)) < TEAM_TOL[np.complex64], (n, d, inplace)


def test_team_kernel_fallback_when_teams_cannot_form():
    """Workgroup 0 reports the wrong XCD: no launch yields full teams, the kernel must give up before touching
    anything and the two-pass schedule queued behind it must produce the result (also in place)."""
    x = O.gen_lcg(4096, 2, 3).astype(np.complex64)
    for inplace in (False, True):
        y, info = E.emu_fft_team(x, -1, log2seats=2, n_xcc=2, threads=16, lds_budget=16384, inplace=inplace, skew=True)
        assert info[0] >= 100
        assert rel(y, oracle(x, -1)) < TEAM_TOL[np.complex64]


def test_team_kernel_status_words():
    """info[5] = 1 + STATUS + 10 * sticky fallbacks + 100 * sticky timeouts: a healthy launch reports 1, a launch that
    could not form its teams 1 + 1 + 10 (exactly one workgroup counts the fallback)."""
    x = O.gen_lcg(4096, 2, 3).astype(np.complex64)
    _, info = E.emu_fft_team(x, -1, log2seats=2, n_xcc=2, threads=16, lds_budget=16384)
    assert info[5] == 1
    _, info = E.emu_fft_team(x, -1, log2seats=2, n_xcc=2, threads=16, lds_budget=16384, skew=True)
    assert info[5] == 12


@pytest.mark.parametrize("plain", [False, True])
@pytest.mark.parametrize("late_block", [0, 5])
def test_team_kernel_late_workgroup_cannot_split_the_launch(late_block, plain, monkeypatch):
    """One workgroup becomes resident only after the formation timeout (a device shared with somebody else's kernel).
    The decision must be the same for every workgroup: the ones that waited poison the registration word and leave,
    the late one finds the poison and leaves too -- nobody runs as a team member, nothing is touched (the in-place
    input survives), status = NO_TEAMS, and the multi-pass plan queued behind the kernel produces the result."""
    if plain:
        monkeypatch.setenv("FFT_EMU_TEAM_PLAIN", "1")
    monkeypatch.setenv("FFT_EMU_FORM_TIMEOUT_MS", "30")
    monkeypatch.setenv("FFT_EMU_LATE_BLOCK", str(late_block))
    monkeypatch.setenv("FFT_EMU_LATE_MS", "400")
    x = O.gen_lcg(4096, 4, 5).astype(np.complex64)
    for inplace in (False, True):
        y, info = E.emu_fft_team(x, -1, log2seats=2, n_xcc=2, threads=16, lds_budget=16384, inplace=inplace)
        assert info[0] >= 400
        assert info[5] == 12, "status must be NO_TEAMS, counted once, no timeout"
        assert rel(y, oracle(x, -1)) < TEAM_TOL[np.complex64]


def test_team_kernel_hung_member_is_reported_even_after_later_launches(monkeypatch):
    """A member of a formed team never arrives (emulation hook): its team's waits run into their bound, the status word
    says TIMEOUT -- written by compare-and-swap, so it can never overwrite NO_TEAMS -- and the STICKY counter keeps the
    report although the control block is zeroed by the next launch (round-1 ADVICE: a timeout of an earlier queued
    execute used to be erased).  info[5] = 1 + status + 10 * sticky fallbacks + 100 * sticky timeouts."""
    monkeypatch.setenv("FFT_EMU_TEAM_TIMEOUT_MS", "300")
    monkeypatch.setenv("FFT_EMU_DROP_BLOCK", "3")
    x = O.gen_lcg(4096, 4, 5).astype(np.complex64)
    _, info = E.emu_fft_team(x, -1, log2seats=2, n_xcc=2, threads=16, lds_budget=16384)
    status, timeouts = (info[5] - 1) % 10, info[5] // 100
    assert status == 2, info[5]
    assert 1 <= timeouts <= 3, "every surviving member of the hung team (4 seats, one dropped) reports once"


def test_team_kernel_slow_start_inside_the_timeout_still_forms_teams(monkeypatch):
    """A workgroup that registers late but inside the formation timeout: the launch proceeds as a team."""
    monkeypatch.setenv("FFT_EMU_FORM_TIMEOUT_MS", "5000")
    monkeypatch.setenv("FFT_EMU_LATE_BLOCK", "3")
    monkeypatch.setenv("FFT_EMU_LATE_MS", "200")
    x = O.gen_lcg(4096, 4, 5).astype(np.complex64)
    y, info = E.emu_fft_team(x, -1, log2seats=2, n_xcc=2, threads=16, lds_budget=16384)
    assert info[5] == 1
    assert rel(y, oracle(x, -1)) < TEAM_TOL[np.complex64]


@pytest.mark.parametrize("dtype,threads", [(np.complex64, 64), (np.complex128, 128)])
def test_team_kernel_three_stage_rows(dtype, threads, monkeypatch):
    """32 x 512 split: the row FFTs have three stages, so the hand-over of phase ph+2 and the next tile's DMA are
    issued from two different mid-stage hooks (the shape of the production geometry, 1024 = 16 x 16 x 4)."""
    monkeypatch.setenv("FFT_HIP_TEAM_L1", "5")
    x = O.gen_lcg(1 << 14, 4, 5).astype(dtype)
    for d, inplace in ((-1, False), (1, True)):
        y, info = E.emu_fft_team(x, d, log2seats=2, n_xcc=2, threads=threads, lds_budget=65536, inplace=inplace)
        assert info[0] >= 400
        assert rel(y, oracle(x, d)) < TEAM_TOL[dtype]


def test_team_kernel_swizzled_last_stage(monkeypatch):
    """64 x 1024 split: the row FFTs are 1024 = 16 x 16 x 4, the production stage sequence, whose last exchange goes
    through the LDS bank swizzle (stage_swizzle in fft_kernels.h)."""
    monkeypatch.setenv("FFT_HIP_TEAM_L1", "6")
    x = O.gen_lcg(1 << 16, 2, 3).astype(np.complex64)
    for d, inplace in ((-1, False), (1, True)):
        y, info = E.emu_fft_team(x, d, log2seats=3, n_xcc=2, threads=128, lds_budget=1 << 17, inplace=inplace)
        assert info[0] >= 400
        assert rel(y, oracle(x, d)) < TEAM_TOL[np.complex64]


@pytest.mark.parametrize("n,batch,dtype,log2seats,threads,lds", [(4096, 5, np.complex64, 2, 16, 16384), (1 << 14, 3, np.complex64, 2, 64, 65536),
                                                                  (2048, 5, np.complex128, 2, 16, 16384)])
def test_team_kernel_even_odd_row_split(n, batch, dtype, log2seats, threads, lds, monkeypatch):
    """ASPLIT: the column step runs on half-height, double-width tiles (even rows, then odd rows: 128-byte row segments
    on the device) joined by a radix-2 butterfly in registers; the row step is unchanged."""
    monkeypatch.setenv("FFT_EMU_TEAM_ASPLIT", "1")
    x = O.gen_lcg(n, 7, batch).astype(dtype)
    for d, inplace in ((-1, False), (1, True)):
        y, info = E.emu_fft_team(x, d, log2seats=log2seats, n_xcc=2, threads=threads, lds_budget=lds, inplace=inplace)
        assert info[0] >= 400 and info[6] == 1, "the split column step was not planned"
        assert rel(y, oracle(x, d)) < TEAM_TOL[dtype]


def test_bluestein_chained_over_several_launch_groups(monkeypatch):
    """m = 4096 fp64 = 64 x 64 chained, 40 transforms in launch groups of 16 (FFT_HIP_CHUNK_MB = 1)."""
    monkeypatch.setenv("FFT_HIP_CHUNK_MB", "1")
    n, batch = 2000, 40
    x = O.gen_lcg(n, n, batch).astype(np.complex128)
    y, info = E.emu_fft(x, -1, lds_budget=40000)
    assert info[0] == 12 and info[4] == 2 and info[7] == 16, info[:8]
    assert rel(y, O.oracle_fft(x, -1, "bluestein")) < 1e-12


@pytest.mark.parametrize("n,batch", [(2048, 3), (4096, 2), (4096, 5)])
def test_fp64_single_pass_without_staging_emulated(n, batch):
    """fp64 n >= 2048: the single-pass plan loads and stores its eight elements per thread directly (no LDS staging; the tile's
    columns are whole transforms, TileParams::col_stride = n), batch not a multiple of the tile's column count, in place too."""
    x = O.gen_lcg(n, n + 1, batch).astype(np.complex128)
    for d in (-1, 1):
        for inplace in (False, True):
            y, info = E.emu_fft(x, d, lds_budget=160 * 1024, inplace=inplace)
            assert info[0] == 1 and info[1] == int(np.log2(n)), info[:4]
            assert rel(y, O.oracle_fft(x, d, "exact")) < 1e-14, (n, d, inplace)


@pytest.mark.parametrize("n,dtype,lds", [(1009, np.complex128, 4096), (1009, np.complex64, 4096), (3000, np.complex64, 12000)])
def test_bluestein_chained_through_the_mirrored_split(n, dtype, lds, monkeypatch):
    """m = 2^11 / 2^13 in two passes with UNEQUAL factors: the inverse transform runs on the mirrored split (Pow2Plan::mirror),
    whose first pass shares the forward transform's last tile, so the pair still chains; several launch groups."""
    monkeypatch.setenv("FFT_HIP_CHUNK_MB", "1")
    batch = 70
    x = O.gen_lcg(n, n + 3, batch).astype(dtype)
    tol = 1e-12 if dtype == np.complex128 else 2e-5
    for d in (-1, 1):
        y, info = E.emu_fft(x, d, lds_budget=lds)
        assert info[0] == 12 and info[4] == 2 and info[7] < batch, info[:8]
        assert rel(y, O.oracle_fft(x.astype(np.complex128), d, "bluestein")) < tol, (n, d)
