"""The team kernel on a device it does not own and the per-device planner state (item 10).

A team launch needs every CU (one 160 KiB-LDS workgroup each).  When somebody else's kernel holds some CUs, the
workgroups that did get a CU wait at most 1 ms for the rest (team_form, csrc/fft_team.h), then ALL leave -- the ones
dispatched later find the registration word poisoned and leave at once -- and the multi-pass plan queued behind the
kernel does the work on whatever CUs are free.  Correct output, status 1, a few milliseconds; never the 0.2 s spin
(x 3 executes) of round 1, never a half-formed team."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPERS = os.path.join(ROOT, "tests", "gpu_helpers")


def _occ():
    so = os.path.join(HELPERS, "libocc.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", HELPERS], check=True)
    lib = C.CDLL(so)
    lib.occupy_start.argtypes = [C.c_int, C.c_int, C.c_int]
    return lib


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.complex128) - b) / np.linalg.norm(b))


@pytest.mark.gpu
@pytest.mark.parametrize("log2n,batch", [(20, 16), (18, 64)])
def test_team_plan_beside_a_foreign_kernel(gpu_lib, log2n, batch):
    import fftlib
    occ = _occ()
    n = 1 << log2n
    x = O.gen_lcg(n, 77, batch).astype(np.complex64)
    ref = O.oracle_fft(x[:2].astype(np.complex128), -1, "dit")
    buf = fftlib.DeviceBuffer(x.nbytes)
    fftlib.set_policy(team=2)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    assert plan.info().team_tiles == 4
    # warm: the team kernel alone forms its teams
    buf.upload(x)
    plan.execute_ptr(buf.ptr, buf.ptr)
    assert plan.team_status() == 0
    assert rel(buf.download(x.shape, np.complex64)[:2], ref) < 2e-6
    # 24 foreign workgroups, 100 KiB of LDS each (no room for a team workgroup beside them), for 60 ms
    buf.upload(x)
    assert occ.occupy_start(24, 100 * 1024, 60000) == 0
    time.sleep(0.002)  # let it become resident
    assert occ.occupy_busy() == 1
    t0 = time.perf_counter()
    plan.execute_ptr(buf.ptr, buf.ptr)
    st = plan.team_status()  # syncs the plan's stream
    wall_ms = (time.perf_counter() - t0) * 1e3
    still_busy = occ.occupy_busy()
    y = buf.download(x.shape, np.complex64)
    assert occ.occupy_wait() == 0
    assert still_busy == 1, "the foreign kernel must have outlived the execute for this test to mean anything"
    assert st == 1, "not every workgroup could be resident: the launch must have fallen back as a whole (status %d)" % st
    assert rel(y[:2], ref) < 2e-6 and not np.isnan(y).any()
    assert wall_ms < 20.0, "execute beside a foreign kernel took %.1f ms" % wall_ms
    # and afterwards the plan forms its teams again (one fallback does not switch the team kernel off)
    buf.upload(x)
    plan.execute_ptr(buf.ptr, buf.ptr)
    assert plan.team_status() == 0
    assert rel(buf.download(x.shape, np.complex64)[:2], ref) < 2e-6
    plan.destroy()
    buf.free()


@pytest.mark.gpu
def test_queued_executes_keep_an_earlier_fallback_visible(gpu_lib):
    """Several executes queued without a sync: the status word is the LAST launch's, the sticky counters cover all of
    them (an earlier execute's report used to be erased by the next execute's memset)."""
    import fftlib
    n, batch = 1 << 18, 32
    x = O.gen_lcg(n, 5, batch).astype(np.complex64)
    ref = O.oracle_fft(x[:1].astype(np.complex128), -1, "dit")
    a = fftlib.DeviceBuffer(x.nbytes)
    b = fftlib.DeviceBuffer(x.nbytes)
    a.upload(x)
    fftlib.set_policy(team=2)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    plan.set_option(fftlib.OPT_TEAM_FORCE_FALLBACK, 1)
    plan.execute_ptr(a.ptr, b.ptr)   # falls back
    plan.set_option(fftlib.OPT_TEAM_FORCE_FALLBACK, 0)
    plan.execute_ptr(a.ptr, b.ptr)   # team kernel
    assert plan.team_status() == 0   # the last launch formed its teams; one fallback of two launches is not a streak
    assert rel(b.download(x.shape, np.complex64)[:1], ref) < 2e-6
    assert plan.sync() == 0
    plan.destroy()
    a.free()
    b.free()


@pytest.mark.gpu
def test_every_device_gets_its_own_planner_state(gpu_lib):
    """fft_gpu_set_device(d) + per-device plans over fft_gpu_device_count() devices (1 on this pool, 8 on a node):
    device properties are read per device, plans remember their device, buffers live where they were allocated."""
    import fftlib
    lib = gpu_lib
    count = lib.fft_gpu_device_count()
    assert count >= 1
    n, batch = 1 << 16, 8
    x = O.gen_lcg(n, 9, batch).astype(np.complex64)
    ref = O.oracle_fft(x[:1].astype(np.complex128), -1, "dit")
    plans, bufs = [], []
    for d in range(count):
        assert lib.fft_gpu_set_device(d) == 0
        assert lib.fft_gpu_get_device_hip() == d
        assert b"CUs" in lib.fft_gpu_get_device_name()
        p = fftlib.Plan(n, batch, -1, np.complex64)
        assert p.info().device == d
        buf = fftlib.DeviceBuffer(x.nbytes)
        buf.upload(x)
        plans.append(p)
        bufs.append(buf)
    assert lib.fft_gpu_set_device(count) == -1  # out of range
    for d in range(count):  # launch all, then sync all: the batch-sharded pattern of SURVEY.md 8e in one process
        plans[d].execute_ptr(bufs[d].ptr, bufs[d].ptr)
    for d in range(count):
        assert plans[d].sync() == 0
        assert lib.fft_gpu_set_device(d) == 0
        assert rel(bufs[d].download(x.shape, np.complex64)[:1], ref) < 2e-6
    for p in plans:
        p.destroy()
    for b in bufs:
        b.free()
    assert lib.fft_gpu_set_device(0) == 0
