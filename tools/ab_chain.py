"""A/B of the chained pass (forward transform's last pass + inverse transform's first pass as one kernel, csrc/fft_kernels_chain.h)
against the same plan with FFT_GPU_OPT_NO_CHAIN: Bluestein sizes and fused consumers.  python tools/ab_chain.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
import fftlib  # noqa: E402


def time_ab(plan, fn, reps=12):
    """median ms of fn() with the chained kernel wherever possible (option value 2) and without (1), measured alternately so
    that neither leg owns the warm-up"""
    ts = {2: [], 1: []}
    for mode in (2, 1):
        plan.set_option(fftlib.OPT_NO_CHAIN, mode)
        fn(); plan.sync()
    for _ in range(reps):
        for mode in (2, 1):
            plan.set_option(fftlib.OPT_NO_CHAIN, mode)
            t = time.perf_counter(); fn(); fn(); plan.sync(); ts[mode].append((time.perf_counter() - t) * 0.5e3)
    return float(np.median(ts[2])), float(np.median(ts[1]))


def main():
    fftlib.init()
    print("bluestein: n batch dtype | chained ms (Gpt/s) | two kernels ms (Gpt/s)")
    for n, batch, dt in ((1000003, 64, np.complex128), (1000003, 64, np.complex64), (100003, 512, np.complex128), (100003, 512, np.complex64),
                         (10007, 4096, np.complex64), (10007, 2048, np.complex128), (50021, 1024, np.complex64), (50021, 512, np.complex128), (30011, 2048, np.complex64), (30011, 1024, np.complex128), (250007, 256, np.complex64),
                         (1500007, 32, np.complex128), (1500007, 32, np.complex64), (3000017, 16, np.complex128), (6000011, 8, np.complex128)):
        x = (np.random.default_rng(1).standard_normal((batch, n)) + 0j).astype(dt)
        buf = fftlib.DeviceBuffer(x.nbytes); buf.upload(x)
        out = fftlib.DeviceBuffer(x.nbytes)
        plan = fftlib.Plan(n, batch, -1, dt)
        plan.set_option(fftlib.OPT_NO_CHAIN, 2)
        fused = plan.info().fused
        row = ["%.3f (%.1f)" % (med, n * batch / med / 1e6) for med in time_ab(plan, lambda: plan.execute_ptr(buf.ptr, out.ptr))]
        print(n, batch, np.dtype(dt).name, "m=2^%d" % int(np.log2(plan.info().bluestein_m)), list(plan.info().factors), "chainable" if fused == 2 else "not chainable", "|", " | ".join(row), flush=True)
        plan.destroy(); buf.free(); out.free()
    print("fused: kind nx batch dtype | chained ms | two kernels ms")
    for kind, nx, nh, batch, dt in (("conv", 1 << 19, 1000, 64, np.complex64), ("circ", 1 << 20, 0, 64, np.complex64), ("circ", 1 << 21, 0, 32, np.complex128),
                                    ("autocorr", 1 << 20, 0, 32, np.complex128), ("circ", 1 << 16, 0, 1024, np.complex64), ("circ", 1 << 16, 0, 512, np.complex128), ("circ", 1 << 18, 0, 256, np.complex64),
                                    ("circ", 1 << 18, 0, 128, np.complex128), ("circ", 1 << 22, 0, 16, np.complex64), ("circ", 1 << 17, 0, 512, np.complex64), ("circ", 1 << 19, 0, 128, np.complex64), ("circ", 1 << 15, 0, 2048, np.complex128),
                                    ("circ", 1 << 14, 0, 4096, np.complex64), ("circ", 4096, 0, 16384, np.complex64)):
        rng = np.random.default_rng(2)
        x = (rng.standard_normal((batch, nx)) + 0j).astype(dt)
        h = (rng.standard_normal(nh if kind == "conv" else nx) + 0j).astype(dt) if kind in ("conv", "circ") else None
        plan = fftlib.ExtPlan.fused(kind, nx, batch, h, dt)
        buf = fftlib.DeviceBuffer(x.nbytes); buf.upload(x)
        out = fftlib.DeviceBuffer(batch * plan.out_len * np.dtype(dt).itemsize)
        plan.set_option(fftlib.OPT_NO_CHAIN, 2)
        fused = plan.info().fused
        row = ["%.3f" % med for med in time_ab(plan, lambda: plan.execute_fused(buf.ptr, None, out.ptr, 1.0))]
        print(kind, nx, batch, np.dtype(dt).name, list(plan.info().factors), "chainable" if fused == 2 else "not chainable", "|", " | ".join(row), flush=True)
        plan.destroy(); buf.free(); out.free()


if __name__ == "__main__":
    main()
