"""Which four-step split is fastest for the PLAIN multi-pass schedule (no team kernel)?  Times every two-pass split l1 + l2 = log2 n
(and a few three-pass ones) with FFT_HIP_FORCE_SPLIT on the experiments build, ~1 GiB of transforms per execute, and marks the
split the planner's cost model picks.  FFT_LIB_PATH must name libfft_mi355x_exp.so.  python tools/split_sweep.py [f32|f64] [log2n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
os.environ.setdefault("FFT_LIB_PATH", os.path.join(ROOT, "fft-implementation-in-c_amd", "libfft_mi355x_exp.so"))
import fftlib  # noqa: E402


def timed(plan, a, b, reps=7):
    plan.execute_ptr(a, b); plan.sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); plan.execute_ptr(a, b); plan.execute_ptr(a, b); plan.sync(); ts.append((time.perf_counter() - t) * 0.5e3)
    return float(np.median(ts))


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
    sizes = [int(v) for v in sys.argv[2:]] or [14, 15, 16, 17, 18, 19, 20, 21, 22]
    dt = np.complex64 if prec == "f32" else np.complex128
    fftlib.init()
    fftlib.set_policy(team=0, min_batch=0, chunk_mb=0)
    for log2n in sizes:
        n = 1 << log2n
        batch = max(1, (1 << 30) // (n * np.dtype(dt).itemsize))
        a = fftlib.DeviceBuffer(n * batch * np.dtype(dt).itemsize)
        b = fftlib.DeviceBuffer(n * batch * np.dtype(dt).itemsize)
        os.environ.pop("FFT_HIP_FORCE_SPLIT", None)
        plan = fftlib.Plan(n, batch, -1, dt)
        info = plan.info()
        auto = [int(np.log2(f)) for f in list(info.factors) if f]
        ms = timed(plan, a.ptr, b.ptr)
        print("n=2^%d %s batch %d | planner: %s %.3f ms %.1f Gpt/s" % (log2n, prec, batch, auto, ms, n * batch / ms / 1e6), flush=True)
        plan.destroy()
        cands = [(l1, log2n - l1) for l1 in range(5, log2n - 4)]
        if log2n >= 19:
            third = log2n // 3
            cands += [(third, log2n - 2 * third, third), (7, log2n - 14, 7), (8, log2n - 16, 8)]
        best = None
        for c in cands:
            if any(v < 4 or v > 12 for v in c):
                continue
            os.environ["FFT_HIP_FORCE_SPLIT"] = ",".join(str(v) for v in c)
            try:
                plan = fftlib.Plan(n, batch, -1, dt)
            except Exception:
                continue
            got = [int(np.log2(f)) for f in list(plan.info().factors) if f]
            if got != list(c):
                plan.destroy()
                continue
            ms = timed(plan, a.ptr, b.ptr)
            mark = " <- planner" if got == auto else ""
            print("    %-14s %.3f ms %.1f Gpt/s%s" % (got, ms, n * batch / ms / 1e6, mark), flush=True)
            if best is None or ms < best[1]:
                best = (got, ms)
            plan.destroy()
        if best:
            print("    best %s %.3f ms" % best)
        a.free(); b.free()


if __name__ == "__main__":
    main()
