"""Median ms / Gpoint/s of plain forward plans for a list of sizes with the library FFT_LIB_PATH names (A/B of builds: run once
per build, twice over).  python tools/time_sizes.py label [team=0|1] n:batch:f32|f64 ..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
import fftlib  # noqa: E402


def main():
    label = sys.argv[1]
    args = sys.argv[2:]
    team = 1
    if args and args[0].startswith("team="):
        team = int(args[0][5:]); args = args[1:]
    fftlib.init()
    fftlib.set_policy(team=team, min_batch=0, chunk_mb=0)
    out = []
    for a in args:
        n, batch, prec = a.split(":")
        n, batch = int(n), int(batch)
        dt = np.complex64 if prec == "f32" else np.complex128
        nbytes = n * batch * np.dtype(dt).itemsize
        x, y = fftlib.DeviceBuffer(nbytes), fftlib.DeviceBuffer(nbytes)
        plan = fftlib.Plan(n, batch, -1, dt)
        plan.execute_ptr(x.ptr, y.ptr); plan.sync()
        ts = []
        for _ in range(9):
            t = time.perf_counter(); plan.execute_ptr(x.ptr, y.ptr); plan.execute_ptr(x.ptr, y.ptr); plan.sync()
            ts.append((time.perf_counter() - t) * 0.5e3)
        ms = float(np.median(ts))
        out.append("%s %.3f ms %.1f" % (a, ms, n * batch / ms / 1e6))
        plan.destroy(); x.free(); y.free()
    print(label, "team=%d |" % team, " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
