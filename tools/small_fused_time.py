"""Median ms of small (single-pass) Bluestein and fused plans: python tools/small_fused_time.py label"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
import fftlib  # noqa: E402


def med(fn, sync):
    fn(); sync()
    ts = []
    for _ in range(9):
        t = time.perf_counter(); fn(); fn(); sync(); ts.append((time.perf_counter() - t) * 0.5e3)
    return float(np.median(ts))


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else ""
    fftlib.init()
    out = []
    for n, batch, dt in ((300, 262144, np.complex64), (1000, 65536, np.complex64), (2000, 32768, np.complex64), (1000, 32768, np.complex128), (2000, 16384, np.complex128)):
        nb = n * batch * np.dtype(dt).itemsize
        a, b = fftlib.DeviceBuffer(nb), fftlib.DeviceBuffer(nb)
        p = fftlib.Plan(n, batch, -1, dt)
        ms = med(lambda: p.execute_ptr(a.ptr, b.ptr), p.sync)
        out.append("blu %d x %d %s %.3f ms %.1f" % (n, batch, np.dtype(dt).name[-3:], ms, n * batch / ms / 1e6))
        p.destroy(); a.free(); b.free()
    for kind, nx, batch, dt in (("psd", 1024, 65536, np.complex64), ("psd", 4096, 16384, np.complex64), ("circ", 2048, 32768, np.complex64), ("circ", 4096, 8192, np.complex128),
                                ("autocorr", 1000, 32768, np.complex64)):
        esz = np.dtype(dt).itemsize
        h = np.ones(nx, dtype=dt) if kind == "circ" else None
        p = fftlib.ExtPlan.fused(kind, nx, batch, h, dt)
        x = fftlib.DeviceBuffer(nx * batch * esz)
        o = fftlib.DeviceBuffer(p.out_len * batch * esz)
        ms = med(lambda: p.execute_fused(x.ptr, None, o.ptr, 1.0), p.sync)
        out.append("%s %d x %d %s %.3f ms" % (kind, nx, batch, np.dtype(dt).name[-3:], ms))
        p.destroy(); x.free(); o.free()
    print(label, "|", " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
