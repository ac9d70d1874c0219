"""Throughput of the plans built on the batched engine: 2D c2c, r2c / c2r, fused consumers.
Algorithmic bytes = input read once + output written once; GB/s against that and the fraction of the 8 TB/s spec roof.
python tools/ext_bench.py  (on an MI355X)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
import fftlib  # noqa: E402


def med_ms(fn, sync, reps=9):
    fn(); sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); fn(); sync(); ts.append((time.perf_counter() - t) * 0.5e3)
    return float(np.median(ts))


def line(label, ms, points, nbytes):
    print("%-58s %8.3f ms %8.1f Gpoint/s %8.0f GB/s alg (%4.1f %% of 8 TB/s)" % (label, ms, points / ms / 1e6, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80.0), flush=True)


def main():
    fftlib.init()
    for rows, cols, nm, dt in ((1024, 1024, 256, np.complex64), (4096, 4096, 16, np.complex64), (8192, 8192, 4, np.complex64), (512, 512, 1024, np.complex64),
                               (2048, 2048, 32, np.complex128), (1000, 1000, 128, np.complex64), (4096, 512, 128, np.complex64)):
        esz = np.dtype(dt).itemsize
        nbytes = rows * cols * nm * esz
        a, b = fftlib.DeviceBuffer(nbytes), fftlib.DeviceBuffer(nbytes)
        plan = fftlib.ExtPlan.fft2d(rows, cols, nm, -1, dt)
        ms = med_ms(lambda: plan.execute_ptr(a.ptr, b.ptr), plan.sync)
        line("2D c2c %d x %d x %d %s" % (rows, cols, nm, np.dtype(dt).name), ms, rows * cols * nm, 2 * nbytes)
        plan.destroy(); a.free(); b.free()
    for n, batch, dt in ((1 << 20, 512, np.float32), (1 << 16, 8192, np.float32), (4096, 131072, np.float32), (1 << 20, 256, np.float64), (1000000, 64, np.float32)):
        rsz = np.dtype(dt).itemsize
        nin, nout = n * batch * rsz, (n // 2 + 1) * batch * 2 * rsz
        a, b = fftlib.DeviceBuffer(nin), fftlib.DeviceBuffer(nout)
        plan = fftlib.ExtPlan.r2c(n, batch, dt)
        ms = med_ms(lambda: plan.execute_ptr(a.ptr, b.ptr), plan.sync)
        line("r2c n=%d x %d %s" % (n, batch, np.dtype(dt).name), ms, n * batch, nin + nout)
        plan.destroy()
        plan = fftlib.ExtPlan.c2r(n, batch, dt)
        ms = med_ms(lambda: plan.execute_ptr(b.ptr, a.ptr), plan.sync)
        line("c2r n=%d x %d %s" % (n, batch, np.dtype(dt).name), ms, n * batch, nin + nout)
        plan.destroy(); a.free(); b.free()
    rng = np.random.default_rng(3)
    for kind, nx, nh, batch, dt in (("conv", 1 << 19, 1000, 256, np.complex64), ("circ", 1 << 20, 0, 256, np.complex64), ("circ", 1 << 16, 0, 4096, np.complex64),
                                    ("autocorr", 1 << 19, 0, 256, np.complex64), ("xcorr", 1 << 19, 0, 128, np.complex64), ("psd", 1 << 20, 0, 256, np.complex64),
                                    ("circ", 1 << 20, 0, 128, np.complex128)):
        esz = np.dtype(dt).itemsize
        h = (rng.standard_normal(nh if kind == "conv" else nx) + 0j).astype(dt) if kind in ("conv", "circ") else None
        plan = fftlib.ExtPlan.fused(kind, nx, batch, h, dt)
        osz = (esz // 2) if kind == "psd" else esz
        nin, nout = nx * batch * esz * (2 if kind == "xcorr" else 1), plan.out_len * batch * osz
        x, y = fftlib.DeviceBuffer(nx * batch * esz), fftlib.DeviceBuffer(nx * batch * esz) if kind == "xcorr" else None
        out = fftlib.DeviceBuffer(nout)
        ms = med_ms(lambda: plan.execute_fused(x.ptr, y.ptr if y else None, out.ptr, 1.0), plan.sync)
        line("fused %s nx=%d x %d %s (fused=%d)" % (kind, nx, batch, np.dtype(dt).name, plan.info().fused), ms, nx * batch, nin + nout)
        plan.destroy(); x.free(); out.free()
        if y:
            y.free()


if __name__ == "__main__":
    main()
