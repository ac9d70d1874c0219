"""Default policy around the team kernels' batch crossovers: for every quad size, batches just below / at / above the planner's
min_batch and a ragged multiple, out of place and in place, forward and inverse, every transform checked against numpy.
python tools/edge_batches.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
import fftlib  # noqa: E402


def main():
    fftlib.init()
    rng = np.random.default_rng(5)
    bad = 0
    for dtype, sizes in ((np.complex64, range(15, 21)), (np.complex128, range(14, 17))):
        esz = np.dtype(dtype).itemsize
        for log2n in sizes:
            n = 1 << log2n
            mib = 128 if (dtype == np.complex64 and log2n >= 19) else 256
            mb = max(1, (mib << 20) // (n * esz))
            x8 = (rng.standard_normal((8, n)) + 1j * rng.standard_normal((8, n))).astype(dtype)
            ref = {-1: np.fft.fft(x8.astype(np.complex128), axis=1), 1: np.fft.ifft(x8.astype(np.complex128), axis=1)}
            tol = 2e-6 if dtype == np.complex64 else 1e-12
            for batch in (mb - 1, mb, mb + 1, 2 * mb + 5):
                x = x8[np.arange(batch) % 8]
                buf, out = fftlib.DeviceBuffer(x.nbytes), fftlib.DeviceBuffer(x.nbytes)
                for d in (-1, 1):
                    plan = fftlib.Plan(n, batch, d, dtype)
                    for inplace in (False, True):
                        buf.upload(x)
                        plan.execute_ptr(buf.ptr, buf.ptr if inplace else out.ptr)
                        st = plan.team_status()
                        y = (buf if inplace else out).download(x.shape, dtype)
                        err = max(float(np.linalg.norm(y[i] - ref[d][i % 8]) / np.linalg.norm(ref[d][i % 8])) for i in range(batch))
                        ok = err < tol and st in (0, -1)
                        bad += not ok
                        if not ok or (inplace and d == 1):
                            print("n=2^%d %s batch %d (crossover %d) d=%d inplace=%d: team status %d, max rel err %.2e %s" %
                                  (log2n, np.dtype(dtype).name, batch, mb, d, inplace, st, err, "" if ok else "  <-- BAD"), flush=True)
                    plan.destroy()
                buf.free(); out.free()
    print("bad:", bad)
    sys.exit(1 if bad else 0)


main()
