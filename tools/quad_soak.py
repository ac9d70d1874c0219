"""Soak of the team kernels' protocol: many executes per size, every transform of every execute compared bit for bit with the first
(verified) result.  python tools/quad_soak.py [iterations]   (default policy: dynamic claims, every exchange protocol, teams of 1 ... 32;
SOAK_FULL=1: the full batches of n = 2^20, 2^18, 2^19)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fftlib  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    fftlib.init()
    cases = [(20, 64, np.complex64), (19, 128, np.complex64), (18, 256, np.complex64), (17, 512, np.complex64), (16, 1024, np.complex64),
             (15, 2048, np.complex64), (14, 2048, np.complex128), (15, 1024, np.complex128), (16, 512, np.complex128), (19, 64, np.complex128)]
    if os.environ.get("SOAK_FULL"):  # the BASELINE batches of the three sizes on the pair protocol (4 / 2 / 4 GiB per execute)
        cases = [(20, 512, np.complex64), (18, 1024, np.complex64), (19, 1024, np.complex64)]
    for log2n, batch, dtype in cases:
        n = 1 << log2n
        x8 = O.gen_lcg(n, 11 + log2n, 8).astype(dtype)
        x = np.tile(x8, (batch // 8, 1))
        a, b = fftlib.DeviceBuffer(x.nbytes), fftlib.DeviceBuffer(x.nbytes)
        a.upload(x)
        plan = fftlib.Plan(n, batch, -1, dtype)
        plan.execute_ptr(a.ptr, b.ptr)
        assert plan.team_status() == 0, (log2n, plan.team_status())
        y0 = b.download(x.shape, dtype)
        ref = np.fft.fft(x8.astype(np.complex128), axis=1)
        tol = 1e-5 if dtype == np.complex64 else 1e-12
        for i in range(batch):
            assert np.linalg.norm(y0[i] - ref[i % 8]) / np.linalg.norm(ref[i % 8]) < tol, (log2n, i)
        t = time.time()
        bad = 0
        for it in range(iters):
            if it % 3 == 2:  # in place every third time
                a2 = fftlib.DeviceBuffer(x.nbytes); a2.upload(x)
                plan.execute_ptr(a2.ptr, a2.ptr)
                st = plan.team_status()
                y = a2.download(x.shape, dtype); a2.free()
            else:
                plan.execute_ptr(a.ptr, b.ptr)
                st = plan.team_status()
                y = b.download(x.shape, dtype)
            if st != 0 or not np.array_equal(y, y0):
                bad += 1
                print("MISMATCH n=2^%d %s iteration %d status %d" % (log2n, np.dtype(dtype).name, it, st), flush=True)
        print("n=2^%d x %d %s: %d executes, %d bad (%.1f s), team kernel %d" % (log2n, batch, np.dtype(dtype).name, iters, bad, time.time() - t, plan.info().team_kernel), flush=True)
        plan.destroy(); a.free(); b.free()
        if bad:
            sys.exit(1)


main()
