"""A/B of the fp64 plans with 128 KiB tiles (16 values per thread, csrc/fft_engine.h build()) against the 64 KiB ones: run once per
value of FFT_HIP_E16 (0 / 1) with the experiments library.  python tools/ab_e16.py [tag] [NxB,...]  (the kernels of this A/B are not in the tree any more: profiles/r3_ab_e16_fp64.txt)
Every line checks the result against numpy (first and last transform)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
import fftlib  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("FFT_HIP_E16", "default")
    fftlib.init()
    cases = [(1 << 20, 128), (1 << 21, 64), (1 << 22, 32), (1000003, 64), (1500007, 32), (600011, 64)]
    if len(sys.argv) > 2:
        cases = [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[2].split(",")]
    for n, batch in cases:
        rng = np.random.default_rng(7)
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex128)
        buf = fftlib.DeviceBuffer(x.nbytes); buf.upload(x)
        out = fftlib.DeviceBuffer(x.nbytes)
        plan = fftlib.Plan(n, batch, -1, np.complex128)
        plan.execute_ptr(buf.ptr, out.ptr); plan.sync()
        y = out.download(x.shape, np.complex128)
        err = 0.0
        for b in (0, batch - 1):
            ref = np.fft.fft(x[b])
            err = max(err, float(np.linalg.norm(y[b] - ref) / np.linalg.norm(ref)))
        assert err < 1e-12, (n, err)
        plan.timed(buf.ptr, out.ptr, 2)
        ms = sorted(plan.timed(buf.ptr, out.ptr, 5) / 5 for _ in range(5))
        info = plan.info()
        print("E16=%-8s n=%8d x %4d fp64: median %.3f ms (min %.3f) = %6.1f Gpoint/s  passes %d factors %s m=%d fused %d  rel err %.1e" %
              (tag, n, batch, ms[2], ms[0], n * batch / ms[2] / 1e6, info.n_passes, list(info.factors), info.bluestein_m, info.fused, err), flush=True)
        plan.destroy(); buf.free(); out.free()


main()
