"""GPU check of the team kernel (fft_team.h): parity against the two-pass schedule and the oracle, status word,
timing.  python tools/team_check.py [log2n] [batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("FFT_LIB_PATH", os.path.join(ROOT, "fft-implementation-in-c_amd", "libfft_mi355x_exp.so"))  # experiment switches live in the -DFFT_EXPERIMENTS build only
import fftlib  # noqa: E402
import oracle_lib as O  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.complex128) - b) / np.linalg.norm(b))


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dtype = np.complex64 if (len(sys.argv) <= 3 or sys.argv[3] == "f32") else np.complex128
    n = 1 << log2n
    os.environ.setdefault("FFT_HIP_TEAM", "2")  # every geometry the team kernel is built for, any batch
    fftlib.init()
    x = O.gen_lcg(n, 3, batch).astype(dtype)
    buf_in = fftlib.DeviceBuffer(x.nbytes)
    buf_out = fftlib.DeviceBuffer(x.nbytes)
    buf_in.upload(x)
    for direction in (-1, 1):
        plan = fftlib.Plan(n, batch, direction, dtype)
        pi = plan.info()
        print("plan: n=2^%d batch=%d dir=%+d passes=%d factors=%s team_tiles=%d" %
              (log2n, batch, direction, pi.n_passes, list(pi.factors)[:pi.n_passes], pi.team_tiles), flush=True)
        poison = np.full_like(x, np.nan)
        buf_out.upload(poison)
        t0 = time.time()
        plan.execute_ptr(buf_in.ptr, buf_out.ptr)
        st = plan.team_status()
        print("  first execute: %.1f ms wall, team status = %d" % ((time.time() - t0) * 1e3, st), flush=True)
        y = buf_out.download(x.shape, dtype)
        os.environ["FFT_HIP_TEAM"] = "0"
        plan2 = fftlib.Plan(n, batch, direction, dtype)
        os.environ["FFT_HIP_TEAM"] = "2"
        assert plan2.info().team_tiles == 0
        buf_out.upload(poison)
        plan2.execute_ptr(buf_in.ptr, buf_out.ptr)
        plan2.sync()
        y2 = buf_out.download(x.shape, dtype)
        print("  team vs two-pass: rel = %.3e, nan = %d" % (rel(y, y2.astype(np.complex128)), int(np.isnan(y).sum())), flush=True)
        for b in sorted(set([0, 1, 7, 8, batch // 2, batch - 1])):
            ref = O.oracle_fft(x[b:b + 1].astype(np.complex128), direction, "exact")
            print("    transform %d vs oracle: %.3e" % (b, rel(y[b:b + 1], ref)), flush=True)
        # in place
        buf_out.upload(x)
        plan.execute_ptr(buf_out.ptr, buf_out.ptr)
        st = plan.team_status()
        yi = buf_out.download(x.shape, dtype)
        print("  in place: status %d, rel vs out-of-place = %.3e" % (st, rel(yi, y.astype(np.complex128))), flush=True)
        for p, name in ((plan, "team"), (plan2, "two-pass")):
            p.timed(buf_in.ptr, buf_out.ptr, 2)
            iters = 10
            ms = p.timed(buf_in.ptr, buf_out.ptr, iters) / iters
            print("  %-8s %.3f ms / execute = %.1f Gpoint/s, %.2f TB/s algorithmic; status %d" %
                  (name, ms, n * batch / ms / 1e6, 2 * x.nbytes / ms / 1e9, p.team_status()), flush=True)
        plan.destroy()
        plan2.destroy()


if __name__ == "__main__":
    main()
