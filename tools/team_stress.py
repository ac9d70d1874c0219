"""Stress test of the team kernel's same-XCD hand-over: many back-to-back executes, every output compared BIT FOR BIT
with the first one (a stale or torn read anywhere in the 4 GiB shows up as a mismatch).
python tools/team_stress.py [log2n] [batch] [iterations] [f32|f64]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("FFT_LIB_PATH", os.path.join(ROOT, "fft-implementation-in-c_amd", "libfft_mi355x_exp.so"))  # experiment switches live in the -DFFT_EXPERIMENTS build only
import fftlib  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    dtype = np.complex64 if (len(sys.argv) <= 4 or sys.argv[4] == "f32") else np.complex128
    os.environ.setdefault("FFT_HIP_TEAM", "2")
    n = 1 << log2n
    fftlib.init()
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((batch, n), dtype=np.float32) + 1j * rng.standard_normal((batch, n), dtype=np.float32)).astype(dtype)
    buf = fftlib.DeviceBuffer(x.nbytes)
    out = fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan = fftlib.Plan(n, batch, -1, dtype)
    assert plan.info().team_tiles == 4
    plan.execute_ptr(buf.ptr, out.ptr)
    assert plan.team_status() == 0
    first = out.download(x.shape, dtype)
    ref = O.oracle_fft(x[batch - 1:batch].astype(np.complex128), -1, "dit")
    err = np.linalg.norm(first[batch - 1:batch] - ref) / np.linalg.norm(ref)
    print("n=2^%d batch=%d %s: rel err of the last transform vs oracle %.3e" % (log2n, batch, np.dtype(dtype).name, err), flush=True)
    bad = 0
    for it in range(iters):
        out.upload(np.zeros(4096, dtype=dtype))  # disturb a little
        for _ in range(4):
            plan.execute_ptr(buf.ptr, out.ptr)
        st = plan.team_status()
        y = out.download(x.shape, dtype)
        same = np.array_equal(y.view(np.uint8), first.view(np.uint8))
        if st != 0 or not same:
            bad += 1
            diff = np.argwhere(y != first)
            print("  iteration %d: status %d, %d differing elements, first at %s" % (it, st, len(diff), diff[:1]), flush=True)
    print("%d x 4 executes: %d mismatching / non-zero-status iterations" % (iters, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
