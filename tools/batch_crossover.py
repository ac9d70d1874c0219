"""Team kernel vs multi-pass schedule over the batch size (the crossover the planner's min_batch encodes).
python tools/batch_crossover.py  ->  one line per (log2n, batch): Gpoint/s of both schedules"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fftlib  # noqa: E402


def main():
    fftlib.init()
    sizes = ((20, np.complex64), (19, np.complex64), (18, np.complex64), (17, np.complex64), (16, np.complex64), (19, np.complex128), (17, np.complex128), (15, np.complex128))
    if len(sys.argv) > 1:  # e.g. "20,18,16": fp32 sizes only (round 3: team_quad_kernel)
        sizes = tuple((int(v), np.complex64) for v in sys.argv[1].split(","))
    for log2n, dtype in sizes:
        n = 1 << log2n
        esz = np.dtype(dtype).itemsize
        for batch in (8, 16, 32, 64, 128, 256, 512):
            b = batch * (1 << (20 - log2n)) if dtype == np.complex64 else batch * (1 << (19 - log2n))
            if b * n * esz > (4 << 30):
                continue
            buf = fftlib.DeviceBuffer(b * n * esz)
            out = fftlib.DeviceBuffer(b * n * esz)
            res = []
            for mode in (0, 2):
                fftlib.set_policy(team=mode)
                p = fftlib.Plan(n, b, -1, dtype)
                p.timed(buf.ptr, out.ptr, 2)
                ms = sorted(p.timed(buf.ptr, out.ptr, 5) / 5 for _ in range(3))[1]
                res.append((n * b / ms / 1e6, p.team_status()))
                p.destroy()
            buf.free()
            out.free()
            print("n=2^%d %s batch %5d (%6.0f MiB): multi-pass %6.1f  team %6.1f Gpoint/s (status %d)  -> %s" %
                  (log2n, np.dtype(dtype).name, b, b * n * esz / 2**20, res[0][0], res[1][0], res[1][1],
                   "team" if res[1][0] > res[0][0] else "multi-pass"), flush=True)


if __name__ == "__main__":
    main()
