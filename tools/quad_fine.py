"""Inside the column chunks of team_quad_kernel: python tools/quad_fine.py   (library built with -DQUAD_FINE_TRACE=1)
Stamps of every wave's first lane in transform 3: per chunk 9 stamps -- B1 passed, stage-1 reads landed, dft, twiddle, writes
done, B2 passed, stage-2 reads landed, dft, chunk twiddle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fftlib  # noqa: E402
import oracle_lib as O  # noqa: E402

NAMES = ["B1 passed", "s1 reads landed", "s1 dft", "s1 twiddle", "s1 writes done", "B2 passed", "s2 reads landed", "s2 dft", "chunk twiddle"]


def main():
    n, batch = 1 << 20, 512
    fftlib.init()
    x = O.gen_lcg(n, 3, 8).astype(np.complex64)
    x = np.tile(x, (batch // 8, 1))
    buf, out = fftlib.DeviceBuffer(x.nbytes), fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    tr = fftlib.DeviceBuffer(256 * 8 * 64 * 8)
    tr.upload(np.zeros(256 * 8 * 64, dtype=np.int64))
    plan.timed(buf.ptr, out.ptr, 3)
    plan.lib.fft_gpu_plan_team_trace_hip(plan.handle, tr.ptr, 2)
    plan.execute_ptr(buf.ptr, out.ptr)
    plan.team_status()
    t = tr.download((256 * 8, 64), np.int64).astype(np.float64) / 100.0  # us, per wave
    for a in range(4):
        print("chunk %d: stamp, mean delta us over 2048 waves (min .. max)" % a)
        prev = t[:, a * 12]
        for i in range(1, 9):
            d = t[:, a * 12 + i] - prev
            print("  %-16s %6.2f  (%5.2f .. %5.2f)" % (NAMES[i], d.mean(), d.min(), d.max()))
            prev = t[:, a * 12 + i]
        if a < 3:
            d = t[:, (a + 1) * 12] - prev
            print("  %-16s %6.2f  (%5.2f .. %5.2f)" % ("-> next B1", d.mean(), d.min(), d.max()))


if __name__ == "__main__":
    main()
