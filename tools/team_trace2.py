"""Timeline of team_defer_kernel (fft_team_defer.h) from its in-kernel clock log.
python tools/team_trace2.py [log2n] [batch] [f32|f64]
Events per workgroup: 0 = team formed; transform 0: 4 column tiles, row phases 0, 1, 2 (7 events); every later
transform: 4 column tiles, the deferred phase 3 of the previous transform, row phases 0, 1, 2 (8 events)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fftlib  # noqa: E402
import oracle_lib as O  # noqa: E402

NAMES = ["A tile 0", "A tile 1", "A tile 2", "A tile 3", "B ph3 (prev)", "B ph0", "B ph1", "B ph2"]


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dtype = np.complex128 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.complex64
    n = 1 << log2n
    fftlib.init()
    fftlib.set_policy(team=2)
    x = O.gen_lcg(n, 3, 8).astype(dtype)
    x = np.tile(x, (batch // 8, 1))
    buf = fftlib.DeviceBuffer(x.nbytes)
    out = fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan = fftlib.Plan(n, batch, -1, dtype)
    assert plan.info().team_tiles == 4
    events = 1 + 7 + 8 * 7 + 1
    tr = fftlib.DeviceBuffer(256 * events * 8)
    tr.upload(np.zeros(256 * events, dtype=np.int64))
    plan.timed(buf.ptr, out.ptr, 3)
    ms = plan.timed(buf.ptr, out.ptr, 10) / 10
    n_teams = 8 << (20 - log2n - (1 if dtype == np.complex128 else 0))
    print("n=2^%d batch %d: %.3f ms = %.1f Gpoint/s (%.2f us per transform per team, %d teams)" %
          (log2n, batch, ms, n * batch / ms / 1e6, ms * 1e3 / (batch / n_teams), n_teams))
    plan.lib.fft_gpu_plan_team_trace_hip(plan.handle, tr.ptr, events)
    plan.execute_ptr(buf.ptr, out.ptr)
    print("status", plan.team_status())
    t = tr.download((256, events), np.int64).astype(np.float64) / 100.0  # us
    t0 = t[:, 0].min()
    print("team formation: spread %.2f us" % (t[:, 0].max() - t0))
    for it in (2, 5):
        base = 1 + 7 + 8 * (it - 1)
        prev = t[:, base - 1]
        print("transform #%d of each team: event, mean delta us over 256 workgroups (min..max), cumulative" % it)
        cum = 0.0
        for i, nm in enumerate(NAMES):
            d = t[:, base + i] - prev
            cum += d.mean()
            print("  %-14s %7.2f  (%6.2f .. %6.2f)   %7.2f" % (nm, d.mean(), d.min(), d.max(), cum))
            prev = t[:, base + i]


if __name__ == "__main__":
    main()
