"""Timeline of the team kernel (fft_team.h) from its in-kernel clock log: where a transform's ~N us go.
python tools/team_trace.py [batch] [ablate]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fftlib  # noqa: E402
import oracle_lib as O  # noqa: E402

NT = 4
A_EV = ["A landed", "A done"]
B_EV = ["B st issued", "B st in L2", "B barrier", "B landed", "B rows done", "B out issued"]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n = 1 << 20
    fftlib.init()
    x = O.gen_lcg(n, 3, 8).astype(np.complex64)
    x = np.tile(x, (batch // 8, 1))
    buf = fftlib.DeviceBuffer(x.nbytes)
    out = fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    per = NT * len(A_EV) + NT * len(B_EV)
    events = 1 + per * 6
    tr = fftlib.DeviceBuffer(256 * events * 8)
    tr.upload(np.zeros(256 * events, dtype=np.int64))
    plan.timed(buf.ptr, out.ptr, 3)
    ms = plan.timed(buf.ptr, out.ptr, 10) / 10
    print("batch %d: %.3f ms = %.1f Gpoint/s (%.1f us per transform per team)" % (batch, ms, n * batch / ms / 1e6, ms * 1e3 / (batch / 8)))
    plan.lib.fft_gpu_plan_team_trace_hip(plan.handle, tr.ptr, events)
    plan.execute_ptr(buf.ptr, out.ptr)
    print("status", plan.team_status())
    t = tr.download((256, events), np.int64).astype(np.float64) / 100.0  # us
    t0 = t[:, 0].min()
    print("team formation: first %.2f us, last %.2f us after the earliest workgroup" % (t[:, 0].min() - t0, t[:, 0].max() - t0))
    names = []
    for k in range(NT):
        names += ["%s %d" % (a, k) for a in A_EV]
    for k in range(NT):
        names += ["%s %d" % (b, k) for b in B_EV]
    for tr_i in (1, 3):  # second and fourth transform of each team
        base = 1 + per * tr_i
        prev = t[:, base - 1]
        print("transform #%d of each team: event, mean delta us over 256 workgroups (min..max), cumulative" % tr_i)
        cum = 0.0
        for i, nm in enumerate(names):
            d = t[:, base + i] - prev
            cum += d.mean()
            print("  %-16s %7.2f  (%6.2f .. %6.2f)   %7.2f" % (nm, d.mean(), d.min(), d.max(), cum))
            prev = t[:, base + i]


if __name__ == "__main__":
    main()
