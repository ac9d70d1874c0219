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


def event_names():
    names = []
    for k in range(NT):
        names += ["A landed %d" % k, "A done+handover %d" % k]
    names += ["A handover in L2 (a0)", "B all a0", "B tile 0 landed (a1)"]
    for k in range(NT):
        names += ["B rows done %d" % k, "B out issued %d" % k]
        if k + 1 < NT:
            names += ["B phase closed %d" % k]
    return names


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n = 1 << 20
    fftlib.init()
    x = O.gen_lcg(n, 3, 8).astype(np.complex64)
    x = np.tile(x, (batch // 8, 1))
    in_off = int(os.environ.get("TEAM_IN_OFF", "0"))  # experiment: shift the input by this many bytes
    buf = fftlib.DeviceBuffer(x.nbytes + 8192)
    out = fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    names = event_names()
    per = len(names)
    events = 2 + per * 6
    tr = fftlib.DeviceBuffer(256 * events * 8)
    tr.upload(np.zeros(256 * events, dtype=np.int64))
    plan.timed((buf.ptr + in_off), out.ptr, 3)
    ms = plan.timed((buf.ptr + in_off), out.ptr, 10) / 10
    print("batch %d: %.3f ms = %.1f Gpoint/s (%.1f us per transform per team)" % (batch, ms, n * batch / ms / 1e6, ms * 1e3 / (batch / 8)))
    plan.lib.fft_gpu_plan_team_trace_hip(plan.handle, tr.ptr, events)
    plan.execute_ptr((buf.ptr + in_off), out.ptr)
    print("status", plan.team_status())
    t = tr.download((256, events), np.int64).astype(np.float64) / 100.0  # us
    t0 = t[:, 0].min()
    print("team formation: first %.2f us, last %.2f us after the earliest workgroup" % (t[:, 0].min() - t0, t[:, 0].max() - t0))
    seat = tr.download((256, events), np.int64)[:, events - 1]
    a_dur = np.zeros(256)
    for tr_i in range(1, 5):
        base = 1 + per * tr_i
        a_dur += (t[:, base + 2 * NT - 1] - t[:, base - 1]) / 4
    by_seat = np.zeros(32)
    for w in range(256):
        by_seat[int(seat[w]) & 255] += a_dur[w] / 8
    print("A-step duration by seat (mean over teams and transforms 1-4), us:")
    print("  " + " ".join("%.1f" % v for v in by_seat))
    by_team = np.zeros(8)
    for w in range(256):
        by_team[int(seat[w]) >> 8] += a_dur[w] / 32
    print("A-step duration by team: " + " ".join("%.1f" % v for v in by_team))
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
