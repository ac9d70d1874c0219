"""Timeline of team_quad_kernel (fft_team_quad.h) from its in-kernel clock log.  python tools/quad_trace.py [batch]
Events per workgroup: 0 = team formed; per transform 16: the four column chunks landed, combine done, round-0 values in L2
(arrival), then per round: team wait over, image landed; final radix-4 done, result stores issued.  Round 4: on teams of 32 the result
stores of a transform are issued in front of the NEXT transform's column chunks (QUAD_DEFER_STORES): the last two events then coincide
and the store time shows up in the chunk events of the following transform."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fftlib  # noqa: E402
import oracle_lib as O  # noqa: E402

NAMES = ["chunk0 landed", "chunk1 landed", "chunk2 landed", "chunk3 landed", "send0 in L2 (arrive)", "combine done",
         "wait0 over", "img0 landed", "wait1 over", "img1 landed", "wait2 over", "img2 landed", "wait3 over", "img3 landed",
         "radix-4 done", "stores issued"]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n = 1 << 20
    fftlib.init()
    x = O.gen_lcg(n, 3, 8).astype(np.complex64)
    x = np.tile(x, (batch // 8, 1))
    buf, out = fftlib.DeviceBuffer(x.nbytes), fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    NTR = 8
    events = 1 + 16 * NTR + 1
    tr = fftlib.DeviceBuffer(256 * events * 8)
    tr.upload(np.zeros(256 * events, dtype=np.int64))
    plan.timed(buf.ptr, out.ptr, 3)
    ms = plan.timed(buf.ptr, out.ptr, 10) / 10
    print("n=2^20 batch %d: %.3f ms = %.1f Gpoint/s (%.2f us per transform per team)" % (batch, ms, n * batch / ms / 1e6, ms * 1e3 / (batch / 8)))
    plan.lib.fft_gpu_plan_team_trace_hip(plan.handle, tr.ptr, events)
    plan.execute_ptr(buf.ptr, out.ptr)
    print("status", plan.team_status())
    t = tr.download((256, events), np.int64).astype(np.float64) / 100.0  # us
    print("team formation: spread %.2f us" % (t[:, 0].max() - t[:, 0].min()))
    for it in (2, 5):
        base = 1 + 16 * it
        prev = t[:, base - 1]
        print("transform #%d of each team: event, mean delta us over 256 workgroups (min..max), cumulative" % it)
        cum = 0.0
        for i, nm in enumerate(NAMES):
            d = t[:, base + i] - prev
            cum += d.mean()
            print("  %-14s %7.2f  (%6.2f .. %6.2f)   %7.2f" % (nm, d.mean(), d.min(), d.max(), cum))
            prev = t[:, base + i]


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("seats", "teams")):
    main()


def per_seat():
    """python tools/quad_trace.py seats: column-step duration by seat (mean over teams and transforms 2..6)"""
    n = 1 << 20
    batch = 512
    fftlib.init()
    x = O.gen_lcg(n, 3, 8).astype(np.complex64)
    x = np.tile(x, (batch // 8, 1))
    buf, out = fftlib.DeviceBuffer(x.nbytes), fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    NTR = 8
    events = 1 + 16 * NTR + 1
    tr = fftlib.DeviceBuffer(256 * events * 8)
    tr.upload(np.zeros(256 * events, dtype=np.int64))
    plan.timed(buf.ptr, out.ptr, 3)
    plan.lib.fft_gpu_plan_team_trace_hip(plan.handle, tr.ptr, events)
    plan.execute_ptr(buf.ptr, out.ptr)
    plan.team_status()  # synchronizes the plan's stream
    raw = tr.download((256, events), np.int64)
    t = raw.astype(np.float64) / 100.0
    ident = raw[:, events - 1]
    team, seat = ident >> 8, ident & 255
    col = np.zeros(32)
    chunks = np.zeros((32, 4))
    lag = np.zeros(32)
    for it in range(2, 7):
        base = 1 + 16 * it
        dur = t[:, base + 4] - t[:, base - 1]  # end of previous transform -> my round-0 values in L2
        for a in range(4):
            d = t[:, base + a] - t[:, base + a - 1]
            for s in range(32):
                chunks[s, a] += d[seat == s].mean() / 5
        arr = t[:, base + 4]
        for s in range(32):
            col[s] += dur[seat == s].mean() / 5
            lag[s] += np.mean([arr[(seat == s) & (team == k)].mean() - arr[team == k].mean() for k in sorted(set(team.tolist()))]) / 5
    print("seat: column step us | chunk0 chunk1 chunk2 chunk3 landed | arrival vs team mean")
    for s in range(32):
        print("%2d: %6.2f | %5.2f %5.2f %5.2f %5.2f | %+5.2f" % (s, col[s], chunks[s, 0], chunks[s, 1], chunks[s, 2], chunks[s, 3], lag[s]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "seats":
    per_seat()


def per_team():
    """python tools/quad_trace.py teams [log2n batch]: kernel entry, team formation, first transform and the end of every team's share
    of the batch (ev() slots per transform: 16; the slot before the identity word holds the workgroup's entry time)"""
    log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    n = 1 << log2n
    fftlib.init()
    x = O.gen_lcg(n, 3, 8).astype(np.complex64)
    x = np.tile(x, (batch // 8, 1))
    buf, out = fftlib.DeviceBuffer(x.nbytes), fftlib.DeviceBuffer(x.nbytes)
    buf.upload(x)
    plan = fftlib.Plan(n, batch, -1, np.complex64)
    n_teams = 8 << (20 - log2n)
    NTR = 2 * batch // n_teams + 2  # (claimed transforms: a team may take more than its even share)
    events = 1 + 16 * NTR + 2
    tr = fftlib.DeviceBuffer(256 * events * 8)
    plan.timed(buf.ptr, out.ptr, 3)
    ms = plan.timed(buf.ptr, out.ptr, 10) / 10
    print("n=2^%d batch %d: %.3f ms per execute = %.1f Gpoint/s" % (log2n, batch, ms, n * batch / ms / 1e6))
    plan.lib.fft_gpu_plan_team_trace_hip(plan.handle, tr.ptr, events)
    for rep in range(3):
        tr.upload(np.zeros(256 * events, dtype=np.int64))
        plan.execute_ptr(buf.ptr, out.ptr)
        plan.team_status()
        raw = tr.download((256, events), np.int64)
        t = raw.astype(np.float64) / 100.0
        entry = t[:, events - 2]
        t0 = entry.min()
        ev = t[:, :events - 2]
        last = np.array([row[row > 0].max() if (row > 0).any() else 0.0 for row in ev])  # the workgroup's last event
        first_done = ev[:, 16]                                    # stores of its first transform issued
        cnt = np.array([(row[1:] > 0).sum() // 16 for row in ev])  # transforms it took part in
        print("run %d: workgroups enter %.1f .. %.1f us, teams formed %.1f .. %.1f, first transform done %.1f .. %.1f, last event %.1f .. %.1f; transforms per workgroup %d .. %d" %
              (rep, 0.0, entry.max() - t0, ev[:, 0].min() - t0, ev[:, 0].max() - t0, first_done.min() - t0, first_done.max() - t0, last.min() - t0, last.max() - t0, cnt.min(), cnt.max()))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "teams":
    per_team()
