"""Timing sweep of the team kernel's variants (GPU): for each size, FFT_HIP_TEAM_DEFER x FFT_HIP_TEAM_NT, one child
process per combination (the knobs are read once per process), against the multi-pass schedule.
python tools/team_sweep.py [f32|f64] [log2n ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%r, "fft-implementation-in-c_amd"))
import fftlib
log2n, dtype = int(sys.argv[1]), (np.complex64 if sys.argv[2] == "f32" else np.complex128)
n = 1 << log2n
batch = int(os.environ.get("SWEEP_BATCH", "0")) or max(1, (4 << 30) // (n * np.dtype(dtype).itemsize))
fftlib.init()
a = fftlib.DeviceBuffer(n * batch * np.dtype(dtype).itemsize)
b = fftlib.DeviceBuffer(n * batch * np.dtype(dtype).itemsize)
a.upload(np.zeros(n * batch, dtype) + 1)
plan = fftlib.Plan(n, batch, -1, dtype)
plan.timed(a.ptr, b.ptr, 3)
best = min(plan.timed(a.ptr, b.ptr, 10) / 10 for _ in range(3))
print("%%.4f %%d %%d" %% (best, plan.info().team_tiles, plan.team_status()))
""" % ROOT


def run(log2n, dtype, env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD, str(log2n), dtype], env=e, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        return None, out.stderr[-300:]
    ms, tiles, st = out.stdout.split()[-3:]
    return float(ms), (int(tiles), int(st))


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    sizes = [int(a) for a in sys.argv[2:]] or ([17, 18, 19, 20] if dtype == "f32" else [15, 16, 17, 18, 19])
    sz = 8 if dtype == "f32" else 16
    for log2n in sizes:
        n = 1 << log2n
        batch = int(os.environ.get("SWEEP_BATCH", "0")) or max(1, (4 << 30) // (n * sz))
        row = []
        ms, info = run(log2n, dtype, {"FFT_HIP_TEAM": "0"})
        row.append("multi-pass %.1f" % (n * batch / ms / 1e6))
        if os.environ.get("SWEEP_QUICK"):  # the shipped defaults only (plus whatever knobs the caller exported), three times
            for _ in range(3):
                ms, info = run(log2n, dtype, {"FFT_HIP_TEAM": "2"})
                row.append("default %s" % ("%.1f" % (n * batch / ms / 1e6) if ms else "fail"))
            print("2^%d %s x %d [Gpoint/s]: %s" % (log2n, dtype, batch, " | ".join(row)), flush=True)
            continue
        for defer in (1, 0):
            for nt in (0, 1, 3, 7):
                ms, info = run(log2n, dtype, {"FFT_HIP_TEAM": "2", "FFT_HIP_TEAM_DEFER": str(defer), "FFT_HIP_TEAM_NT": str(nt)})
                row.append("d%d/nt%d %s" % (defer, nt, "%.1f" % (n * batch / ms / 1e6) if ms else "fail"))
        print("2^%d %s x %d [Gpoint/s]: %s" % (log2n, dtype, batch, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
