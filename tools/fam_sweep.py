"""Time every butterfly-family assignment (single-pass, column pass, row pass) for a few shapes (experiment tool)."""
import itertools, os, subprocess, sys, json
shapes = sys.argv[1:] or ["1048576,128,f64", "65536,1024,f64", "1024,32768,f64", "1048576,512,f32"]
code = r'''
import sys
sys.path.insert(0, "fft-implementation-in-c_amd")
import numpy as np, fftlib
n, batch, dt = int(sys.argv[1]), int(sys.argv[2]), (np.complex128 if sys.argv[3] == "f64" else np.complex64)
fftlib.init()
p = fftlib.Plan(n, batch, -1, dt)
nbytes = n * batch * np.dtype(dt).itemsize
a = fftlib.DeviceBuffer(nbytes); b = fftlib.DeviceBuffer(nbytes)
p.timed(a.ptr, b.ptr, 2)
print(p.timed(a.ptr, b.ptr, 10) / 10)
'''
for sh in shapes:
    n, batch, dt = sh.split(",")
    res = {}
    for fams in itertools.product((0, 1, 2), repeat=3):
        if int(n) <= 4096 and (fams[1], fams[2]) != (0, 0):
            continue
        if int(n) > 4096 and fams[0] != 0:
            continue
        env = dict(os.environ, FFT_HIP_AUTO_FAMS="%d,%d,%d" % fams)
        out = subprocess.run([sys.executable, "-c", code, n, batch, dt], capture_output=True, text=True, env=env)
        try:
            res[fams] = float(out.stdout.strip().splitlines()[-1])
        except Exception:
            res[fams] = None
    best = sorted((v, k) for k, v in res.items() if v)
    print(sh, " ".join("%s:%.3f" % ("".join(map(str, k)), v) for v, k in best))
