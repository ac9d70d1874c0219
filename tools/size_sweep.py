import sys, time
sys.path.insert(0, "fft-implementation-in-c_amd"); sys.path.insert(0, "tests")
import numpy as np, fftlib
lib = fftlib.init()
for n, batch, dt in ((1<<20, 128, np.complex128), (1<<16, 1024, np.complex128), (1024, 32768, np.complex128), (1<<22, 64, np.complex64), (1<<24, 16, np.complex64), (4096, 16384, np.complex64), (256, 1<<18, np.complex64), (64, 1<<20, np.complex64)):
    p = fftlib.Plan(n, batch, -1, dt)
    nbytes = n * batch * np.dtype(dt).itemsize
    a = fftlib.DeviceBuffer(nbytes); b = fftlib.DeviceBuffer(nbytes)
    x = (np.random.default_rng(0).standard_normal((min(batch,4), n)) + 0j).astype(dt)
    p.timed(a.ptr, b.ptr, 2)
    ms = p.timed(a.ptr, b.ptr, 10) / 10
    info = p.info()
    print("n=%d batch=%d %s: %.3f ms  %.1f Gpt/s  alg %.2f TB/s  passes %s chunk %d" % (n, batch, np.dtype(dt).name, ms, n*batch/ms/1e6, 2*nbytes/ms/1e9, [v for v in info.factors if v], info.chunk_batch))
    p.destroy(); a.free(); b.free()
