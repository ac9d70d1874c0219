"""fft_gpu_dft_1d_batch with pageable against page-locked host arrays (the pipelined path): seconds per call, GB/s each way.
python tools/host_batch_time.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
import fftlib  # noqa: E402


def main():
    lib = fftlib.init()
    for n, batch in ((4096, 6221), (4096, 40000), (65536, 2048), (1 << 20, 128)):
        x = (np.random.default_rng(1).standard_normal((batch, n)) + 0j).astype(np.complex128)
        out = np.zeros_like(x)
        gb = x.nbytes / 1e9
        ts = []
        for _ in range(3):
            t = time.perf_counter(); assert lib.fft_gpu_dft_1d_batch(x.ctypes.data, out.ctypes.data, n, batch, -1) == 0; ts.append(time.perf_counter() - t)
        t_page = min(ts)
        keep = out.copy()
        assert lib.fft_gpu_host_register_hip(x.ctypes.data, x.nbytes) == 0 and lib.fft_gpu_host_register_hip(out.ctypes.data, out.nbytes) == 0
        ts = []
        for _ in range(3):
            t = time.perf_counter(); assert lib.fft_gpu_dft_1d_batch(x.ctypes.data, out.ctypes.data, n, batch, -1) == 0; ts.append(time.perf_counter() - t)
        t_pin = min(ts)
        same = np.array_equal(out, keep)
        lib.fft_gpu_host_unregister_hip(x.ctypes.data); lib.fft_gpu_host_unregister_hip(out.ctypes.data)
        print("n=%d x %d (%.2f GB each way): pageable %.3f s (%.1f GB/s each way)  page-locked %.3f s (%.1f GB/s)  identical %s" %
              (n, batch, gb, t_page, gb / t_page, t_pin, gb / t_pin, same), flush=True)


if __name__ == "__main__":
    main()
