"""The box's stream yardsticks: python tools/stream_bench.py [MiB ...]  (copy / read-only / write-only, GB/s; fft_gpu_stream_bench_hip)
Small buffers (<= 64 MiB per direction) stay in the 256 MiB Infinity Cache: what a stream costs that never reaches HBM."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))
import fftlib  # noqa: E402

lib = fftlib.init()
for mib in ([int(v) for v in sys.argv[1:]] or [1024]):
    for name, mode in (("copy (read + written)", 0), ("read only", 1), ("write only", 2)):
        print("%5d MiB  %-24s %8.1f GB/s" % (mib, name, lib.fft_gpu_stream_bench_hip(mib << 20, 20 if mib <= 128 else 5, mode)), flush=True)
