#!/bin/bash
# Round 4: every stream-yardstick shape's rate (the experiments build prints them), 1 GiB per direction
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so FFT_HIP_STREAM_VERBOSE=1 timeout -k 10 300 python3 tools/stream_bench.py 1024 > gpurun_out/r4_stream2.txt 2>&1 || { tail -5 gpurun_out/r4_stream2.txt; exit 1; }
sort -k8 -n -r gpurun_out/r4_stream2.txt | grep "mode 0" | head -8
grep MiB gpurun_out/r4_stream2.txt
