#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
run() { timeout -k 10 120 python3 $R/tools/team_time.py "$@" || exit 1; }
for rep in 1 2 3; do
  for sl in 1 2; do
    FFT_HIP_QUAD_SLOTS=$sl run 18 1024 f32 "2^18 slots=$sl"
    FFT_HIP_QUAD_SLOTS=$sl run 18 2048 f32 "2^18 slots=$sl"
  done
done
