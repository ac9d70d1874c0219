#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
run() { timeout -k 10 120 python3 $R/tools/team_time.py "$@" || exit 1; }
for rep in 1 2; do
  for sz in 14:8192 15:4096 16:2048 14:16384 16:4096; do
    FFT_HIP_TEAM_QUAD=0 run ${sz%%:*} ${sz##*:} f64 "fp64 before"
    run ${sz%%:*} ${sz##*:} f64 "fp64 quad"
  done
done
