#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
run() { timeout -k 10 120 python3 $R/tools/team_time.py "$@" || exit 1; }
for rep in 1 2; do
  FFT_HIP_TEAM_QUAD=0 run 15 16384 f32 "2^15 multi-pass/defer"
  run 15 16384 f32 "2^15 quad team of one"
  FFT_HIP_TEAM_QUAD=0 run 15 8192 f32 "2^15 multi-pass/defer"
  run 15 8192 f32 "2^15 quad team of one"
done
python3 $R/tools/batch_crossover.py 15
