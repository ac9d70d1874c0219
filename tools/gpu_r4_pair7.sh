#!/bin/bash
# Round 4: the pair protocol (FFT_HIP_QUAD_SLOTS=3) against each size's shipped protocol, n = 2^17 ... 2^20, every transform checked
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
for rep in 1 2 3; do
  for sz in 20:512 19:1024 18:1024 17:2048; do
    timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "shipped" || exit 1
    FFT_HIP_QUAD_SLOTS=3 timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "pair" || exit 1
  done
done
for sz in 20:8 20:72 19:24 18:40 17:136; do
  FFT_HIP_QUAD_SLOTS=3 timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "pair, small batch" || exit 1
done
} 2>&1 | tee $R/gpurun_out/r4_pair7.txt
