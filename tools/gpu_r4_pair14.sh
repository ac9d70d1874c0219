#!/bin/bash
# Round 4: n = 2^18 and 2^17 once more on the pair protocol with its final schedule (signal from behind stage 1's butterflies), against their shipped
# one-slot team protocol; every transform checked
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
for rep in 1 2 3 4; do
  for sz in 18:1024 17:2048; do
    timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "shipped" || exit 1
    FFT_HIP_QUAD_SLOTS=3 timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "pair" || exit 1
  done
done
} 2>&1 | tee $R/gpurun_out/r4_pair14.txt
