#!/bin/bash
# round 3: team_quad_kernel against team_defer_kernel at n = 2^20, 2^18, 2^16 (BASELINE configs 3, 4 shard, 2), same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
for rep in 1 2; do
for sz in 20:512 18:1024 16:4096 18:2048 16:8192; do
FFT_HIP_TEAM_QUAD=1 timeout -k 10 120 python3 tools/team_time.py ${sz%%:*} ${sz##*:} f32 quad || echo "FAILED quad $sz rc=$?"
FFT_HIP_TEAM_QUAD=0 timeout -k 10 120 python3 tools/team_time.py ${sz%%:*} ${sz##*:} f32 defer || echo "FAILED defer $sz rc=$?"
done
done
