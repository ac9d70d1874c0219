#!/bin/bash
# round 3: team_quad_kernel one workgroup per CU (QUAD=1) against two per CU (QUAD=2), same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
for rep in 1 2; do
FFT_HIP_TEAM_QUAD=1 timeout -k 10 120 python3 tools/team_time.py 20 512 f32 quad1 || echo "FAILED rc=$?"
FFT_HIP_TEAM_QUAD=2 timeout -k 10 120 python3 tools/team_time.py 20 512 f32 quad2 || echo "FAILED rc=$?"
done
FFT_HIP_TEAM_QUAD=2 timeout -k 10 120 python3 tools/team_time.py 20 128 f32 quad2-b128 || echo "FAILED rc=$?"
FFT_HIP_TEAM_QUAD=2 timeout -k 10 120 python3 tools/team_time.py 20 2048 f32 quad2-b2048 || echo "FAILED rc=$?"
