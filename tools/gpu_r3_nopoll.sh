#!/bin/bash
# What the team waits of team_quad_kernel cost: FFT_HIP_TEAM_ABLATE=8 skips the polls (results invalid; static split so that no index is needed)
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
export FFT_HIP_TEAM_DYNAMIC=0
for rep in 1 2; do
  for ab in 0 8; do
    for sz in 20:512 19:1024 18:2048 16:8192; do
      AB_NOCHECK=1 FFT_HIP_TEAM_ABLATE=$ab timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "ablate=$ab" || exit 1
    done
  done
done
