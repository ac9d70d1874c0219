#!/bin/bash
# cache-policy bits of team_quad_kernel's streams: FFT_HIP_TEAM_NT bit 0 column DMA nt, bit 1 result stores nt, bit 2 window DMA sc1 nt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
for rep in 1 2; do
for nt in 3 7 1 2 0; do
FFT_HIP_TEAM_NT=$nt timeout -k 10 120 python3 tools/team_time.py 20 512 f32 nt$nt || exit 1
done
done
