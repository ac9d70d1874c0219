#!/bin/bash
# Round 4: cache-policy bits of the team kernel's streams with the deferred result stores (FFT_HIP_TEAM_NT: 1 chunk DMA nt, 2 result stores nt, 4 window loads sc1 nt; default 7)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
{
for rep in 1 2; do
  for nt in 7 5 3 6 1 0; do
    FFT_HIP_TEAM_NT=$nt timeout -k 10 120 python3 $R/tools/team_time.py 20 512 f32 "nt=$nt" || exit 1
  done
done
} 2>&1 | tee $R/gpurun_out/r4_nt.txt
