#!/bin/bash
# Round 4: the small knobs re-measured on the pair protocol (n = 2^20 x 512 and 2^19 x 1024; every transform checked): result stores not deferred /
# deferred by halves of the waves, no early chunk requests, poll sleep, cache-policy bits of the streams
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
SIZES="20:512 19:1024" REPS=2 bash $R/tools/ab_quad.sh run base d0 d3 ec00 sl1 sl4 || exit 1
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
for nt in 7 5 3 6 1 0; do
  FFT_HIP_TEAM_NT=$nt timeout -k 10 120 python3 $R/tools/team_time.py 20 512 f32 "nt=$nt" || exit 1
done
} 2>&1 | tee $R/gpurun_out/r4_pair8.txt
