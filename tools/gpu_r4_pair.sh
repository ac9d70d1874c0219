#!/bin/bash
# Round 4: team_quad_kernel at n = 2^20 with the pair protocol (SLOTS = 3: one image per seat, per-seat counters; FFT_HIP_QUAD_SLOTS=3 in the
# experiments build) against the shipped two-slot team protocol: every transform checked, then timed.
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
for rep in 1 2 3; do
  for slots in 2 3 1; do
    FFT_HIP_QUAD_SLOTS=$slots timeout -k 10 120 python3 $R/tools/team_time.py 20 512 f32 "slots=$slots" || exit 1
  done
done
FFT_HIP_QUAD_SLOTS=3 timeout -k 10 120 python3 $R/tools/team_time.py 20 8 f32 "slots=3 one per team" || exit 1
FFT_HIP_QUAD_SLOTS=3 timeout -k 10 120 python3 $R/tools/team_time.py 20 72 f32 "slots=3 ragged" || exit 1
} 2>&1 | tee $R/gpurun_out/r4_pair.txt
