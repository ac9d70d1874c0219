#!/bin/bash
# Round 4, first call: what each part of team_quad_kernel's transform costs at n = 2^20 x 512 (timing-only ablation builds, QUAD_ABL bits of
# fft_team_quad.h: 1 no result stores, 2 / 4 no stage barrier in the column / row step, 8 / 32 no column / row arithmetic, 16 no window stores),
# alone and with the window aliased onto 2 MiB (FFT_HIP_TEAM_ABLATE=32), and a fresh timeline.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export AB_NOCHECK=1
mkdir -p $R/gpurun_out
{
for rep in 1 2; do
  echo "## window as shipped"
  REPS=1 bash $R/tools/ab_quad.sh run base abl1 abl6 abl8 abl32 abl40 abl16 abl17 abl41 abl57 || exit 1
  echo "## the same with the window aliased onto 2 MiB (FFT_HIP_TEAM_ABLATE=32)"
  for v in base abl1 abl6 abl40 abl41; do
    FFT_HIP_TEAM_ABLATE=32 REPS=1 bash $R/tools/ab_quad.sh run $v || exit 1
  done
done
FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so timeout -k 10 120 python3 $R/tools/quad_trace.py 512 || exit 1
} 2>&1 | tee $R/gpurun_out/r4_price.txt
