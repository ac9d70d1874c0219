#!/bin/bash
# Round 4: the price list of tools/gpu_r4_price.sh at BASELINE configs 2 and 4's sizes (n = 2^16 x 4096 on teams of 2, 2^18 x 1024 on teams of 8, one window slot)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export AB_NOCHECK=1
mkdir -p $R/gpurun_out
{
for rep in 1 2; do
  SIZES="16:4096 18:1024" REPS=1 bash $R/tools/ab_quad.sh run old abl1 abl6 abl8 abl32 abl40 abl16 abl17 abl41 abl57 || exit 1
  echo "## no team polls (FFT_HIP_TEAM_ABLATE=8)"
  for v in old abl1 abl41; do
    SIZES="16:4096 18:1024" FFT_HIP_TEAM_ABLATE=8 REPS=1 bash $R/tools/ab_quad.sh run $v || exit 1
  done
done
} 2>&1 | tee $R/gpurun_out/r4_price2.txt
