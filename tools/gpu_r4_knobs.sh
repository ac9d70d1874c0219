#!/bin/bash
# Round 4: the small knobs of team_quad_kernel re-measured with the deferred result stores in place (n = 2^20 x 512)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
SIZES="20:512" REPS=2 bash $R/tools/ab_quad.sh run base ec1off ec0off sl0 sl1 sl4 || exit 1
for rot in 1 2 3; do
  echo "## FFT_HIP_TEAM_SEAT_ROT=$rot"
  FFT_HIP_TEAM_SEAT_ROT=$rot SIZES="20:512" REPS=1 bash $R/tools/ab_quad.sh run base || exit 1
done
echo "## FFT_HIP_TEAM_DYNAMIC=0"
FFT_HIP_TEAM_DYNAMIC=0 SIZES="20:512" REPS=1 bash $R/tools/ab_quad.sh run base || exit 1
} 2>&1 | tee $R/gpurun_out/r4_knobs.txt
