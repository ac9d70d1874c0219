#!/bin/bash
# Round 4: the schedule variants again, but with the window aliased onto 2 MiB (FFT_HIP_TEAM_ABLATE=32, results invalid): with the full window the
# launch sits on the memory side's pace for its traffic mix (tools/membench8.hip) and no schedule can show; with 2 MiB that pace is 1.77 ms
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export AB_NOCHECK=1
mkdir -p $R/gpurun_out
{
for rep in 1 2; do
  echo "## window aliased onto 2 MiB"
  FFT_HIP_TEAM_ABLATE=32 SIZES="20:512" REPS=1 bash $R/tools/ab_quad.sh run old base defer3 free free3 free0 abl1 abl41 || exit 1
  echo "## ... and no team polls either (40)"
  FFT_HIP_TEAM_ABLATE=40 SIZES="20:512" REPS=1 bash $R/tools/ab_quad.sh run old base free || exit 1
done
} 2>&1 | tee $R/gpurun_out/r4_alias.txt
