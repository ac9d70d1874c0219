#!/bin/bash
# Round 4: free-running column step (timing only, QUAD_ABL=66) WITH the deferred stores staggered over wave groups (a half / a quarter of the waves at a time)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export AB_NOCHECK=1
mkdir -p $R/gpurun_out
{
SIZES="20:512" REPS=3 bash $R/tools/ab_quad.sh run base free free3 free2 || exit 1
} 2>&1 | tee $R/gpurun_out/r4_free2.txt
