#!/bin/bash
# Round 4: the next round's image requested at the START of a round (QUAD_EARLY_REQ=1) instead of behind stage 1; every transform checked
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
SIZES="20:512 19:1024" REPS=3 bash $R/tools/ab_quad.sh run base ereq || exit 1
} 2>&1 | tee $R/gpurun_out/r4_ereq.txt
