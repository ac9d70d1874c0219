#!/bin/bash
# Round 4: deferred result stores issued by a quarter (QUAD_DEFER_STORES=2) / a half (=3) of the waves at a time, at different points of the chunk
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
SIZES="20:512" REPS=3 bash $R/tools/ab_quad.sh run base defer1 defer2 defer3 || exit 1
SIZES="18:1024 16:4096 19:1024" REPS=2 bash $R/tools/ab_quad.sh run base defer3 defer2 || exit 1
} 2>&1 | tee $R/gpurun_out/r4_defer4.txt
