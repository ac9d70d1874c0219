#!/bin/bash
# Round 4: what a column step whose waves run free of each other would be worth (timing only: no landing barrier, no stage barrier in the
# column step = QUAD_ABL 66, results invalid) -- with the deferred result stores (free), without (free0), and with no result stores (free1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export AB_NOCHECK=1
mkdir -p $R/gpurun_out
{
SIZES="20:512" REPS=3 bash $R/tools/ab_quad.sh run base free free0 free1 || exit 1
SIZES="18:1024 16:4096" REPS=2 bash $R/tools/ab_quad.sh run base free0 || exit 1
} 2>&1 | tee $R/gpurun_out/r4_free.txt
