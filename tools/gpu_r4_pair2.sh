#!/bin/bash
# Round 4: where the pair protocol's time goes (timing only, results invalid): its guards off (QUAD_ABL=128), its waits for the senders off (256),
# both (384); static split of the batch (the dynamic claim rides on a wait).  Against the shipped two-slot kernel and the one-slot team protocol.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export AB_NOCHECK=1 FFT_HIP_TEAM_DYNAMIC=0
mkdir -p $R/gpurun_out
{
FFT_HIP_QUAD_SLOTS=2 REPS=2 bash $R/tools/ab_quad.sh run base || exit 1
FFT_HIP_QUAD_SLOTS=1 REPS=2 bash $R/tools/ab_quad.sh run base || exit 1
FFT_HIP_QUAD_SLOTS=3 REPS=2 bash $R/tools/ab_quad.sh run base pg pw pgw || exit 1
} 2>&1 | tee $R/gpurun_out/r4_pair2.txt
