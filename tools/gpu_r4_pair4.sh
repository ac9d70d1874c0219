#!/bin/bash
# Round 4: pair protocol with the round's hand-offs one step later each (QUAD_PAIR_LATE=1; valid: every transform checked), and the same without
# guards (timing only), against the shipped two-slot kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
FFT_HIP_QUAD_SLOTS=3 REPS=1 bash $R/tools/ab_quad.sh run late || exit 1
export AB_NOCHECK=1
for rep in 1 2; do
FFT_HIP_QUAD_SLOTS=2 REPS=1 bash $R/tools/ab_quad.sh run base || exit 1
FFT_HIP_QUAD_SLOTS=3 REPS=1 bash $R/tools/ab_quad.sh run base late lateg || exit 1
done
} 2>&1 | tee $R/gpurun_out/r4_pair4.txt
