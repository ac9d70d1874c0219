#!/bin/bash
# Round 4: pair protocol with one copy of the `landed` line per reading seat (every transform checked), early and late hand-offs, against the
# shipped two-slot kernel; and without guards (timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
for rep in 1 2; do
FFT_HIP_QUAD_SLOTS=2 REPS=1 bash $R/tools/ab_quad.sh run base || exit 1
FFT_HIP_QUAD_SLOTS=3 REPS=1 bash $R/tools/ab_quad.sh run base late || exit 1
AB_NOCHECK=1 FFT_HIP_QUAD_SLOTS=3 REPS=1 bash $R/tools/ab_quad.sh run pg lateg || exit 1
done
} 2>&1 | tee $R/gpurun_out/r4_pair6.txt
