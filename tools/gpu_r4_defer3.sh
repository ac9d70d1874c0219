#!/bin/bash
# Round 4: deferred result stores, the last part (or half of it) behind the arrival of round 0 (QUAD_LAST_PART_LATE = 2 / 1 / 0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
SIZES="20:512" REPS=3 bash $R/tools/ab_quad.sh run old base late1 late0 || exit 1
SIZES="19:1024 18:1024 17:2048 16:4096 15:8192" REPS=2 bash $R/tools/ab_quad.sh run old base late0 || exit 1
FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so timeout -k 10 120 python3 $R/tools/quad_trace.py 512 || exit 1
} 2>&1 | tee $R/gpurun_out/r4_defer3.txt
