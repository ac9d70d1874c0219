#!/bin/bash
# Round 4: the deferred result stores cut into slices inside the chunk's arithmetic (QUAD_STORE_SPLIT), with / without QUAD_EARLY_DMA
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
SIZES="20:512" REPS=2 bash $R/tools/ab_quad.sh run old base s4ne s2 s2ne s1ne || exit 1
SIZES="18:1024 16:4096" REPS=2 bash $R/tools/ab_quad.sh run old base s4ne || exit 1
FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so timeout -k 10 120 python3 $R/tools/quad_trace.py 512 || exit 1
} 2>&1 | tee $R/gpurun_out/r4_defer2.txt
