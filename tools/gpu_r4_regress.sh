#!/bin/bash
# Round 4: the product library after the pair-protocol work against the same library with the quad kernels of the commit before it (same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/fft-implementation-in-c_amd
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
for rep in 1 2 3; do
  for lib in $P/libfft_mi355x.so $P/build/variants/libq_oldp.so; do
    for sz in 20:512 18:1024 16:4096; do
      FFT_LIB_PATH=$lib timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "$(basename $lib)" || exit 1
    done
  done
done
} 2>&1 | tee $R/gpurun_out/r4_ab_regress.txt
