#!/bin/bash
# Round 4: result stores deferred into the next transform's column step (QUAD_DEFER_STORES) and column chunks requested as soon as their
# image is free (QUAD_EARLY_DMA): same-box A/B against the round-3 schedule ("old"), every transform of every execute checked.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
SIZES="20:512" REPS=2 bash $R/tools/ab_quad.sh run old base defer early || exit 1
SIZES="19:1024 18:1024 17:2048 16:4096 15:8192" REPS=1 bash $R/tools/ab_quad.sh run old base || exit 1
for name in old base; do
  if [ "$name" = base ]; then export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so; else export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/build/variants/libq_$name.so; fi
  for sz in 14:16384 15:8192 16:4096; do
    timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f64 "$name f64" || exit 1
  done
done
SIZES="20:512 18:1024 16:4096" REPS=1 bash $R/tools/ab_quad.sh run old base || exit 1
FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so timeout -k 10 120 python3 $R/tools/quad_trace.py 512 || exit 1
} 2>&1 | tee $R/gpurun_out/r4_defer.txt
