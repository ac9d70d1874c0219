#!/bin/bash
# Round 4: column chunk a + 2 requested as soon as chunk a's image is free, on EVERY quad shape (QUAD_COL_AHEAD=2; shipped: teams of 32 only),
# every transform checked; fp32 2^15 ... 2^20 and fp64 2^14 ... 2^16
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
P=$R/fft-implementation-in-c_amd
{
for rep in 1 2 3; do
  for name in base ah2; do
    if [ "$name" = base ]; then export FFT_LIB_PATH=$P/libfft_mi355x_exp.so; else export FFT_LIB_PATH=$P/build/variants/libq_$name.so; fi
    for sz in 19:1024 18:1024 17:2048 16:4096 15:8192; do
      timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "$name" || exit 1
    done
    for sz in 16:2048 15:4096 14:8192; do
      timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f64 "$name" || exit 1
    done
  done
done
} 2>&1 | tee $R/gpurun_out/r4_ahead2.txt
