#!/bin/bash
# Same-box A/B of kernel variants selected by environment switches (the -DFFT_EXPERIMENTS library reads them once per
# process):  tools/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...   with SIZES="20:512 19:1024" (log2n:batch) and REPS=2
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
for rep in $(seq 1 ${REPS-2}); do
  for variant in "$@"; do
    for sz in ${SIZES-20:512 19:1024}; do
      env $variant python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} ${PREC-f32} "$variant" || exit 1
    done
  done
done
