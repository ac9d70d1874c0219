#!/bin/bash
# Same-box A/B of library BUILDS: tools/ab_libs2.sh lib1.so lib2.so ...  (paths relative to fft-implementation-in-c_amd/)
# SIZES="20:512 18:2048" REPS=2 PREC=f32; the libraries must be -DFFT_EXPERIMENTS builds if VARIANT env switches are used
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in $(seq 1 ${REPS-2}); do
  for lib in "$@"; do
    for sz in ${SIZES-20:512 19:1024 18:2048 16:8192}; do
      FFT_LIB_PATH=$R/fft-implementation-in-c_amd/$lib FFT_HIP_TEAM=2 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} ${PREC-f32} "$lib" || exit 1
    done
  done
done
