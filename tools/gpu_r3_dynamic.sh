#!/bin/bash
# Same-box A/B: team_quad_kernel with the static split of the batch (FFT_HIP_TEAM_DYNAMIC=0) and with claimed transforms (1)
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
run() { timeout -k 10 120 python3 $R/tools/team_time.py "$@" || exit 1; }
for rep in $(seq 1 ${REPS-3}); do
  for d in ${DYN-0 1}; do
    export FFT_HIP_TEAM_DYNAMIC=$d
    run 20 512 f32 "2^20 dynamic=$d"
    run 19 1024 f32 "2^19 dynamic=$d"
    run 18 2048 f32 "2^18 dynamic=$d"
    [ -n "$BIG" ] || run 17 4096 f32 "2^17 dynamic=$d"
    [ -n "$BIG" ] || run 16 8192 f32 "2^16 dynamic=$d"
  done
done
