#!/bin/bash
# Same-box A/B of the non-square team_quad_kernel shapes (n = 2^19, 2^17) against round 2's team kernel, all transforms checked.
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/fft-implementation-in-c_amd
export FFT_LIB_PATH=$P/libfft_mi355x_exp.so
run() { timeout -k 10 120 python3 $R/tools/team_time.py "$@" || exit 1; }
for rep in 1 2; do
  FFT_HIP_TEAM_QUAD=0 run 19 1024 f32 "2^19 defer (round 2)"
  FFT_HIP_QUAD_SLOTS=2 run 19 1024 f32 "2^19 quad two slots"
  FFT_HIP_QUAD_SLOTS=1 run 19 1024 f32 "2^19 quad one slot"
  FFT_HIP_TEAM_QUAD=0 run 17 4096 f32 "2^17 defer (round 2)"
  run 17 4096 f32 "2^17 quad one slot"
done
run 20 512 f32 "2^20 quad"
run 18 2048 f32 "2^18 quad"
run 16 8192 f32 "2^16 quad"
