#!/bin/bash
# Same-box A/B of team_quad_kernel build variants.
#   local:  tools/ab_quad.sh build name1 "-DQUAD_X=.." name2 "-D.." ...     (libraries under fft-implementation-in-c_amd/build/variants/)
#   GPU:    tools/ab_quad.sh run name1 name2 ...   ("base" = the shipped experiments library; SIZES="20:512", REPS=2)
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/fft-implementation-in-c_amd
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p $P/build/variants
  while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift 2
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed $flags -I$R/include -c $P/csrc/fft_team_quad.hip -o $P/build/variants/q_$name.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $P/build/fft_hip_backend_exp.o $P/build/fft_rows_o2.o $P/build/variants/q_$name.o $P/build/fft_gpu.o $P/build/fft_auto.o $P/build/fft_apps.o $P/build/fft_io.o -o $P/build/variants/libq_$name.so -lpthread || exit 1
    echo "built $name ($flags)"
  done
  exit 0
fi
for rep in $(seq 1 ${REPS-2}); do
  for name in "$@"; do
    if [ "$name" = base ]; then export FFT_LIB_PATH=$P/libfft_mi355x_exp.so; else export FFT_LIB_PATH=$P/build/variants/libq_$name.so; fi
    for sz in ${SIZES-20:512}; do
      timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "$name" || exit 1
    done
  done
done
