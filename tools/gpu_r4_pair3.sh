#!/bin/bash
# Round 4: in-kernel timelines (tools/quad_trace.py) of the two-slot kernel, the one-slot team protocol, the pair protocol and the pair protocol
# without its guards (timing only)
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/fft-implementation-in-c_amd
cd /tmp && export TMPDIR=/tmp
export FFT_HIP_TEAM_DYNAMIC=0
mkdir -p $R/gpurun_out
{
for v in "2 $P/libfft_mi355x_exp.so" "1 $P/libfft_mi355x_exp.so" "3 $P/libfft_mi355x_exp.so" "3 $P/build/variants/libq_pg.so" "3 $P/build/variants/libq_pgw.so"; do
  set -- $v
  echo "#### FFT_HIP_QUAD_SLOTS=$1 $(basename $2)"
  FFT_HIP_QUAD_SLOTS=$1 FFT_LIB_PATH=$2 timeout -k 10 120 python3 $R/tools/quad_trace.py 512 || exit 1
done
} 2>&1 | tee $R/gpurun_out/r4_pair3.txt
