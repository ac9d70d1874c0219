#!/bin/bash
# Round 4: BASELINE config 5 (Bluestein, n = 1000003 fp64 x 64) with the passes' tiles walked transform-fastest (FFT_HIP_ORDER_A / _B, the
# experiments build): the chirp and FFT(b) tables are shared by the batch -- resident workgroups on the same tile of different transforms read
# the same table lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
{
for rep in 1 2; do
for ab in "-1 0" "32 0" "-1 32" "32 32" "8 8" "64 64"; do
  set -- $ab
  echo "## ORDER_A=$1 ORDER_B=$2"
  FFT_HIP_ORDER_A=$1 FFT_HIP_ORDER_B=$2 timeout -k 10 200 python3 bench.py --workload prime --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-live-traffic 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.2f Gpoint/s, %.3f ms/step, check %s' % (d['value'], d['ms_per_step'], d['check'].get('ok')))
for p in d['roofline']['per_pass']: print('   ', p['kernel'], p['launches_per_step'], '%.1f us' % (p['avg_launch_ms']*1e3))
" || exit 1
done
done
} 2>&1 | tee gpurun_out/r4_prime_order.txt
