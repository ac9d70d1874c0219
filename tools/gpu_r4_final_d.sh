#!/bin/bash
# Round 4, validation of the build with the pair protocol as the default at n = 2^20 and 2^19: the GPU suite, the soak, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4_gputests_final.txt 2>&1; rc=$?
tail -4 gpurun_out/r4_gputests_final.txt
[ $rc = 0 ] || exit $rc
timeout -k 10 900 python3 tools/quad_soak.py 250 > gpurun_out/r4_quad_soak.txt 2>&1 || { tail -5 gpurun_out/r4_quad_soak.txt; exit 1; }
tail -12 gpurun_out/r4_quad_soak.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.txt 2>&1 || { tail -5 gpurun_out/r4_smoke.txt; exit 1; }
tail -1 gpurun_out/r4_smoke.txt
timeout -k 10 600 python3 bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err || { tail -5 gpurun_out/r4_bench_default.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r4_bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench: value %.1f Gpoint/s ms_per_step %.3f frac %.3f frac_kernel %.3f traffic_over_algorithmic %s (%s) secondary %s cpu %s' % (d['value'], d['ms_per_step'], r['frac'], r['frac_kernel'], r['traffic_over_algorithmic'], r.get('traffic_source'), d['secondary_fp64']['value'], d['cpu_baseline']['value']))"
