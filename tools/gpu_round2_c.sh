set -e
mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -x -q > gpurun_out/r2d/gpu_tests.log 2>&1 || { tail -60 gpurun_out/r2d/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r2d/gpu_tests.log
python tools/team_trace2.py 20 512 > gpurun_out/r2d/trace_20.txt 2>&1 || true
python bench.py --steps 20 --warmup 3 > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err || true
python -c "
import json
r=json.loads(open('gpurun_out/r2d/bench.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','value_median','ms_per_step','ms_per_step_median','ms_per_step_min')}, r['roofline']['frac'], r['roofline'].get('copy_gbs'), r['roofline'].get('frac_of_copy'), r.get('secondary_fp64'), (r.get('cpu_baseline') or {}).get('value'))
"
cat gpurun_out/r2d/trace_20.txt
