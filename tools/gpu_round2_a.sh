set -e
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r2a/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r2a/gpu_tests.log
python tools/team_trace2.py 20 512 > gpurun_out/r2a/trace_20.txt 2>&1 || true
python tools/team_trace2.py 18 2048 > gpurun_out/r2a/trace_18.txt 2>&1 || true
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err || true
tail -c 1500 gpurun_out/r2a/bench.json
cat gpurun_out/r2a/trace_20.txt
