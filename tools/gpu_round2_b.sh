set -e
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_ext.py -x -q > gpurun_out/r2b/gpu_ext.log 2>&1 || { tail -60 gpurun_out/r2b/gpu_ext.log; exit 1; }
tail -3 gpurun_out/r2b/gpu_ext.log
python bench.py --workload prime --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench_prime.json 2> gpurun_out/r2b/bench_prime.err || true
tail -c 600 gpurun_out/r2b/bench_prime.json
python -m pytest tests -m gpu -x -q > gpurun_out/r2b/gpu_tests.log 2>&1 || { tail -60 gpurun_out/r2b/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r2b/gpu_tests.log
