# the switches below exist only in the -DFFT_EXPERIMENTS build of the library
export FFT_LIB_PATH=${FFT_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/fft-implementation-in-c_amd/libfft_mi355x_exp.so}
for rep in 1 2; do
for mb in 1024 2048 4096; do
  for w in 1m 64k 256k; do
    v=$(FFT_HIP_CHUNK_MB=$mb python bench.py --workload $w --no-cpu-baseline --no-check --steps 20 2>/dev/null | python -c "import sys,json; print('%.1f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    m=$(FFT_HIP_TEAM=0 FFT_HIP_CHUNK_MB=$mb python bench.py --workload $w --no-cpu-baseline --no-check --steps 20 2>/dev/null | python -c "import sys,json; print('%.1f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "chunk_mb $mb $w team $v multipass $m"
  done
done
done
