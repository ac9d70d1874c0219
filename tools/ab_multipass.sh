export FFT_HIP_TEAM=0
for rep in 1 2; do
for lib in A B; do
  if [ $lib = B ]; then export FFT_LIB_PATH=$PWD/tools/libfft_variant_b.so; else unset FFT_LIB_PATH; fi
  for w in 1k 64k 256k 1m; do
    v=$(python bench.py --workload $w --no-cpu-baseline --no-check --steps 20 2>/dev/null | python -c "import sys,json; print('%.1f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$lib $w $v"
  done
done
done
