# same-box A/B of library builds: tools/ab_libs.sh <lib or "-" for the shipped one> ...   (FFT_LIB_PATH per run)
export SWEEP_QUICK=1
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset FFT_LIB_PATH; else export FFT_LIB_PATH=$PWD/$lib; fi
  echo "== $lib"
  python3 tools/team_sweep.py f32 ${SIZES-18 19 20}
done
done
