#!/bin/bash
# L2 memory-side traffic (FETCH_SIZE x2 for 128-byte requests, WRITE_SIZE) and L2 hit counters of the team kernel for
# kernel variants selected by environment switches:  tools/pmc_variants.sh "VAR=a" "VAR=b" ...   SIZE=20:512
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmcv
rm -rf $O; mkdir -p $O
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
SZ=${SIZE-20:512}
cd /tmp && export TMPDIR=/tmp
i=0
for variant in "$@"; do
  i=$((i+1))
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    ( export $variant; timeout -k 10 240 rocprofv3 --pmc $set -d $O/v${i}_$tag --output-format csv -- python3 $R/tools/team_time.py ${SZ%%:*} ${SZ##*:} f32 "v$i" > $O/v${i}_$tag.log 2>&1 ) || { echo "failed: $variant $set"; tail -5 $O/v${i}_$tag.log; exit 1; }
  done
done
python3 - "$@" <<PY
import csv,glob,sys,collections
O="$O"
alg = 2.0 * (1 << ${SZ%%:*}) * ${SZ##*:} * 8 / 1e9
for i,variant in enumerate(sys.argv[1:],1):
    out={}
    for tag in ("FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum"):
        agg=collections.defaultdict(list)
        for c in glob.glob(O+"/v%d_%s/**/*counter_collection.csv"%(i,tag), recursive=True):
            for row in csv.DictReader(open(c)):
                if "team_" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k,v in agg.items(): out[k]=sum(v)/len(v)
    f=out.get("FETCH_SIZE",0)*2.048e-6; w=out.get("WRITE_SIZE",0)*1.024e-6
    h=out.get("TCC_HIT_sum",0); m=out.get("TCC_MISS_sum",0)
    print("%-60s FETCH %.2f GB  WRITE %.2f GB  (algorithmic %.2f + %.2f)  traffic/alg %.2f  L2 hit %.1f %%" % (variant, f, w, alg/2, alg/2, (f+w)/alg, 100*h/max(1,h+m)))
PY
