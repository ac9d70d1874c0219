#!/bin/bash
# What evicts the hand-over window from L2?  HBM-side traffic of team_fft_kernel (FFT_HIP_TEAM_DEFER=0) with its
# streams switched off one by one (FFT_HIP_TEAM_ABLATE 4 = no result stores, 8 = no column-tile DMA) and with the
# non-temporal bits (FFT_HIP_TEAM_NT).  Results are garbage under ablation: --no-check.  Output: gpurun_out/ablate/.
# the switches below exist only in the -DFFT_EXPERIMENTS build of the library
export FFT_LIB_PATH=${FFT_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/fft-implementation-in-c_amd/libfft_mi355x_exp.so}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ablate
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FFT_HIP_TEAM_DEFER=${DEFER-0}
for cfg in ${CFGS-0:0 0:4 0:8 0:12 3:0 3:4 3:8}; do
  export FFT_HIP_TEAM_NT=${cfg%%:*} FFT_HIP_TEAM_ABLATE=${cfg##*:}
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $set -d $O/c_${cfg/:/_}_$set --output-format csv -- python3 $R/bench.py --workload ${WL-1m} --steps 3 --warmup 1 --no-check --no-cpu-baseline --no-live-traffic > $O/c_${cfg/:/_}_$set.log 2>&1 || exit 1
  done
  timeout -k 10 300 python3 $R/bench.py --workload ${WL-1m} --steps 10 --warmup 2 --no-check --no-cpu-baseline > $O/c_${cfg/:/_}_bench.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections,json,os
O="$O"
for d in sorted(glob.glob(O+"/c_*_bench.log")):
    cfg=os.path.basename(d)[2:-10]
    out={}
    for s,f in (("FETCH_SIZE",2.048e-6),("WRITE_SIZE",1.024e-6)):
        v=[]
        for c in glob.glob(O+"/c_%s_%s/**/*counter_collection.csv"%(cfg,s), recursive=True):
            for row in csv.DictReader(open(c)):
                if "team_" in row["Kernel_Name"]: v.append(float(row["Counter_Value"]))
        out[s]=round(sum(v)/max(1,len(v))*f,2)
    try: ms=json.loads(open(d).read().strip().splitlines()[-1])["ms_per_step"]
    except Exception as e: ms=None
    print("nt_ablate", cfg, out, "GB per launch (algorithmic 4.29 + 4.29); ms_per_step", ms)
PY
