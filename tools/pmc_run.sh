#!/bin/bash
# usage: tools/pmc_run.sh <tag> [env assignments...]   -- four rocprofv3 --pmc passes over a 2-step bench run
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" ; do
  i=$((i+1))
  env "$@" rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${tag}_$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-check --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for i in range(1,5):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob("$R/gpurun_out/pmc_${tag}_%d/**/*counter_collection.csv"%i, recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]
            if "tile_fft" not in k: continue
            k=k[k.index("tile_fft"):k.index(">")+1]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"])
            cnt[(k,row["Counter_Name"])]+=1
    for k,v in agg.items():
        print(i,k," ".join("%s=%.4g"%(c, val/cnt[(k,c)]) for c,val in v.items()))
PY
