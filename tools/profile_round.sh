#!/bin/bash
# Round profile: kernel trace + stats of the default bench command, then separate --pmc passes (never combined
# with tracing) for HBM traffic and the SQ/LDS picture.  Writes under gpurun_out/; summaries get copied to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_trace --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_trace_bench.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/${tag}_pmc_$i --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-check --no-cpu-baseline > $R/gpurun_out/${tag}_pmc_$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
R="$R"; tag="$tag"
out={"kernel_stats":[], "pmc":{}}
for f in glob.glob(R+"/gpurun_out/%s_trace/**/*kernel_stats.csv"%tag, recursive=True):
    for row in csv.DictReader(open(f)):
        if "fftk::" in row["Name"]:
            out["kernel_stats"].append({k: row[k] for k in ("Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs")})
for i in range(1,6):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(R+"/gpurun_out/%s_pmc_%d/**/*counter_collection.csv"%(tag,i), recursive=True):
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]
            if "fftk::" not in k: continue
            k=k[k.index("fftk::"):k.index(">")+1]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    for k,v in agg.items():
        out["pmc"].setdefault(k,{}).update({c: val/cnt[(k,c)] for c,val in v.items()})
json.dump(out, open(R+"/gpurun_out/%s_summary.json"%tag,"w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
tail -1 $R/gpurun_out/${tag}_trace_bench.log | cut -c1-1500
