#!/bin/bash
# membench5 only: timings, then memory-side FETCH / WRITE bytes per dispatch (separate --pmc passes): do the hand-over
# windows' dirty lines get written back when the window is small against the 4 MiB L2?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/l2probe2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -x $R/tools/membench5 ] || hipcc --offload-arch=gfx950 -O3 $R/tools/membench5.hip -o $R/tools/membench5 || exit 1
timeout -k 10 300 $R/tools/membench5 > $O/membench5.txt 2>&1 || { tail -3 $O/membench5.txt; exit 1; }
for set in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $set -d $O/mb5_$set --output-format csv -- $R/tools/membench5 > $O/mb5_$set.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections
O="$O"
lines=[l for l in open(O+"/membench5.txt") if "us per iteration" in l]
res={}
for tag in ("FETCH_SIZE","WRITE_SIZE"):
    d=collections.defaultdict(float)
    for f in glob.glob(O+"/mb5_%s/**/*counter_collection.csv"%tag, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_phase" in row["Kernel_Name"]:
                d[int(row["Dispatch_Id"])]+=float(row["Counter_Value"])
    rows=[v for _,v in sorted(d.items())]
    res[tag]=rows
print("dispatches", len(res["FETCH_SIZE"]), len(res["WRITE_SIZE"]), "lines", len(lines))
# every configuration = warm-up dispatch + timed dispatch; report the timed one
f=res["FETCH_SIZE"][1::2]; w=res["WRITE_SIZE"][1::2]
for i,l in enumerate(lines):
    if i < len(f) and i < len(w):
        print(l.rstrip()[:150], "| FETCH %.2f GB WRITE %.2f GB" % (f[i]/1e6*2.048/1e3*1e0, w[i]/1e6*1.024/1e3))
PY
