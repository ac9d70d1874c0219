#!/bin/bash
# L2-residency probe: membench5 timings, then its HBM-side traffic per dispatch (separate --pmc passes), then the
# team kernel's traffic with and without the non-temporal bits on its HBM streams.  Writes under gpurun_out/.
# the switches below exist only in the -DFFT_EXPERIMENTS build of the library
export FFT_LIB_PATH=${FFT_LIB_PATH:-${GRAFT_REPO_ROOT:-/root/repo}/fft-implementation-in-c_amd/libfft_mi355x_exp.so}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/l2probe
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 $R/tools/membench5 > $O/membench5.txt 2>&1 &&
for set in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $set -d $O/mb5_$set --output-format csv -- $R/tools/membench5 > $O/mb5_$set.log 2>&1 || exit 1
done &&
for nt in ${NTS-0 3}; do
  export FFT_HIP_TEAM_DEFER=0 FFT_HIP_TEAM_NT=$nt
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $set -d $O/team_nt${nt}_$set --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-check --no-cpu-baseline --no-live-traffic > $O/team_nt${nt}_$set.log 2>&1 || exit 1
  done
  timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/team_nt${nt}_bench.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections
O="$O"
for tag in ("mb5_FETCH_SIZE","mb5_WRITE_SIZE"):
    d=collections.defaultdict(float)
    for f in glob.glob(O+"/%s/**/*counter_collection.csv"%tag, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_phase" in row["Kernel_Name"]:
                d[int(row["Dispatch_Id"])]+=float(row["Counter_Value"])
    rows=sorted(d.items())
    print(tag, [round(v/1e6*(2.048 if "FETCH" in tag else 1.024),2) for _,v in rows[1::2]], "GB")
for nt in [int(x) for x in "${NTS-0 3}".split()]:
    for s in ("FETCH_SIZE","WRITE_SIZE"):
        agg=collections.defaultdict(list)
        for f in glob.glob(O+"/team_nt%d_%s/**/*counter_collection.csv"%(nt,s), recursive=True):
            for row in csv.DictReader(open(f)):
                if "team_" in row["Kernel_Name"]: agg[row["Kernel_Name"][:50]].append(float(row["Counter_Value"]))
        print("team nt",nt,s,{k:[round(x/1e6,3) for x in v] for k,v in agg.items()})
PY
cat $O/membench5.txt
for nt in ${NTS-0 3}; do tail -1 $O/team_nt${nt}_bench.log | cut -c1-300; done
