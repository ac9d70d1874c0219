#!/bin/bash
# memory-side traffic of team_quad_kernel variants: tools/pmc_ab.sh name ...   (libraries from tools/ab_quad.sh build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/build/variants/libq_$name.so
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/pmc_ab
    AB_NOCHECK=1 timeout -k 10 200 rocprofv3 --pmc $set -d /tmp/pmc_ab --output-format csv -- python3 $R/tools/team_time.py ${LOG2N-20} ${BATCH-512} f32 $name > /tmp/pmc_ab.log 2>&1 || { tail -3 /tmp/pmc_ab.log; exit 1; }
    python3 - "$name" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_ab/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "team_quad" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in agg.items():
    m = sum(v) / len(v)
    extra = ""
    if k == "FETCH_SIZE": extra = " = %.2f GB read (x2 corrected)" % (m * 2.048e-6)
    if k == "WRITE_SIZE": extra = " = %.2f GB written" % (m * 1.024e-6)
    print("%-8s %-14s %14.0f per launch (%d launches)%s" % (sys.argv[1], k, m, len(v), extra), flush=True)
PY
  done
done
