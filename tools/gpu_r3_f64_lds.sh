#!/bin/bash
# LDS bank conflicts of the fp64 quad shapes (rocprofv3 --pmc, one pass): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
cd /tmp && export TMPDIR=/tmp
for sz in 14:8192:f64 16:2048:f64 16:4096:f32; do
  IFS=: read l b t <<< "$sz"
  timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $R/gpurun_out/f64_lds/$l$t --output-format csv -- python3 $R/tools/team_time.py $l $b $t pmc > /dev/null 2>&1 || { echo "pmc failed $sz"; exit 1; }
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for d in sorted(glob.glob(R + "/gpurun_out/f64_lds/*")):
    agg = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "team_quad_kernel" in row["Kernel_Name"]:
                agg[row["Counter_Name"]] += float(row["Counter_Value"])
    print(os.path.basename(d), "bank conflict cycles / LDS active cycles = %.3f; LDS instruction cycles / wave cycles = %.3f" %
          (agg["SQ_LDS_BANK_CONFLICT"] / max(1.0, agg["SQ_LDS_IDX_ACTIVE"]), agg["SQ_ACTIVE_INST_LDS"] / max(1.0, agg["SQ_WAVE_CYCLES"])))
PY
