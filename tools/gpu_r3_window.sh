#!/bin/bash
# What a smaller exchange window would be worth at n = 2^20 / 2^19 (two-slot protocol): FFT_HIP_TEAM_ABLATE=16: the second slot aliases half of
# the first (3 MiB of window per XCD), 32: all of it (2 MiB); results invalid.  Time + memory-side traffic (rocprofv3 --pmc, separate passes).
R=${GRAFT_REPO_ROOT:-/root/repo}
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
export AB_NOCHECK=1
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for ab in 0 16 32; do
    for sz in 20:512 19:1024; do
      FFT_HIP_TEAM_ABLATE=$ab timeout -k 10 120 python3 $R/tools/team_time.py ${sz%%:*} ${sz##*:} f32 "window ablate=$ab" || exit 1
    done
  done
done
for ab in 0 16 32; do
  export FFT_HIP_TEAM_ABLATE=$ab
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $set -d $R/gpurun_out/win_pmc/${ab}_$set --output-format csv -- python3 $R/tools/team_time.py 20 512 f32 "pmc" > /dev/null 2>&1 || { echo "pmc $ab $set failed"; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for ab in (0, 16, 32):
    out = {}
    for cset in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, n = 0.0, 0
        for f in glob.glob("%s/gpurun_out/win_pmc/%d_%s/**/*counter_collection.csv" % (R, ab, cset), recursive=True):
            for row in csv.DictReader(open(f)):
                if "team_quad_kernel" in row["Kernel_Name"] and row["Counter_Name"] == cset:
                    tot += float(row["Counter_Value"]); n += 1
        out[cset] = tot / max(n, 1)
    print("ablate=%d: reads %.2f GB (FETCH_SIZE x 2), writes %.2f GB per launch of 512 transforms (algorithmic 4.29 + 4.29)" % (ab, out["FETCH_SIZE"] * 2.048e-6, out["WRITE_SIZE"] * 1.024e-6))
PY
