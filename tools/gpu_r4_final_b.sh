#!/bin/bash
# Round 4, validation part B: the whole GPU test suite on the final build, then the size sweep at 2 GiB per execute
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4_gputests.txt 2>&1; rc=$?
tail -5 gpurun_out/r4_gputests.txt
[ $rc = 0 ] || exit $rc
bash tools/size_sweep3.sh > gpurun_out/r4_size_sweep.txt 2>&1 || { tail -5 gpurun_out/r4_size_sweep.txt; exit 1; }
cat gpurun_out/r4_size_sweep.txt
