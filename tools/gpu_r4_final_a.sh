#!/bin/bash
# Round 4, validation part A: the write-port microbenchmark, the stream yardsticks with the tile-wise shapes, the whole GPU test suite
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 120 ./tools/membench7 > gpurun_out/r4_membench7.txt 2>&1 || { tail -5 gpurun_out/r4_membench7.txt; exit 1; }
cat gpurun_out/r4_membench7.txt
timeout -k 10 200 python3 tools/stream_bench.py 1024 2048 > gpurun_out/r4_stream.txt 2>&1 || { tail -5 gpurun_out/r4_stream.txt; exit 1; }
cat gpurun_out/r4_stream.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4_gputests.txt 2>&1; rc=$?
tail -5 gpurun_out/r4_gputests.txt
exit $rc
