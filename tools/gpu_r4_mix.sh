#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 200 ./tools/membench8 2>&1 | tee gpurun_out/r4_membench8.txt
