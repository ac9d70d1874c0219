#!/bin/bash
# Round 4: soak of the final build (the deferred-store schedule at n = 2^20 among the shapes): 250 executes per shape, every transform bit-identical to the first, verified result
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 900 python3 tools/quad_soak.py 250 2>&1 | tee gpurun_out/r4_quad_soak.txt
