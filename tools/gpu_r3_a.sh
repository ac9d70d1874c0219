#!/bin/bash
# round 3, first look at team_quad_kernel: parity spot check + timing against team_defer_kernel on the same box, timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
export FFT_LIB_PATH=$R/fft-implementation-in-c_amd/libfft_mi355x_exp.so
{
timeout -k 10 120 python3 tools/team_time.py 20 512 f32 quad || echo "QUAD FAILED rc=$?"
FFT_HIP_TEAM_QUAD=0 timeout -k 10 120 python3 tools/team_time.py 20 512 f32 defer
timeout -k 10 120 python3 tools/team_time.py 20 512 f32 quad
FFT_HIP_TEAM_NT=0 timeout -k 10 120 python3 tools/team_time.py 20 512 f32 quad-nt0
FFT_HIP_TEAM_NT=7 timeout -k 10 120 python3 tools/team_time.py 20 512 f32 quad-nt7
timeout -k 10 120 python3 tools/team_time.py 20 128 f32 quad-b128
timeout -k 10 120 python3 tools/quad_trace.py 512
} > gpurun_out/r3_a.txt 2>&1
tail -60 gpurun_out/r3_a.txt
