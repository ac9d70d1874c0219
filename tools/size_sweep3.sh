#!/bin/bash
# every power-of-two size at 2 GiB per execute, default policy (what profiles/r2_size_sweep.txt held in round 2): tools/size_sweep3.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
f32=""; for l in 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 24; do f32="$f32 $((1<<l)):$((1<<(28-l))):f32"; done
f64=""; for l in 6 8 10 12 13 14 15 16 17 18 19 20 21; do f64="$f64 $((1<<l)):$((1<<(27-l))):f64"; done
for rep in 1 2; do
  python3 tools/time_sizes.py r3 team=1 $f32 | tr '|' '\n'
  python3 tools/time_sizes.py r3 team=1 $f64 | tr '|' '\n'
done
