#!/bin/bash
# Round profile (run on the GPU box): kernel trace + stats of the default bench command, separate --pmc passes (never
# combined with tracing) for memory-side traffic / L2 hits / the SQ-LDS picture, and the bench lines of every workload.
# Everything lands under gpurun_out/<tag>/; tools/collect_profiles.py copies the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r2}
O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-live-traffic > $O/trace_bench.log 2>&1 || { tail -5 $O/trace_bench.log; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $O/pmc_$i --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-check --no-cpu-baseline --no-secondary --no-live-traffic > $O/pmc_$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $O/pmc_$i.log; exit 1; }
done
cd $R
for wl in 1m 64k 256k prime 1k; do
  extra="--no-cpu-baseline"; [ $wl = 1m ] && extra=""
  timeout -k 10 500 python3 bench.py --workload $wl --steps 20 --warmup 3 $extra > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "bench $wl failed"; tail -5 $O/bench_$wl.err; exit 1; }
done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1  # warm
python3 tools/collect_profiles.py $tag
