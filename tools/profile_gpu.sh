#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench command.
# Usage: tools/profile_gpu.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r03}; shift
ARGS="$@"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# kernel trace: the default bench command (20 steps, 3 warm-up) so that its average agrees with bench.py's own HIP-event time;
# counter passes: 5 steps are enough (every launch is identical)
TRACE="python3 bench.py --no-cpu-baseline --no-work-frames $ARGS"
BENCH="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-work-frames $ARGS"
echo "== kernel trace" | tee $OUT/log.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE > $OUT/bench_trace.out 2>> $OUT/log.txt || exit 1
grep '^{' $OUT/bench_trace.out | tail -1 > $OUT/bench_line.json
if [ -n "$PT_PROFILE_TRACE_ONLY" ]; then      # only the per-kernel durations (the counter passes of an unchanged kernel need no rerun)
  find $OUT/trace -type f ! -name "*_kernel_stats.csv" -delete
  exit 0
fi
# Every pass under its own timeout (a pass that asks for more counters than a block has slots aborts and then hangs).
# TCP_TOTAL_CACHE_ACCESSES = L1 tag lookups (one per active lane and 16-B load of a divergent access): the rate that binds scenes in
# global memory.  TCC: 4 slots (FETCH_SIZE takes 3, WRITE_SIZE 2), SQ: 8, GRBM: 2.  TCC_EA0_RDREQ_{32B,64B,128B} split the L2's fabric-side
# read requests by size, which settles what FETCH_SIZE (= requests x 64 B) leaves open (MI355X_MICROARCH.md, HBM section).
for PASS in "FETCH_SIZE" "WRITE_SIZE" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
            "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TOTAL_CACHE_ACCESSES_sum"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-60)
  echo "== pmc $PASS" | tee -a $OUT/log.txt
  timeout -k 10 400 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$NAME -- $BENCH >> $OUT/log.txt 2>&1 || echo "pass failed: $PASS" | tee -a $OUT/log.txt
done
find $OUT -name "*.csv" | head -50
