#!/bin/bash
# Runs on the GPU box (via gpurun): memory-system PMC passes of the bench command for one scene — what the L1 (TCP),
# the texture addresser (TA), the L2 (TCC) and the L2's fabric side (EA) see per trace_kernel launch.
# Usage: tools/profile_diag.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/diag_*/
set -o pipefail
TAG=${1:-diag}; shift
ARGS="$@"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $ARGS"
# rocprofv3 aborts (and then hangs) when a pass asks for more counters than a block has slots: TA and TD take 2 per
# pass, TCC and TCP 4, SQ 8 — and every pass runs under its own timeout.
for PASS in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "TCC_EA0_RDREQ_DRAM_sum TCC_READ_sum TCC_REQ_sum TCC_READ_SECTORS_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" \
            "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
            "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  NAME=$(echo $PASS | tr ' ' '_' | cut -c1-60)
  echo "== pmc $PASS" | tee -a $OUT/diag_log.txt
  if [ -n "$DIAG_ONLY" ] && ! echo "$PASS" | grep -q "$DIAG_ONLY"; then continue; fi
  timeout -k 10 180 rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/diag_$NAME -- $BENCH >> $OUT/diag_log.txt 2>&1 || echo "pass failed: $PASS" | tee -a $OUT/diag_log.txt
done
python3 - "$OUT" <<'EOF'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "diag_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
json.dump(c, open(os.path.join(out, "diag_counters.json"), "w"), indent=1)
print(json.dumps(c, indent=1))
EOF
