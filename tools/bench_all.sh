#!/bin/bash
# Runs on the GPU box: the full GPU test suite, then bench.py on all five BASELINE configs (exact traversal, with the
# cpu_baseline leg), the pruned bunny line and the progressive (interactive-shape) lines.  -> gpurun_out/<tag>_*
TAG=${1:-r03}
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.log
for S in cbox bunny scene1 buddha_standin dragon_standin; do
  python3 bench.py --scene $S > gpurun_out/${TAG}_bench_$S.json 2> gpurun_out/${TAG}_bench_$S.err || echo "bench $S failed"
  python3 -c "import json,sys; d=json.load(open('gpurun_out/${TAG}_bench_$S.json')); print('$S', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('physical_bound'), d['roofline'].get('physical_frac'), d.get('cpu_baseline',{}).get('value'), d.get('parity_rows'))"
done
python3 bench.py --scene bunny --traversal pruned --no-cpu-baseline > gpurun_out/${TAG}_bench_bunny_pruned.json 2>/dev/null
# the same configs on the caller's (reference median-split) tree instead of the library's internal one: same images, more node visits
for S in cbox bunny buddha_standin dragon_standin; do
  python3 bench.py --scene $S --tree caller --no-cpu-baseline > gpurun_out/${TAG}_bench_${S}_callers_tree.json 2>/dev/null
  python3 -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_${S}_callers_tree.json')); print('$S caller tree', d['ms_per_step'], d['value'])"
done
# interactive shape: 2 samples per frame; a host sync per frame, then a display that runs one frame behind the renderer
for LAG in 0 1; do for S in scene1 cbox bunny; do python3 bench.py --scene $S --progressive 2 --steps 400 --warmup 20 --display-lag $LAG; done; done > gpurun_out/${TAG}_bench_progressive.jsonl 2>/dev/null
cat gpurun_out/${TAG}_bench_progressive.jsonl | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['metric'][:60], 'display lag', d['config'].get('display_lag'), d['value'])"
