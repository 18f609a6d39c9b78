#!/bin/bash
# batch-size sweep of one workload (units resident per GPU and processed per step): tools/batch_sweep.sh kitti "512 1024 2048"
cd "$(dirname "$0")/.."
W=${1:-kitti}; BB=${2:-"512 1024 2048"}
for B in $BB; do
  t0=$(date +%s)
  python bench.py --full-line --no-detail --workload $W --no-e2e --no-cpu-baseline --no-latency --batch $B > gpurun_out/b_bs.json 2> gpurun_out/b_bs.err
  t1=$(date +%s)
  python - $B $((t1 - t0)) <<'PY'
import json, sys
j = json.loads(open("gpurun_out/b_bs.json").read().strip().splitlines()[-1])
print("batch %5s  value %8.0f  ms/step %7.3f  spread %.2f %%  (bench wall %s s)" % (sys.argv[1], j["value"], j["ms_per_step"], j["repeats"]["spread_pct"], sys.argv[2]))
PY
done
