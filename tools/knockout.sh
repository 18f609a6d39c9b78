#!/bin/bash
# How much does each stage add to the PIPELINED step?  Every run skips the launches of one stage ($ORBFE_KNOCKOUT, after the
# warm-up chunks ran in full; the downstream stages work on the previous step's -- real -- data) and reports value and
# ms/step next to the full pipeline.  Timing experiment: outputs unchecked.   usage: tools/knockout.sh [workload]
WL=${1:-kitti}
mkdir -p gpurun_out
OUT=gpurun_out/knockout_$WL.txt
: > $OUT
run() {
  local label=$1 mask=$2
  ORBFE_KNOCKOUT=$mask ORBFE_KNOCKOUT_AFTER=24 ORBFE_BENCH_NO_CHECK=1 timeout -k 10 200 python bench.py --full-line --no-detail --workload $WL --no-cpu-baseline --no-e2e --no-latency \
      --input-cache /tmp/orbfe_ab_cache > gpurun_out/_ko.json 2>> gpurun_out/_ko.err
  local rc=$?
  if [ $rc -ne 0 ] || [ ! -s gpurun_out/_ko.json ]; then printf "%-40s FAILED rc=%d; no further runs in this call\n" "$label" $rc | tee -a $OUT; exit $rc; fi
  python - "$label" >> $OUT <<'PY'
import json, sys
j = json.loads(open("gpurun_out/_ko.json").read().strip().splitlines()[-1])
print("%-40s value %8.0f  ms/step %6.3f" % (sys.argv[1], j["value"], j["ms_per_step"]))
PY
  tail -1 $OUT
}
run "full pipeline" 0
run "without pyramid + blur" 1
run "without FAST" 2
run "without gather + octree" 4
run "without orientation + descriptors" 8
run "without the stereo matcher" 16
run "without octree, orient_desc, stereo" 28
run "without pyramid + blur and FAST" 3
run "full pipeline (again)" 0
