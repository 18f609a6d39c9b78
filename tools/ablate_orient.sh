#!/bin/bash
# k_orient_desc with parts switched off ($ORBFE_ORIENT_ABLATE: 1 no moment-row loads, 2 no patch loads, 4 no LDS sampling):
# exclusive stage time, live time and the pipelined step.  Timing experiment, outputs unchecked.  usage: ablate_orient.sh [workload]
WL=${1:-kitti}
mkdir -p gpurun_out
OUT=gpurun_out/ablate_orient_$WL.txt
: > $OUT
run() {
  local label=$1; shift
  env "$@" ORBFE_BENCH_NO_CHECK=1 timeout -k 10 200 python bench.py --full-line --no-detail --workload $WL --no-cpu-baseline --no-e2e --no-latency \
      --input-cache /tmp/orbfe_ab_cache > gpurun_out/_ao.json 2>> gpurun_out/_ao.err
  local rc=$?
  if [ $rc -ne 0 ] || [ ! -s gpurun_out/_ao.json ]; then printf "%-40s FAILED rc=%d; no further runs in this call\n" "$label" $rc | tee -a $OUT; exit $rc; fi
  python - "$label" >> $OUT <<'PY'
import json, sys
j = json.loads(open("gpurun_out/_ao.json").read().strip().splitlines()[-1])
st = j["roofline"]["stages"]["orient_desc"]
print("%-40s orient_desc excl %.3f ms  live %.3f  | value %.0f  ms/step %.3f" % (sys.argv[1], st["ms_per_step_exclusive"], st["ms_per_step_live"], j["value"], j["ms_per_step"]))
PY
  tail -1 $OUT
}
run "full kernel" ORBFE_ORIENT_ABLATE=0
run "no moment-row loads" ORBFE_ORIENT_ABLATE=1
run "no patch loads" ORBFE_ORIENT_ABLATE=2
run "no loads at all" ORBFE_ORIENT_ABLATE=3
run "no loads, no LDS sampling" ORBFE_ORIENT_ABLATE=7
run "full kernel, uncapped grid" ORBFE_ORIENT_GRID=0
run "no loads at all, uncapped grid" ORBFE_ORIENT_ABLATE=3 ORBFE_ORIENT_GRID=0
run "full kernel, 2 per CU" ORBFE_ORIENT_GRID=2
