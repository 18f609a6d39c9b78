#!/bin/bash
# k_blur7<.., RESIZE> with parts switched off ($ORBFE_BLUR_ABLATE: 1 no blurred-level stores, 2 no next-level stores, 4 no staging
# loads, 8 no resize at all): exclusive stage time and the pipelined step.  Timing experiment: later stages see stale levels
# (the previous, full, step's), outputs unchecked.  usage: ablate_blur.sh [workload]
WL=${1:-kitti}
mkdir -p gpurun_out
OUT=gpurun_out/ablate_blur_$WL.txt
: > $OUT
run() {
  local label=$1; shift
  env "$@" ORBFE_BENCH_NO_CHECK=1 timeout -k 10 200 python bench.py --full-line --no-detail --workload $WL --no-cpu-baseline --no-e2e --no-latency \
      --input-cache /tmp/orbfe_ab_cache > gpurun_out/_ab.json 2>> gpurun_out/_ab.err
  local rc=$?
  if [ $rc -ne 0 ] || [ ! -s gpurun_out/_ab.json ]; then printf "%-44s FAILED rc=%d; no further runs in this call\n" "$label" $rc | tee -a $OUT; exit $rc; fi
  python - "$label" >> $OUT <<'PY'
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1])
st = j["roofline"]["stages"]["pyramid"]
print("%-44s pyramid+blur excl %.3f ms  live %.3f  | value %.0f  ms/step %.3f" % (sys.argv[1], st["ms_per_step_exclusive"], st["ms_per_step_live"], j["value"], j["ms_per_step"]))
PY
  tail -1 $OUT
}
run "full kernel" ORBFE_BLUR_ABLATE=0
run "no blurred-level stores" ORBFE_BLUR_ABLATE=1
run "no next-level stores" ORBFE_BLUR_ABLATE=2
run "no stores at all" ORBFE_BLUR_ABLATE=3
run "no staging loads" ORBFE_BLUR_ABLATE=4
run "no resize (arithmetic and stores)" ORBFE_BLUR_ABLATE=8
run "no loads, no stores" ORBFE_BLUR_ABLATE=7
run "no loads, no stores, no resize" ORBFE_BLUR_ABLATE=15
