#!/bin/bash
# Exclusive time of the orientation + descriptor stage (tile form) with parts of the kernel switched off
# ($ORBFE_DESC_TILES_ABLATE bits: 1 moments, 2 atan2/sincos, 4 descriptors, 8 blurred-tile loads, 16 level-tile loads).
# Outputs are wrong by construction: timing experiment only (ORBFE_BENCH_NO_CHECK).  usage: ablate_desc_tiles.sh [workload]
set -u
WL=${1:-kitti}
mkdir -p gpurun_out
OUT=gpurun_out/ablate_desc_tiles_$WL.txt
: > $OUT
run() {  # label, env...
  local label=$1; shift
  rm -f gpurun_out/_abl.json
  env "$@" ORBFE_BENCH_NO_CHECK=1 timeout -k 10 200 python bench.py --full-line --no-detail --workload $WL --steps 10 --min-seconds 0.2 --no-cpu-baseline --no-e2e \
      --input-cache /tmp/orbfe_cache > gpurun_out/_abl.json 2>> gpurun_out/_abl.err
  local rc=$?
  if [ $rc -ne 0 ] || [ ! -s gpurun_out/_abl.json ]; then   # a dead run is a ROW of the table, and the last GPU step of this call
    printf "%-44s FAILED rc=%d (see gpurun_out/_abl.err); no further runs in this call\n" "$label" $rc | tee -a $OUT
    exit $rc
  fi
  python - "$label" >> $OUT <<'PY'
import json, sys
j = json.loads(open("gpurun_out/_abl.json").read().strip().splitlines()[-1])
st = j["roofline"]["stages"]
print("%-44s orient_desc excl %.3f ms  live %.3f  | value %.0f  ms/step %.3f" % (sys.argv[1], st["orient_desc"]["ms_per_step_exclusive"],
      st["orient_desc"]["ms_per_step_live"], j["value"], j["ms_per_step"]))
PY
  tail -1 $OUT
}
run "per-keypoint form (k_orient_desc)" ORBFE_DESC_TILES=0
run "tile form" ORBFE_DESC_TILES=1
run "tile form, no moments" ORBFE_DESC_TILES=1 ORBFE_DESC_TILES_ABLATE=1
run "tile form, no atan2/sincos" ORBFE_DESC_TILES=1 ORBFE_DESC_TILES_ABLATE=2
run "tile form, no descriptors" ORBFE_DESC_TILES=1 ORBFE_DESC_TILES_ABLATE=4
run "tile form, no moments/sincos/descriptors" ORBFE_DESC_TILES=1 ORBFE_DESC_TILES_ABLATE=7
run "tile form, no loads" ORBFE_DESC_TILES=1 ORBFE_DESC_TILES_ABLATE=24
run "tile form, nothing but the list scan" ORBFE_DESC_TILES=1 ORBFE_DESC_TILES_ABLATE=31
for g in 1 2 3 4; do run "tile form, $g workgroups per CU" ORBFE_DESC_TILES=1 ORBFE_DESC_TILES_GRID=$g; done
