#!/bin/bash
# hardware-queue binding experiment: k idle streams created in front of the handle's extra streams ($ORBFE_STREAM_SKEW)
for rep in 1 2; do for k in 0 1 2 3 4 5 6; do
  ORBFE_STREAM_SKEW=$k python3 bench.py --full-line --no-detail --workload ${1:-kitti} --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache > gpurun_out/b_d.json 2>gpurun_out/b_d.err
  python3 - "$k" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/b_d.json").read().strip().splitlines()[-1])
print("[skew %s] value %8.0f ms/step %.3f" % (sys.argv[1], j["value"], j["ms_per_step"]))
PY
done; done
