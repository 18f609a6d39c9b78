#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh path/to/liborbfe_A.so [workloads]   (B = the in-tree build)
# prints value, ms/step and the exclusive stage times of every run (2 rounds, A and B alternating)
A=$1; WL=${2:-"kitti tum euroc"}
for rep in 1 2; do
  for lib in "$A" ""; do
    for w in $WL; do
      ORBFE_LIB=$lib python bench.py --full-line --no-detail --workload $w --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache > gpurun_out/b_abl.json 2> gpurun_out/b_abl.err
      python - "$w" "${lib:-in-tree}" <<'PY'
import json, sys, os
j = json.loads(open("gpurun_out/b_abl.json").read().strip().splitlines()[-1])
st = j["roofline"]["stages"]
print("[%-10s %-7s] value %8.0f  ms/step %7.3f | excl " % (os.path.basename(sys.argv[2])[:10], sys.argv[1], j["value"], j["ms_per_step"]) +
      "  ".join("%s %.3f" % (k, v["ms_per_step_exclusive"]) for k, v in st.items()))
PY
    done
  done
done
