#!/bin/bash
# same-box A/B of SEVERAL builds of the library: tools/ab_libs.sh "workloads" libA.so libB.so ... ("" = the in-tree build)
# two rounds, builds alternating; one line per run with value, ms/step and the exclusive stage times (cf. ab_lib.sh)
WL=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    for w in $WL; do
      ORBFE_LIB=$lib python bench.py --full-line --no-detail --workload $w --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache > gpurun_out/b_abl.json 2> gpurun_out/b_abl.err
      python - "$w" "${lib:-in-tree}" <<'PY'
import json, sys, os
j = json.loads(open("gpurun_out/b_abl.json").read().strip().splitlines()[-1])
st = j["roofline"]["stages"]
print("[%-18s %-7s] value %8.0f  ms/step %7.3f | excl " % (os.path.basename(sys.argv[2])[:18], sys.argv[1], j["value"], j["ms_per_step"]) +
      "  ".join("%s %.3f" % (k, v["ms_per_step_exclusive"]) for k, v in st.items()))
PY
    done
  done
done
