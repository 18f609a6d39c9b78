#!/bin/bash
# same-box A/B of environment knobs: tools/ab_env.sh "kitti tum" "A=1 B=2" "A=0" ...   (first argument = workloads, each
# further argument = one configuration; two rounds, configurations alternating)
WL=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    for w in $WL; do
      env $cfg python bench.py --full-line --no-detail --workload $w --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache > gpurun_out/b_ab.json 2> gpurun_out/b_ab.err
      python - "$w" "$cfg" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/b_ab.json").read().strip().splitlines()[-1])
st = j["roofline"]["stages"]
print("[%-22s %-7s] value %8.0f  ms/step %7.3f (host enqueue %.3f, on idle queues %.3f) | excl " % (sys.argv[2][:22], sys.argv[1], j["value"], j["ms_per_step"], j.get("repeats", {}).get("host_enqueue_ms_per_step", float("nan")), j.get("repeats", {}).get("host_enqueue_ms_per_step_idle_queues", float("nan"))) +
      "  ".join("%s %.3f" % (k, v["ms_per_step_exclusive"]) for k, v in st.items()))
PY
    done
  done
done
