for cfg in "--schedule streams" "--schedule lanes" "--schedule lanes --streams 4" "--schedule lanes --streams 16"; do
  python bench.py --full-line --no-detail --workload kitti --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache $cfg > gpurun_out/b_l.json 2>/dev/null
  python - "$cfg" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/b_l.json").read().strip().splitlines()[-1])
print("[%-32s] value %8.0f ms/step %.3f" % (sys.argv[1], j["value"], j["ms_per_step"]))
PY
done
for q in 2 3 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --full-line --no-detail --workload kitti --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache --schedule lanes > gpurun_out/b_l.json 2>/dev/null
  python - "lanes GPU_MAX_HW_QUEUES=$q" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/b_l.json").read().strip().splitlines()[-1])
print("[%-32s] value %8.0f ms/step %.3f" % (sys.argv[1], j["value"], j["ms_per_step"]))
PY
done
