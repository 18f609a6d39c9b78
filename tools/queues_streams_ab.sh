for cfg in "4 8" "6 12" "5 10" "3 6" "3 9" "8 16" "6 6" "2 8"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$1 python3 bench.py --full-line --no-detail --workload kitti --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache --streams $2 > gpurun_out/b_d.json 2>gpurun_out/b_d.err
  python3 - "$1 queues, $2 streams" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/b_d.json").read().strip().splitlines()[-1])
print("[%-22s] value %8.0f ms/step %.3f" % (sys.argv[1], j["value"], j["ms_per_step"]))
PY
done
