run() { local label=$1; shift
  env "$@" python3 bench.py --full-line --no-detail --workload kitti --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache $EXTRA > gpurun_out/b_d.json 2>gpurun_out/b_d.err
  python3 - "$label" <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/b_d.json").read().strip().splitlines()[-1])
    print("[%-52s] value %8.0f ms/step %.3f" % (sys.argv[1], j["value"], j["ms_per_step"]))
except Exception as e:
    print("[%-52s] FAILED %s" % (sys.argv[1], e)); print(open("gpurun_out/b_d.err").read()[-600:])
PY
}
EXTRA="" run "normal priority, no process group" A=1
EXTRA="" run "high priority, no process group" ORBFE_STREAM_PRIORITY=high
EXTRA="" run "low priority, no process group" ORBFE_STREAM_PRIORITY=low
EXTRA="--force-dist --eager-dist" run "normal priority, nccl at start-up" A=1
EXTRA="--force-dist --eager-dist" run "high priority, nccl at start-up" ORBFE_STREAM_PRIORITY=high
EXTRA="--force-dist --eager-dist" run "low priority, nccl at start-up" ORBFE_STREAM_PRIORITY=low
