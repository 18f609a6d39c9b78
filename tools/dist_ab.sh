#!/bin/bash
# Does an initialised torch.distributed / RCCL communicator in the process change the single-GPU rate?  (round 4: yes, -14 %)
run() { local label=$1; shift
  env "$@" python3 bench.py --full-line --no-detail --workload kitti --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache $EXTRA > gpurun_out/b_d.json 2>gpurun_out/b_d.err
  python3 - "$label" <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/b_d.json").read().strip().splitlines()[-1])
    print("[%-44s] value %8.0f ms/step %.3f" % (sys.argv[1], j["value"], j["ms_per_step"]))
except Exception as e:
    print("[%-44s] FAILED %s" % (sys.argv[1], e)); print(open("gpurun_out/b_d.err").read()[-600:])
PY
}
EXTRA="" run "no process group" A=1
EXTRA="--force-dist --eager-dist" run "nccl (RCCL) created at start-up, 1 rank" A=1
EXTRA="--force-dist" run "nccl created at the first barrier (LazyDist)" A=1
EXTRA="--force-dist --dist-backend gloo" run "gloo, 1 rank" A=1
EXTRA="--force-dist --eager-dist" run "nccl at start-up, GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
EXTRA="" run "no process group (again)" A=1
# all four workloads in one process: the later workloads' handles are created AFTER the communicator -- they must get the
# first workload's streams back from the library's stream pool
runall() { local label=$1; shift
  env "$@" python3 bench.py --full-line --no-detail --workload all --no-e2e --no-cpu-baseline --no-latency --input-cache /tmp/orbfe_ab_cache $EXTRA > gpurun_out/b_d.json 2>gpurun_out/b_d.err
  python3 - "$label" <<'PY'
import json, sys
try:
    j = json.loads(open("gpurun_out/b_d.json").read().strip().splitlines()[-1])
    print("[%-44s] kitti %7.0f  " % (sys.argv[1], j["value"]) + "  ".join("%s %7.0f" % (s_["key"], s_["value"]) for s_ in j.get("secondary", [])))
except Exception as e:
    print("[%-44s] FAILED %s" % (sys.argv[1], e)); print(open("gpurun_out/b_d.err").read()[-600:])
PY
}
EXTRA="" runall "all, no process group" A=1
EXTRA="--force-dist" runall "all, nccl at the first barrier" A=1
EXTRA="--force-dist --eager-dist" runall "all, nccl at start-up" A=1
