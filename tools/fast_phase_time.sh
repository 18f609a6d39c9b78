#!/bin/bash
# Exclusive duration of k_fast_cells with the kernel cut off after staging / A / B / C and in full ($ORBFE_FAST_CUTOFF;
# timing experiment: outputs are empty by construction, ORBFE_BENCH_NO_CHECK).   usage: fast_phase_time.sh [workload]
cd "$(dirname "$0")/.."
W=${1:-tum}
for C in 1 2 3 4 0; do
  ORBFE_FAST_CUTOFF=$C ORBFE_BENCH_NO_CHECK=1 python bench.py --full-line --no-detail --workload $W --no-e2e --no-cpu-baseline --no-latency --min-seconds 0.5 --input-cache /tmp/orbfe_cache_fpt > gpurun_out/b_fpt.json 2> gpurun_out/b_fpt.err
  python - $C $W <<'PY'
import json, sys
j = json.loads(open("gpurun_out/b_fpt.json").read().strip().splitlines()[-1])
names = {"1": "staging only", "2": "staging + A", "3": "staging + A + B", "4": "staging + A + B + C", "0": "full kernel (+ D)"}
print("%-8s %-22s fast excl %.3f ms per step (live %.3f)" % (sys.argv[2], names[sys.argv[1]], j["roofline"]["stages"]["fast"]["ms_per_step_exclusive"], j["roofline"]["stages"]["fast"]["ms_per_step_live"]))
PY
done
