#!/bin/bash
# rocprofv3 --kernel-trace --stats of the timed multi-stream bench run of one workload: tools/stats_quick.sh kitti
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-kitti}
python3 bench.py --full-line --no-detail --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --input-cache /tmp/ic_sq > /dev/null 2>&1
rm -rf gpurun_out/sq_$W
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sq_$W -- python3 bench.py --full-line --no-detail --workload $W --no-cpu-baseline --no-e2e --render-procs 1 --input-cache /tmp/ic_sq > gpurun_out/sq_$W.json 2>/dev/null
cp gpurun_out/sq_$W/*/*kernel_stats.csv gpurun_out/sq_${W}_kernel_stats.csv
python3 - gpurun_out/sq_${W}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    n = r["Name"].replace("void ", "").replace("orbfe::", "").replace("(anonymous namespace)::", "")[:30]
    print("%-30s calls %5s total %8.1f ms avg %8.1f us %5.1f%%" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
