#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate --pmc passes, the guide's gfx950 correction) of every kernel of ONE workload with the
# current environment (e.g. ORBFE_DESC_TILES=1): tools/traffic_one.sh kitti 64 tiles -> gpurun_out/traffic_<w>_<tag>.json
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-kitti}; B=${2:-64}; TAG=${3:-x}
case $W in kitti|euroc_stereo) IMGS=$((2*B));; *) IMGS=$B;; esac
OUT=gpurun_out/traffic_${W}_$TAG
rm -rf ${OUT}_f ${OUT}_w
ARGS="--workload $W --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --render-procs 1 --streams 1 --batch $B --input-cache /tmp/orbfe_cache_t"
timeout -k 10 300 python3 bench.py --full-line --no-detail --workload $W --steps 1 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --batch $B --input-cache /tmp/orbfe_cache_t > /dev/null 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${OUT}_f -- python3 bench.py --full-line --no-detail $ARGS > /dev/null 2>&1; echo "fetch pass rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${OUT}_w -- python3 bench.py --full-line --no-detail $ARGS > /dev/null 2>&1; echo "write pass rc=$?"
python3 tools/collect_traffic.py ${OUT}_f ${OUT}_w $OUT.json $IMGS $W > /dev/null && python3 - $OUT.json <<'PY'
import json, sys
t = json.load(open(sys.argv[1]))
for k, e in t["kernels"].items():
    print("%-28s %9.2f MB per launch = %7.3f MB per image" % (k, e["traffic_bytes_per_launch"] / 1e6, e["traffic_bytes_per_launch"] / 1e6 / t["batch"]))
PY
rm -rf ${OUT}_f ${OUT}_w
