#!/bin/bash
# streams x batch sweep of one workload with the current kernels (same box): tools/stream_sweep3.sh kitti "4 8 12 16" "512 1024"
cd "$(dirname "$0")/.."
W=${1:-kitti}; SS=${2:-"4 8 12 16"}; BB=${3:-"0"}
for B in $BB; do
  for S in $SS; do
    python bench.py --full-line --no-detail --workload $W --no-e2e --no-cpu-baseline --no-latency --streams $S --batch $B --input-cache /tmp/orbfe_cache_ss > gpurun_out/b_ss.json 2> gpurun_out/b_ss.err
    echo "[$W streams=$S batch=$B] $(python tools/show_bench.py gpurun_out/b_ss.json | grep -E 'value' | tr '\n' ' ' | sed -E 's/ +/ /g' | cut -c1-160)"
  done
done
