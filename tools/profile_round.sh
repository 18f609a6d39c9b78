#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box (run through gpurun):
#   tools/profile_round.sh r02 [workloads...]          (default: kitti tum euroc)
# per workload W -> gpurun_out/<tag>_W_{valu,traffic}.json (PMC summaries, separate passes) and
#   <tag>_W_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --workload W`, the timed multi-stream run);
# then <tag>_bench.json = the default bench.py line (all three workloads; reads the PMC summaries from profiles/).
# Copy the files into profiles/.
set -e
TAG=${1:-r02}
shift || true
WLS=${@:-kitti tum euroc}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
CACHE=/tmp/orbfe_inputs_$$
for W in $WLS; do  # render every batch once, unprofiled (a profiled process must not fork the renderer pool)
  case $W in kitti) B=64;; *) B=256;; esac
  python3 bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --batch $B --input-cache $CACHE > /dev/null 2>&1
  python3 bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --input-cache $CACHE > /dev/null 2>&1
  echo "inputs $W rendered"
done
for W in $WLS; do
  case $W in kitti) B=64; IMGS=128;; tum) B=256; IMGS=256;; euroc) B=256; IMGS=256;; esac
  rm -rf $OUT/${TAG}_${W}_pmc_valu $OUT/${TAG}_${W}_pmc_fetch $OUT/${TAG}_${W}_pmc_write $OUT/${TAG}_${W}_stats
  PMC_ARGS="--workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --render-procs 1 --input-cache $CACHE --streams 1 --batch $B"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_${W}_pmc_valu -- python3 bench.py $PMC_ARGS > /dev/null 2>$OUT/${TAG}_${W}_pmc_valu.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${W}_pmc_fetch -- python3 bench.py $PMC_ARGS > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${W}_pmc_write -- python3 bench.py $PMC_ARGS > /dev/null 2>&1
  python3 tools/collect_valu.py $OUT/${TAG}_${W}_pmc_valu $OUT/${TAG}_${W}_valu.json $IMGS $W > /dev/null
  python3 tools/collect_traffic.py $OUT/${TAG}_${W}_pmc_fetch $OUT/${TAG}_${W}_pmc_write $OUT/${TAG}_${W}_traffic.json $IMGS $W > /dev/null
  cp $OUT/${TAG}_${W}_valu.json $OUT/${TAG}_${W}_traffic.json profiles/   # bench.py reads the summaries from profiles/
  echo "pmc $W done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${W}_stats -- python3 bench.py --workload $W --no-cpu-baseline --no-e2e --render-procs 1 --input-cache $CACHE > $OUT/${TAG}_${W}_stats_bench.json 2>/dev/null
  cp $OUT/${TAG}_${W}_stats/*/*kernel_stats.csv $OUT/${TAG}_${W}_kernel_stats.csv
  echo "stats $W done"
done
python3 bench.py --input-cache $CACHE > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rm -rf $CACHE
tail -c 400 $OUT/${TAG}_bench.json; echo
echo done
