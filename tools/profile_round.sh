#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box (run through gpurun):
#   tools/profile_round.sh r03 [workloads...]          (default: kitti tum euroc euroc_stereo)
# per workload W -> gpurun_out/<tag>_W_{valu,traffic}.json (PMC summaries, separate passes) and
#   <tag>_W_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --workload W`, the timed multi-stream run) and
#   <tag>_W_kernel_stats_exclusive.csv (the same on ONE stream: every kernel alone on the chip);
# then <tag>_bench.json = the default bench.py line (all workloads; reads the PMC summaries from profiles/).
# Copy the files into profiles/.
# Every profiler pass runs under `timeout -k 10` and appends a line to gpurun_out/<tag>_progress.txt when it ends: a pass
# that dies (round 2: rocprofv3 aborting inside a counter configuration and never returning) costs its own limit, not
# the call, and the call is never silent for minutes.
TAG=${1:-r03}
shift || true
WLS=${@:-kitti tum euroc euroc_stereo}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
PROG=$OUT/${TAG}_progress.txt
: > $PROG
echo "commit $(cat .profile_commit 2>/dev/null || echo unknown) liborbfe.so sha256 $(sha256sum orb_slam2_annotate_amd/liborbfe.so | cut -c1-16)" >> $PROG
note() { echo "$(date +%T) $*" | tee -a $PROG; }
CACHE=/tmp/orbfe_inputs_$$
pbatch() { case $1 in kitti|euroc_stereo) echo 64;; *) echo 256;; esac; }
pimgs() { case $1 in kitti|euroc_stereo) echo 128;; *) echo 256;; esac; }
for W in $WLS; do  # render every batch once, unprofiled (a profiled process must not fork the renderer pool)
  timeout -k 10 300 python3 bench.py --full-line --no-detail --workload $W --steps 1 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --batch $(pbatch $W) --input-cache $CACHE > /dev/null 2>&1
  timeout -k 10 300 python3 bench.py --full-line --no-detail --workload $W --steps 1 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --input-cache $CACHE > /dev/null 2>&1
  note "inputs $W rendered rc=$?"
done
for W in $WLS; do
  B=$(pbatch $W); IMGS=$(pimgs $W)
  rm -rf $OUT/${TAG}_${W}_pmc_valu $OUT/${TAG}_${W}_pmc_fetch $OUT/${TAG}_${W}_pmc_write $OUT/${TAG}_${W}_stats
  PMC_ARGS="--workload $W --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --render-procs 1 --input-cache $CACHE --streams 1 --batch $B"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_${W}_pmc_valu -- python3 bench.py --full-line --no-detail $PMC_ARGS > /dev/null 2>$OUT/${TAG}_${W}_pmc_valu.err
  note "pmc valu $W rc=$?"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${W}_pmc_fetch -- python3 bench.py --full-line --no-detail $PMC_ARGS > /dev/null 2>&1
  note "pmc fetch $W rc=$?"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${W}_pmc_write -- python3 bench.py --full-line --no-detail $PMC_ARGS > /dev/null 2>&1
  note "pmc write $W rc=$?"
  python3 tools/collect_valu.py $OUT/${TAG}_${W}_pmc_valu $OUT/${TAG}_${W}_valu.json $IMGS $W > /dev/null && \
  python3 tools/collect_traffic.py $OUT/${TAG}_${W}_pmc_fetch $OUT/${TAG}_${W}_pmc_write $OUT/${TAG}_${W}_traffic.json $IMGS $W > /dev/null && \
  cp $OUT/${TAG}_${W}_valu.json $OUT/${TAG}_${W}_traffic.json profiles/   # bench.py reads the summaries from profiles/
  note "pmc summaries $W rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${W}_stats -- python3 bench.py --full-line --no-detail --workload $W --no-cpu-baseline --no-e2e --no-latency --render-procs 1 --input-cache $CACHE > $OUT/${TAG}_${W}_stats_bench.json 2>/dev/null
  note "kernel stats $W rc=$?"
  cp $OUT/${TAG}_${W}_stats/*/*kernel_stats.csv $OUT/${TAG}_${W}_kernel_stats.csv
  # the same command on ONE stream: every kernel alone on the chip -> the exclusive durations behind roofline.exclusive and
  # the per-stage exclusive fractions, reproducible from profiles/ without trusting bench-printed HIP events
  rm -rf $OUT/${TAG}_${W}_stats1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${W}_stats1 -- python3 bench.py --full-line --no-detail --workload $W --streams 1 --steps 20 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --render-procs 1 --input-cache $CACHE > /dev/null 2>&1
  note "kernel stats, 1 stream $W rc=$?"
  cp $OUT/${TAG}_${W}_stats1/*/*kernel_stats.csv $OUT/${TAG}_${W}_kernel_stats_exclusive.csv
  rm -rf $OUT/${TAG}_${W}_stats1
  rm -rf $OUT/${TAG}_${W}_pmc_valu $OUT/${TAG}_${W}_pmc_fetch $OUT/${TAG}_${W}_pmc_write $OUT/${TAG}_${W}_stats   # (raw traces: tens of MB)
done
# the default run: stdout = the compact contract line (what the driver parses), --detail-out = the full result dict
timeout -k 10 600 python3 bench.py --input-cache $CACHE --detail-out $OUT/${TAG}_bench.json > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
note "bench rc=$?"
rm -rf $CACHE
tail -c 4200 $OUT/${TAG}_bench_line.json; echo
echo done
