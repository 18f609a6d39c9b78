#!/bin/bash
# Per-phase instruction split of k_fast_cells: SQ_INSTS_VALU / SALU / LDS per wave (= per grid cell) with the kernel cut off
# after staging / phase A / B / C ($ORBFE_FAST_CUTOFF; outputs are empty by construction, ORBFE_BENCH_NO_CHECK) and in full.
# One rocprofv3 --pmc pass each, under timeout, with a progress line.   usage: fast_phase_split.sh [workload] [batch] [TAG]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-tum}; B=${2:-256}; TAG=${3:-}
OUT=gpurun_out/fast_phase_split_$W${TAG:+_$TAG}
mkdir -p gpurun_out
: > $OUT.txt
timeout -k 10 300 python3 bench.py --full-line --no-detail --workload $W --steps 1 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --batch $B --input-cache /tmp/orbfe_cache_fps > /dev/null 2>&1
for C in 1 2 3 4 0; do
  rm -rf ${OUT}_c$C
  ORBFE_FAST_CUTOFF=$C ORBFE_BENCH_NO_CHECK=1 timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d ${OUT}_c$C -- \
    python3 bench.py --full-line --no-detail --workload $W --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --render-procs 1 --streams 1 --batch $B --input-cache /tmp/orbfe_cache_fps > /dev/null 2>${OUT}_c$C.err
  echo "cutoff $C rc=$? $(date +%T)" >> $OUT.progress
  python3 - $C ${OUT}_c$C >> $OUT.txt <<'PY'
import csv, glob, sys, collections
c, d = sys.argv[1], sys.argv[2]
names = {"1": "staging only", "2": "staging + A", "3": "staging + A + B", "4": "staging + A + B + C", "0": "full kernel (+ D)"}
fs = glob.glob(d + "/*/*counter_collection.csv")
if not fs:
    print("cutoff", c, "no counters"); sys.exit(0)
acc = collections.defaultdict(float)
for r in csv.DictReader(open(fs[0])):
    if "k_fast_cells" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
w = acc["SQ_WAVES"] or 1
print("%-22s VALU/wave %7.1f  SALU/wave %7.1f  LDS/wave %6.1f   (%d waves sampled)" % (names[c], acc["SQ_INSTS_VALU"] / w, acc["SQ_INSTS_SALU"] / w, acc["SQ_INSTS_LDS"] / w, w))
PY
  rm -rf ${OUT}_c$C
done
cat $OUT.txt
