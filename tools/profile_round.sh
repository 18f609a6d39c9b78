#!/bin/bash
# Regenerates the judged profile artefacts of a round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01
# -> gpurun_out/<tag>_{valu,traffic}.json (PMC summaries), <tag>_bench.json (bench.py line, which reads the
#    two summaries), <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same bench command).
# Copy the four files into profiles/.
set -e
TAG=${1:-r01}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_pmc_valu $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_stats
PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --streams 1 --batch 256"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d $OUT/${TAG}_pmc_valu -- python3 bench.py $PMC_ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 bench.py $PMC_ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 bench.py $PMC_ARGS > /dev/null 2>&1
python3 tools/collect_valu.py $OUT/${TAG}_pmc_valu $OUT/${TAG}_valu.json 256 tum > /dev/null
python3 tools/collect_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_traffic.json 256 tum > /dev/null
cp $OUT/${TAG}_valu.json $OUT/${TAG}_traffic.json profiles/   # bench.py reads the summaries from profiles/
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
cp $OUT/${TAG}_stats/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
echo done
