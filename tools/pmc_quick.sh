#!/bin/bash
# VALU instructions per wave and per step for every kernel (rocprofv3 --pmc, 1 stream):  pmc_quick.sh [workload] [batch]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-tum}; B=${2:-256}
rm -rf gpurun_out/pmcq
timeout -k 10 300 python3 bench.py --full-line --no-detail --workload $W --steps 1 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --batch $B --input-cache /tmp/orbfe_cache_q > /dev/null 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmcq -- python3 bench.py --full-line --no-detail --workload $W --steps 3 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --render-procs 1 --streams 1 --batch $B --input-cache /tmp/orbfe_cache_q > /dev/null 2>&1
echo "pmc pass rc=$?"
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/pmcq/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']]+=float(r['Counter_Value'])
tot=0
for k,v in sorted(agg.items()):
    if 'orbfe' not in k: continue
    w=v['SQ_WAVES']; steps=5
    print(k.replace('void ','').replace('orbfe::','')[:48].ljust(50), 'VALU/wave %5.0f'%(v['SQ_INSTS_VALU']/w), 'SALU/wave %5.0f'%(v['SQ_INSTS_SALU']/w), 'VALU M/step %6.1f'%(v['SQ_INSTS_VALU']/steps/1e6))
    tot+=v['SQ_INSTS_VALU']/steps/1e6
print('total VALU M wave-instr per step: %.1f'%tot)
PY
