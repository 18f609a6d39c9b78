#!/bin/bash
# HBM fetch bytes (FETCH_SIZE, KB units, doubled per the gfx950 note) and duration per kernel, 1 stream, B=256
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/pmcf
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcf -- python3 bench.py --full-line --no-detail --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --batch 256 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/pmcf/*/*counter_collection.csv')[0]
t=glob.glob('gpurun_out/pmcf/*/*kernel_trace.csv')[0]
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    agg[k]+=float(r['Counter_Value']); n[k]+=1
dur=collections.defaultdict(float)
for r in csv.DictReader(open(t)):
    dur[r['Kernel_Name'].split('(')[0]]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k in sorted(agg):
    if 'orbfe' not in k: continue
    steps=5
    mb=agg[k]*1024*2/steps/1e6
    print(k.ljust(30),'fetch MB/step %8.1f'%mb,'us/step %7.1f'%(dur[k]/steps),'fetch TB/s %.2f'%(mb/(dur[k]/steps)))
PY
