#!/bin/bash
# Per-kernel memory-system counters (exclusive, 1 stream): where a latency-bound kernel waits.
# One hardware block per pass and at most three of its counters: larger sets abort with "exceeds the capabilities of the
# hardware to collect" (the TA_* stall counters abort with it even alone on this image, so they are not collected).  Every pass is bounded by timeout and reports to gpurun_out/ so a failed pass cannot stall the run.
#   tools/pmc_mem.sh tum 256   -> gpurun_out/pmc_mem_<workload>.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-tum}; B=${2:-256}
OUT=gpurun_out/pmc_mem_$W
rm -rf ${OUT}_p*
python3 bench.py --full-line --no-detail --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --batch $B --input-cache /tmp/ic_mem > /dev/null 2>&1
ARGS="--workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --render-procs 1 --input-cache /tmp/ic_mem --streams 1 --batch $B"
PASSES=(
  "TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES"
  "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY"
  "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST"
  "TCC_HIT TCC_MISS"
  "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM"
  "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM"
)
i=0
for P in "${PASSES[@]}"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $P GRBM_GUI_ACTIVE --output-format csv -d ${OUT}_p$i -- python3 bench.py --full-line --no-detail $ARGS > /dev/null 2>${OUT}_p$i.err
  echo "pass $i ($P) rc=$?" | tee -a ${OUT}_progress.log
  i=$((i+1))
done
python3 - "$OUT" ${#PASSES[@]} <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in range(n):
    fs = glob.glob(f"{out}_p{p}/*/*counter_collection.csv")
    if not fs:
        print("no counters in pass", p, open(f"{out}_p{p}.err").read()[:900]); continue
    for r in csv.DictReader(open(fs[0])):
        if "orbfe::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("orbfe::", "").split("<")[0]
        c = r["Counter_Name"]
        agg[k][f"GUI{p}" if c == "GRBM_GUI_ACTIVE" else c].append(float(r["Counter_Value"]))
def m(k, c):
    return (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else float("nan")
def per_cu(k, c, p):   # fraction of the kernel's cycles, per CU (GUI is summed over the 8 XCDs)
    return m(k, c) / (256 * m(k, f"GUI{p}") / 8)
with open(out + ".txt", "w") as f:
    for k in sorted(agg):
        line = (f"{k:22s} cyc {m(k,'GUI0')/8:9.0f} | tcp_pending_stall {per_cu(k,'TCP_PENDING_STALL_CYCLES',0):.3f} "
                f"L1 acc/cyc/CU {per_cu(k,'TCP_TOTAL_CACHE_ACCESSES',0):.3f} L1 acc {m(k,'TCP_TOTAL_CACHE_ACCESSES'):.3g} | L1->L2 rd req {m(k,'TCP_TCC_READ_REQ'):.3g} lat {m(k,'TCP_TCC_READ_REQ_LATENCY')/max(m(k,'TCP_TCC_READ_REQ'),1):.0f} "
                f"| tlb miss/req {m(k,'TCP_UTCL1_TRANSLATION_MISS')/max(m(k,'TCP_UTCL1_REQUEST'),1):.4f} | L2 hit {m(k,'TCC_HIT')/max(m(k,'TCC_HIT')+m(k,'TCC_MISS'),1):.3f} "
                f"| wait_any/wavecyc {m(k,'SQ_WAIT_ANY')/m(k,'SQ_WAVE_CYCLES'):.3f} vmem_busy {4*m(k,'SQ_ACTIVE_INST_VMEM')/(1024*m(k,'GUI4')/8):.3f} "
                f"vmem_level {m(k,'SQ_INST_LEVEL_VMEM')/max(m(k,'SQ_INSTS_VMEM'),1):.0f}")
        print(line); f.write(line + "\n")
PY
