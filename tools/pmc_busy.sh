#!/bin/bash
# Per-kernel utilisation of the shared execution resources (exclusive, 1 stream): VALU issue, LDS array, memory wait.
# GRBM_GUI_ACTIVE comes summed over the 8 XCDs (see tools/collect_valu.py), hence the /8 in the busy fractions.
#   tools/pmc_busy.sh tum 256      -> gpurun_out/pmc_busy_<workload>[_$TAG].txt   (every pass under `timeout -k 10 240`,
#   a progress line after each: a pass that dies must not leave the call silent)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-tum}; B=${2:-256}
OUT=gpurun_out/pmc_busy_$W${TAG:+_$TAG}
mkdir -p gpurun_out
rm -rf ${OUT}_a ${OUT}_b ${OUT}_c
ARGS="--workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --render-procs 1 --streams 1 --batch $B"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d ${OUT}_a -- python3 bench.py --full-line --no-detail $ARGS > /dev/null 2>${OUT}_a.err; echo "pass a done rc=$? $(date +%T)" >> ${OUT}.progress
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d ${OUT}_b -- python3 bench.py --full-line --no-detail $ARGS > /dev/null 2>${OUT}_b.err; echo "pass b done rc=$? $(date +%T)" >> ${OUT}.progress
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d ${OUT}_c -- python3 bench.py --full-line --no-detail $ARGS > /dev/null 2>${OUT}_c.err; echo "pass c done rc=$? $(date +%T)" >> ${OUT}.progress
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for suf in "abc":
    fs = glob.glob(f"{out}_{suf}/*/*counter_collection.csv")
    if not fs:
        print("no counters in pass", suf, open(f"{out}_{suf}.err").read()[-400:]); continue
    for r in csv.DictReader(open(fs[0])):
        if "orbfe::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("orbfe::", "").split("<")[0]
        agg[k][suf + ":" + r["Counter_Name"]].append(float(r["Counter_Value"]))
m = lambda k, c: (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else float("nan")
with open(out + ".txt", "w") as f:
    for k in sorted(agg):
        gui_a, gui_b, gui_c = m(k, "a:GRBM_GUI_ACTIVE"), m(k, "b:GRBM_GUI_ACTIVE"), m(k, "c:GRBM_GUI_ACTIVE")
        line = (f"{k:22s} disp {len(agg[k]['a:SQ_WAVES']):3d} cyc {gui_a:11.0f} waves {m(k,'a:SQ_WAVES'):9.0f} "
                f"VALU/wave {m(k,'a:SQ_INSTS_VALU')/m(k,'a:SQ_WAVES'):7.1f} valu_busy {4*m(k,'a:SQ_ACTIVE_INST_VALU')/(1024*gui_a/8):.3f} "
                f"| LDS/wave {m(k,'b:SQ_INSTS_LDS')/m(k,'a:SQ_WAVES'):6.1f} lds_busy {m(k,'b:SQ_LDS_IDX_ACTIVE')/(256*gui_b/8):.3f} "
                f"lds_conflict {m(k,'b:SQ_LDS_BANK_CONFLICT')/(256*gui_b/8):.3f} inst_lds_busy {4*m(k,'b:SQ_ACTIVE_INST_LDS')/(1024*gui_b/8):.3f} "
                f"| SALU/wave {m(k,'c:SQ_INSTS_SALU')/m(k,'a:SQ_WAVES'):6.1f} VMEM/wave {(m(k,'c:SQ_INSTS_VMEM_RD')+m(k,'c:SQ_INSTS_VMEM_WR'))/m(k,'a:SQ_WAVES'):5.1f} "
                f"wait_inst/wavecyc {m(k,'c:SQ_WAIT_INST_ANY')/m(k,'c:SQ_WAVE_CYCLES'):.3f}")
        print(line); f.write(line + "\n")
PY
