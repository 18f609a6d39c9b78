#!/usr/bin/env python3
"""Turns a rocprofv3 --pmc pass of bench.py (SQ_WAVES SQ_INSTS_VALU [SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE]) into
profiles/<round>_<workload>_valu.json: VALU wave-instructions per IMAGE for every kernel (average per dispatch x
dispatches per step / images per step) and, when the two extra counters are present, the hardware's own VALU
utilisation VALUBusy = 4 * SQ_ACTIVE_INST_VALU / (SIMDs * GRBM_GUI_ACTIVE / 8 XCDs) (SQ_ACTIVE_INST_* count quad-cycles).

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv \
        -d gpurun_out/pmc_valu -- python3 bench.py --workload tum --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --batch 256
    python3 tools/collect_valu.py gpurun_out/pmc_valu gpurun_out/r02_tum_valu.json 256 tum
(third argument: IMAGES per step -- 2 x --batch for the stereo workload)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict
from pathlib import Path

# launches per step of a kernel = its sampled dispatches / those of k_fast_cells (one launch per step): 7 for k_resize_flat,
# 8 for k_blur7 when the pyramid is fused into the per-level blur (7 fused launches + the top level), 1 otherwise
N_SIMD = 256 * 4
N_XCD = 8  # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (checked against HIP-event times: k_blur7 at 256
           # frames counts 3.90 M = 8 x 488 k cycles = 8 x 0.203 ms at 2.4 GHz, its event time); the SQ counters are chip sums


def kname(full):
    """orbfe::k_x(args) / void orbfe::k_y<64>(args) -> k_x / k_y"""
    n = full.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("orbfe::", "")
    return n.split("<")[0]


def main():
    d, out, images, workload = sys.argv[1:5]
    files = glob.glob(f"{d}/*/*counter_collection.csv")
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if "orbfe::" in r["Kernel_Name"] or "k_remap" in r["Kernel_Name"] or "k_cvt" in r["Kernel_Name"]:
            acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"batch": int(images), "workload": workload,
           "unit": "VALU wave64 instructions (SQ_INSTS_VALU) per image; valu_busy = 4*SQ_ACTIVE_INST_VALU/(1024 SIMDs*GRBM_GUI_ACTIVE/8)",
           "kernels": {}}
    base = len(acc["k_fast_cells"]["SQ_INSTS_VALU"]) if "k_fast_cells" in acc else 0
    for k, c in sorted(acc.items()):
        if not c["SQ_INSTS_VALU"]:
            continue
        n = max(1, round(len(c["SQ_INSTS_VALU"]) / base)) if base else 1
        valu = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"]) * n
        waves = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"]) * n
        e = {"dispatches_sampled": len(c["SQ_INSTS_VALU"]), "dispatches_per_step": n,
             "valu_per_wave": valu / max(waves, 1), "valu_wave_instr_per_image": valu / int(images)}
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
            act = sum(c["SQ_ACTIVE_INST_VALU"]) / len(c["SQ_ACTIVE_INST_VALU"])
            gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
            e["SQ_ACTIVE_INST_VALU"] = act
            e["GRBM_GUI_ACTIVE"] = gui
            e["valu_busy"] = 4.0 * act / (N_SIMD * gui / N_XCD) if gui > 0 else None
        res["kernels"][k] = e
    res["total_valu_wave_instr_per_image"] = sum(v["valu_wave_instr_per_image"] for v in res["kernels"].values())
    Path(out).write_text(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
