#!/usr/bin/env python3
"""Turns a rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES pass of bench.py into profiles/<round>_valu.json:
VALU wave-instructions per frame for every kernel (average per dispatch x dispatches per step / batch).

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_valu -- \
        python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --batch 256
    python3 tools/collect_valu.py gpurun_out/pmc_valu gpurun_out/r01_valu.json 256 tum
"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path

LAUNCHES_PER_STEP = {"k_resize_flat": 7, "k_resize": 7}


def kname(full):
    """orbfe::k_x(args) / void orbfe::k_y<64>(args) -> k_x / k_y"""
    n = full.split("(")[0].replace("void ", "").replace("orbfe::", "")
    return n.split("<")[0]


def main():
    d, out, batch, workload = sys.argv[1:5]
    files = glob.glob(f"{d}/*/*counter_collection.csv")
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(max(files, key=lambda f: __import__('os').path.getmtime(f)))):
        if "orbfe::" in r["Kernel_Name"]:
            acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"batch": int(batch), "workload": workload,
           "unit": "VALU wave64 instructions (SQ_INSTS_VALU) per frame; issue peak = 256 CU x 4 SIMD x 2.4 GHz / 4 cycles",
           "kernels": {}}
    for k, c in sorted(acc.items()):
        n = LAUNCHES_PER_STEP.get(k, 1)
        valu = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"]) * n
        waves = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"]) * n
        res["kernels"][k] = {"dispatches_sampled": len(c["SQ_INSTS_VALU"]), "dispatches_per_step": n,
                             "valu_per_wave": valu / waves, "valu_wave_instr_per_frame": valu / int(batch)}
    res["total_valu_wave_instr_per_frame"] = sum(v["valu_wave_instr_per_frame"] for v in res["kernels"].values())
    Path(out).write_text(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
