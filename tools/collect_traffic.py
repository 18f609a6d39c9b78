#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of bench.py into profiles/<round>_traffic.json.

Run on the GPU box (see profiles/README.md for the exact gpurun command):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE ... -d gpurun_out/pmc_write -- python3 bench.py ...
FETCH_SIZE and WRITE_SIZE need separate passes (TCC counter slots, MI355X_MICROARCH.md).  Units are
KB.  gfx950 correction from the same guide: FETCH_SIZE counts 128-B read requests as 64 B, so it is
DOUBLED before it is compared with a byte count; WRITE_SIZE is used as is.
"""
import csv
import glob
import json
import sys
from collections import defaultdict
from pathlib import Path


def kname(full):
    """orbfe::k_x(args) / void orbfe::k_y<64>(args) -> k_x / k_y"""
    n = full.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("orbfe::", "")
    return n.split("<")[0]


def per_kernel(dirname, counter):
    files = glob.glob(f"{dirname}/*/*counter_collection.csv")
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirname}")
    acc = defaultdict(list)
    for r in csv.DictReader(open(max(files, key=lambda f: __import__('os').path.getmtime(f)))):
        if r["Counter_Name"] == counter and ("orbfe::" in r["Kernel_Name"] or "k_remap" in r["Kernel_Name"] or "k_cvt" in r["Kernel_Name"]):
            acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, out, batch, workload = sys.argv[1:6]
    fetch, n = per_kernel(fetch_dir, "FETCH_SIZE")
    write, _ = per_kernel(write_dir, "WRITE_SIZE")
    res = {"batch": int(batch), "workload": workload, "unit": "bytes per launch",
           "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
           "kernels": {}}
    for k in sorted(fetch):
        res["kernels"][k] = {"launches_sampled": n[k], "FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write.get(k, 0.0),
                             "traffic_bytes_per_launch": (2 * fetch[k] + write.get(k, 0.0)) * 1024}
    Path(out).write_text(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
