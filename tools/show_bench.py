#!/usr/bin/env python3
"""Human-readable summary of a bench.py JSON line: python tools/show_bench.py gpurun_out/b.json"""
import json
import sys


def show(r, name):
    rf = r["roofline"]
    print("==", name, "value", round(r["value"]), r["unit"], "ms/step", round(r["ms_per_step"], 3))
    print("  dom", rf["stage"], "frac %.4f" % rf["frac"], "ach", round(rf["achieved"]), "traffic", rf["traffic"],
          "| pipeline frac %.3f" % rf["pipeline"]["frac"], "sum_excl %.2f" % rf["pipeline"]["sum_exclusive_ms"],
          "| schedule", rf.get("schedule"), rf.get("streams"))
    for s, v in rf["stages"].items():
        vi = rf.get("valu_issue", {}).get("stages", {}).get(s, {})
        print("    %-12s excl %7.3f live %7.3f hbm_excl %.3f" % (s, v["ms_per_step_exclusive"], v["ms_per_step_live"], v["hbm_frac_exclusive"]),
              ("valu_frac %.3f busy %s" % (vi["frac_exclusive"], vi.get("valu_busy_pmc"))) if vi else "")
    if "matching" in rf:
        m = rf["matching"]
        print("    matching: pairs/unit %.0f  pairs/s %.3g  popcnt frac %.4f" % (m["distance_pairs_per_unit"], m["distance_pairs_per_s"], m["frac"]))
    if "valu_issue" in rf:
        print("    valu pipeline frac", rf["valu_issue"].get("pipeline_frac"), "bound_closest", rf.get("bound_closest"))
    cb = r.get("cpu_baseline")
    if cb:
        print("    cpu", round(cb["value"], 1), cb["unit"], "cores", cb["cores"],
              {k: (round(v["value"], 1), v["cores"], round(v["mean_ms"], 2), round(v["median_ms"], 2)) for k, v in cb["variants"].items()},
              "vs_cpu", round(r.get("vs_cpu_baseline", 0)))
    if "e2e" in r:
        e = r["e2e"]
        print("    e2e %.0f images/s (%d images, %.1f ms) pcie" % (e["images_per_s"], e["images"], e["ms"]), {k: round(v, 1) for k, v in e["pcie_GBps"].items()})
    print("    parity", r["parity_check"]["units_checked"])


txt = open(sys.argv[1]).read().strip()
try:
    j = json.loads(txt)  # the detail file (bench.py --detail-out): one pretty-printed object
except ValueError:
    j = json.loads(txt.splitlines()[-1])  # a stdout capture of `bench.py --full-line`
if "stages" not in j.get("roofline", {}):
    raise SystemExit("this is the compact contract line; pass the detail file (gpurun_out/bench_detail.json) or a --full-line capture")
show(j, "HEAD " + j["config"]["workload"][:40])
for r in j.get("secondary", []):
    show(r, r["key"])
