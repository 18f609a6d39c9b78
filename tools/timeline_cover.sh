#!/bin/bash
# How full is the device timeline?  rocprofv3 --kernel-trace of one bench workload, then: share of the traced span with at
# least one kernel running, mean number of kernels running, and the share of the span each kernel is resident.
#   tools/timeline_cover.sh kitti    -> gpurun_out/timeline_<workload>.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=${1:-kitti}
OUT=gpurun_out/timeline_$W
rm -rf ${OUT}_trace
timeout -k 10 300 python3 bench.py --full-line --no-detail --workload $W --steps 1 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --input-cache /tmp/orbfe_cache_tl > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d ${OUT}_trace -- python3 bench.py --full-line --no-detail --workload $W --steps 40 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-e2e --no-latency --render-procs 1 --input-cache /tmp/orbfe_cache_tl > ${OUT}_bench.json 2>/dev/null
python3 - ${OUT}_trace $W > $OUT.txt <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
rows = [r for r in csv.DictReader(open(fs[0])) if "orbfe::" in r["Kernel_Name"]]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("orbfe::", "").split("<")[0]) for r in rows]
ev.sort()
# steady state: the middle 60 % of the dispatches
lo, hi = ev[len(ev) // 5][0], ev[4 * len(ev) // 5][0]
pts = []
res = collections.defaultdict(int)
for s, e, k in ev:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        pts += [(s, 1), (e, -1)]
        res[k] += e - s
pts.sort()
cov = conc = 0
cur = 0
last = lo
for t, d in pts:
    if cur > 0:
        cov += t - last
    conc += cur * (t - last)
    cur += d
    last = t
span = hi - lo
print("# %s: %d dispatches, steady-state span %.1f ms" % (sys.argv[2], len(ev), span / 1e6))
print("at least one kernel running: %.3f of the span; mean kernels running: %.2f" % (cov / span, conc / span))
def union(iv):
    iv.sort()
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)
per = collections.defaultdict(list)
for s, e, k in ev:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        per[k].append((s, e))
for k, v in sorted(res.items(), key=lambda kv: -kv[1]):
    print("  %-26s sum of durations %.3f of the span, at least one running %.3f, mean duration %.1f us" % (k, v / span, union(per[k]) / span, v / len(per[k]) / 1e3))
qcol0 = [c for c in rows[0].keys() if "Queue" in c]
if qcol0:
    perq = collections.defaultdict(list)
    for r in rows:
        s_, e_ = max(int(r["Start_Timestamp"]), lo), min(int(r["End_Timestamp"]), hi)
        if e_ > s_:
            perq[r[qcol0[0]]].append((s_, e_))
    print("hardware queues busy:", "  ".join("q%s: %.3f" % (q, union(v) / span) for q, v in sorted(perq.items())))
# an excerpt of the timeline: 70 consecutive dispatches from the middle, start / end in us relative to the first, queue
qcol = [c for c in rows[0].keys() if "Queue" in c]
mid = len(rows) // 2
ex = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))[mid:mid + 70]
t0 = int(ex[0]["Start_Timestamp"])
print("excerpt (start us, end us, queue, kernel, grid):")
for r in ex:
    print("  %9.1f %9.1f  q%-3s %-24s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
          r[qcol[0]] if qcol else "?", r["Kernel_Name"].split("(")[0].replace("void ", "").replace("orbfe::", "").split("<")[0],
          r.get("Grid_Size_X", r.get("Grid_Size", ""))))
hist = collections.Counter()
cur, last = 0, lo
for t, d in pts:
    hist[cur] += t - last
    cur += d
    last = t
print("kernels running at once:", "  ".join("%d: %.3f" % (n, hist[n] / span) for n in sorted(hist)))
PY
rm -rf ${OUT}_trace
cat $OUT.txt
