#!/usr/bin/env python3
"""Where do the hardware queues of the bench wait?  Reads a rocprofv3 kernel_trace.csv of `bench.py --workload W` and lists, per
hardware queue, how much of the steady-state span it is idle, the distribution of its idle gaps and, for the long ones, the
kernel in front of and behind the gap (a queue serves the streams bound to it in enqueue order: a long gap means the host
had not yet enqueued the next kernel of that queue, or that kernel waited for an event).
    python tools/queue_gaps.py path/to/kernel_trace.csv [min_gap_us]
CAVEAT, found the first time this was used: under rocprofv3 a launch costs the host ~35 us, the bench's ~160 launches per
5 ms step make the PROFILED run host-bound (bench.py reports `host_enqueue_ms_per_step_idle_queues` 5.6 ms there against
0.38 ms unprofiled), and the > 500 us gaps this tool then finds between the chains of a queue (20-27 % of every queue) are
the profiler's, not the pipeline's: the unprofiled host is an order of magnitude ahead of the GPU."""
import collections
import csv
import sys


def short(n):
    return n.split("(")[0].replace("void ", "").replace("orbfe::", "").replace("(anonymous namespace)::", "").split("<")[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    lo = int(rows[len(rows) // 5]["Start_Timestamp"])
    hi = int(rows[4 * len(rows) // 5]["Start_Timestamp"])
    perq = collections.defaultdict(list)
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e > lo and s < hi:
            perq[r["Queue_Id"]].append((s, e, short(r["Kernel_Name"]), r.get("Stream_Id", "?")))
    span = hi - lo
    print(f"# steady-state span {span / 1e6:.1f} ms, {len(rows)} dispatches")
    for q, v in sorted(perq.items()):
        v.sort()
        idle = 0
        gaps = []
        end = max(v[0][0], lo)
        prev = None
        for s, e, k, st in v:
            if s > end:
                g = (s - end) / 1e3
                idle += s - end
                gaps.append((g, prev, (k, st), (end - lo) / 1e3))
            if e > end:
                end, prev = e, (k, st)
        hist = collections.Counter()
        for g, *_ in gaps:
            hist["<10" if g < 10 else "10-40" if g < 40 else "40-150" if g < 150 else "150-500" if g < 500 else ">500"] += g
        print(f"queue {q}: idle {idle / span:.3f} of the span; idle time by gap length (us): " +
              "  ".join(f"{k}: {hist[k] / 1e3:.2f} ms" for k in ("<10", "10-40", "40-150", "150-500", ">500")))
        pairs = collections.Counter()
        for g, p, n, t in gaps:
            if g >= min_gap and p:
                pairs[(p[0], n[0], p[1] == n[1])] += g
        for (a, b, same), tot in pairs.most_common(8):
            print(f"    {tot / 1e3:7.2f} ms of gaps >= {min_gap:.0f} us between {a:22s} -> {b:22s} ({'same stream' if same else 'other stream'})")


if __name__ == "__main__":
    main()
