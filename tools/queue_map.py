#!/usr/bin/env python3
"""Dev tool: which hardware queue ran what, phase by phase, from a rocprofv3 kernel trace of a whole bench run (gpurun_out/<tag>/trace):
for every stretch of the trace dominated by one workload's batched association kernel, the busy share of each hardware queue and the
streams that fed it.  Two lock-step groups on one queue run one after the other -- this is how that shows.   python tools/queue_map.py <tag>"""
import collections, csv, glob, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03q"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "trace", "*", "*kernel_trace.csv")), key=os.path.getmtime, reverse=True)
rd = csv.DictReader(open(f[0]))
print("columns:", rd.fieldnames)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", ""), r.get("Stream_Id", "")) for r in rd]
rows.sort()
# phases: split where no batched LM / association kernel ran for > 20 ms
batch = [r for r in rows if "_batch" in r[2]]
phases, cur = [], [batch[0]]
for r in batch[1:]:
    if r[0] - cur[-1][1] > 20_000_000:
        phases.append(cur); cur = []
    cur.append(r)
phases.append(cur)
for ph in phases:
    t0, t1 = ph[0][0], ph[-1][1]
    names = collections.Counter(r[2] for r in ph).most_common(2)
    busy = collections.defaultdict(int); streams = collections.defaultdict(set)
    for s, e, k, q, st in ph:
        busy[q] += e - s; streams[q].add(st)
    print(f"phase {1e-6 * (t1 - t0):8.1f} ms, {len(ph):6d} batched launches, mostly {names[0][0][:40]}: " +
          "  ".join(f"q{q}: {100 * b / (t1 - t0):.0f} % (streams {sorted(streams[q])})" for q, b in sorted(busy.items())))
