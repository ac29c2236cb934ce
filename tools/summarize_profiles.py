#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (written by tools/gpu_profile.sh on the GPU box) into the tracked summaries under profiles/."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
lines = []
f = sorted(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    wl = tag.rsplit("_", 1)[1] if tag.rsplit("_", 1)[-1] in ("c1", "c3", "c4") else "c2"
    st, wu = 20, 5
    try:
        dd = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
        st, wu = dd["steps"], dd["warmup"]
    except Exception:
        pass
    lines.append(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {st} --warmup {wu} --no-cpu-baseline --no-legs --workload {wl}  (default batch: 8 pairs in flight; the driver's command without its CPU leg and side legs)")
    for r in rows[:8]:
        lines.append(f"  {r['Name'].split('(')[0][:58]:58s} calls {r['Calls']:>6}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
b = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(b):
    try:
        d = json.loads(open(b).read().strip().splitlines()[-1])
        lines.append(f"bench line under rocprof: value {d['value']:.1f} {d['unit']}; kernels by share of the timed region (HIP events, every 8th launch bracketed): " +
                     "; ".join(f"{k['kernel']} {100 * k['share']:.0f} % avg {k['avg_launch_us']:.1f} us frac {k['frac']:.4f}" for k in d.get("kernels", [])))
    except Exception as e:
        lines.append(f"(bench line unreadable: {e})")
# the same kernel split by phase of the bench run: the --stats average mixes the timed region (8 pairs in flight) with the
# single-pair leg; the bench line's figure is the timed region only
f = sorted(glob.glob(os.path.join(src, "trace", "*", "*kernel_trace.csv")), key=os.path.getmtime, reverse=True)
if f and os.path.exists(b):
    try:
        d = json.loads(open(b).read().strip().splitlines()[-1])
        cfg = d["config"]
        per_ctx = 12 * cfg["Nq"] + 12 * cfg["Nt"] + 28 * cfg["Nq"]                     # B_assoc of one context's round
        arow = next((kk for kk in d.get("kernels", []) if kk["kernel"].startswith("assoc")), d["roofline"])
        k = max(1, round(arow["algorithmic_bytes_per_launch"] / per_ctx))                # contexts served by one launch
        per_step = cfg["pairs_in_flight_per_gpu"] * 6 // k
        rows = sorted((r for r in csv.DictReader(open(f[0])) if "assoc_search" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        w0, t0 = d["warmup"] * per_step, (d["warmup"] + d["steps"]) * per_step
        timed, alone = sorted(dur[w0:t0]), dur[t0:]
        if timed and alone:
            pc = lambda q: timed[min(len(timed) - 1, int(q * len(timed)))]
            lines.append(f"association launches in the kernel trace by phase: timed region ({len(timed)} launches of {k} contexts each, 8 pairs in flight) avg {sum(timed)/len(timed):.1f} us"
                         f" (p10 {pc(.1):.1f} / p50 {pc(.5):.1f} / p90 {pc(.9):.1f}); single-pair leg ({len(alone)} launches) avg {sum(alone)/len(alone):.1f} us")
            lines.append(f"  bench line, same run: timed region {arow['avg_launch_us']:.1f} us (HIP events of hipExtLaunchKernelGGL: the start event is a marker ahead of the"
                         f" kernel, so the bracket adds the command processor's hand-over between the two packets: ~15 us with 4+ busy queues, ~11 us under the"
                         f" profiler's interception even alone, <1 us alone without it -- a plain bench.py run reports 58-62 us for the single-pair leg);"
                         f" single-pair leg {d['single_pair']['assoc_avg_launch_us']:.1f} us")
    except Exception as e:
        lines.append(f"(kernel trace split unavailable: {e})")
if f:
    try:
        allrows = list(csv.DictReader(open(f[0])))
        for pat in ("eval_step_batch", "lm_iter_kernel"):
            dur = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in allrows if pat in r["Kernel_Name"])
            live = [x for x in dur if x > 6.0]
            if live:
                lines.append(f"{pat}* launches in the kernel trace: {len(dur)}, of which live (> 6 us) {len(live)}: avg {sum(live)/len(live):.1f} us (p10 {live[int(.1*len(live))]:.1f} / p50 {live[len(live)//2]:.1f} / p90 {live[int(.9*len(live))]:.1f}); all launches avg {sum(dur)/len(dur):.1f} us")
    except Exception as e:
        lines.append(f"(LM launch split unavailable: {e})")
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_fetch_b2", "FETCH_SIZE"), ("pmc_write_b2", "WRITE_SIZE")):
    f = sorted(glob.glob(os.path.join(src, name, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    lines.append(f"{counter} per dispatch ({name}: {'one lock-step group of two contexts alone' if name.endswith('_b2') else 'one pair in flight'}; KiB as rocprofv3 reports it; gfx950: FETCH_SIZE counts wide coalesced reads at 1/2):")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
        lines.append(f"  {k[:58]:58s} n {len(v):4d}  mean {sum(v)/len(v):12.1f}")
f = sorted(glob.glob(os.path.join(src, "pmc_sq", "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
    lines.append("SQ counters, sum over XCDs per dispatch, mean over dispatches:")
    for k in agg:
        if "assoc" in k or "lm_iter" in k or "eval" in k:
            parts = []
            for cn, vals in agg[k].items():
                per = collections.defaultdict(float)
                for did, v in vals:
                    per[did] += v
                parts.append(f"{cn}={sum(per.values())/len(per):.3g}")
            lines.append(f"  {k[:50]}: " + " ".join(parts))
open(os.path.join(dst, f"{tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
