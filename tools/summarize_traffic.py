#!/usr/bin/env python3
"""profiles/<tag>_traffic.json: measured HBM-side bytes per association launch from the separate FETCH_SIZE / WRITE_SIZE
PMC passes (rocprofv3 reports KiB; gfx950 counts wide coalesced reads at 1/2 -- MI355X_MICROARCH.md, HBM section -- so the
read side is doubled).  bench.py copies this number into roofline.traffic when the file exists."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, name, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
    vals = collections.defaultdict(float)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] == counter and "assoc_search" in r["Kernel_Name"]:
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    out[counter] = (sum(vals.values()) / len(vals)) if vals else None
if out["FETCH_SIZE"] is not None and out["WRITE_SIZE"] is not None:
    res = {"kernel": "assoc_search_v5_kernel", "fetch_KiB_raw": out["FETCH_SIZE"], "write_KiB": out["WRITE_SIZE"],
           "read_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B)",
           "traffic_bytes_per_launch": (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0,
           "source": f"gpurun_out/{tag}/pmc_fetch + pmc_write (rocprofv3 --pmc, separate passes), bench.py --batch 1"}
    # instruction counters of the same kernel (pmc_sq pass): the bound that actually binds (the data is L2 / Infinity-Cache resident)
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "pmc_sq", "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
    if f:
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f[0])):
            if "assoc_search" in r["Kernel_Name"]:
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        res["sq_per_launch"] = {k: sum(v.values()) / len(v) for k, v in per.items()}
    # every kernel of the PMC passes (one pair in flight; "_b2": one lock-step group of two contexts alone on the chip): mean HBM-side bytes
    # per LIVE launch (launches behind the end of a solve move nothing: those below a quarter of the kernel's largest launch are left out)
    by = {}
    for suffix in ("", "_b2"):
        tot = collections.defaultdict(lambda: collections.defaultdict(float))
        for name, counter, mul in (("pmc_fetch" + suffix, "FETCH_SIZE", 2.0), ("pmc_write" + suffix, "WRITE_SIZE", 1.0)):
            g = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, name, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
            if not g:
                continue
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(g[0])):
                if r["Counter_Name"] == counter:
                    per[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("velo::", "").split("<")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for k, d in per.items():
                v = list(d.values())
                live = [x for x in v if x >= 0.25 * max(v)] or v
                tot[k][counter] = mul * 1024.0 * sum(live) / len(live)
        for k, d in tot.items():
            if k.startswith("__amd") or k in by:
                continue
            by[k] = d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)
    res["traffic_by_kernel"] = by

    # round 4: the passes taken on the bench's own command shape (8 pairs in flight): bytes and SQ / busy counters per launch, keyed by
    # the kernel instantiation that actually ran.  Launches behind the end of a solve (the chain's margin) move nothing and are left out
    # of the byte figures like above; the SQ / busy counters are means over ALL launches of the name.
    def short(name):
        return name.split("(")[0].replace("void ", "").replace("velo::", "").split("<")[0]

    def newest(name):
        g = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, name, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)
        return g[0] if g else None

    load = {}
    tot = collections.defaultdict(dict)
    for name, counter, mul in (("pmc_fetch_b8", "FETCH_SIZE", 2.0), ("pmc_write_b8", "WRITE_SIZE", 1.0)):
        f8 = newest(name)
        if not f8:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f8)):
            if r["Counter_Name"] == counter:
                per[short(r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, d in per.items():
            v = list(d.values())
            live = [x for x in v if x >= 0.25 * max(v)] or v
            tot[k][counter] = mul * 1024.0 * sum(live) / len(live)
    for k, d in tot.items():
        if not k.startswith("__amd") and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            load[k] = d["FETCH_SIZE"] + d["WRITE_SIZE"]
    if load:
        res["traffic_by_kernel_in_flight"] = load
        res["in_flight_source"] = f"gpurun_out/{tag}/pmc_fetch_b8 + pmc_write_b8: bench.py's default batch (8 pairs in flight), separate passes"
    counters = {}
    for name in ("pmc_sq_b8", "pmc_busy_b8"):
        f8 = newest(name)
        if not f8:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
        for r in csv.DictReader(open(f8)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, cs in per.items():
            if k.startswith("__amd"):
                continue
            counters.setdefault(k, {}).update({cn: sum(d.values()) / len(d) for cn, d in cs.items()})
            counters[k]["launches"] = max(len(d) for d in cs.values())
    if counters:
        keep = {k: v for k, v in counters.items() if any(t in k for t in ("assoc_search", "eval_step", "lm_iter", "lm_solve", "assoc_direct", "target_ingest", "scan_lookback", "grid_"))}
        res["counters_by_kernel_in_flight"] = keep
    json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
    print(res)
