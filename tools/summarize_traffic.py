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
    json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
    print(res)
