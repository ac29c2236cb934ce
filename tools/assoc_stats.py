#!/usr/bin/env python3
"""Dev tool: per-launch counters of the association kernel (VELO_DEBUG_SKIP=16): clusters, row chunks, staged candidates,
phase-2 staged candidates, rows, sum of expansions; plus per-workgroup duration percentiles (VELO_DEBUG_SKIP=32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
import velo_amd
from velo_amd import api, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
os.environ["VELO_DEBUG_SKIP"] = "48"
d = synth.scan_to_map(2_000_000) if wl == "c4" else synth.scan_pair()
for it, x in ((1, d["x0"]), (1, d["x_true"]), (2, d["x_true"])):
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    c.associate(x, it)
    print("iter", it, "x0" if x is d["x0"] else "x_true", "-> dbg = [clusters, rowchunks, staged, staged_phase2, rows, sum_e]", flush=True)
    c.close()
