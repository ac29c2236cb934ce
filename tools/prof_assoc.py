#!/usr/bin/env python3
"""Dev tool for rocprofv3: a few association rounds of one variant (env VELO_ASSOC_VARIANT / VELO_CLUSTER_W)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
import velo_amd
from velo_amd import api, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
d = synth.scan_to_map(2_000_000) if wl == "c4" else synth.scan_pair()
c = api.Context(0, icp_skip=1)
c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
for _ in range(3):
    c.associate(d["x0"], 1); c.associate(d["x_true"], 1); c.associate(d["x_true"], 2)
c.close()
