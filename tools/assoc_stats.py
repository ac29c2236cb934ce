#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import velo_amd
from velo_amd import api, synth
os.environ["VELO_DEBUG_SKIP"] = "16"
os.environ["VELO_CLUSTER_W"] = sys.argv[1] if len(sys.argv) > 1 else "6"
d = synth.scan_pair()
for it, x in ((1, d["x0"]), (1, d["x_true"]), (2, d["x_true"])):
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    c.associate(x, it)
    print("iter", it, "-> [clusters, phases(rowchunks), staged, staged_phase2, rows, sum_e] printed at close:", flush=True)
    c.close()
