#!/usr/bin/env python3
"""Dev tool (GPU box, diagnostics build): per-section cycle totals of the tube kernel's lead wave for every association round of one
frame_to_frame call (VELO_DEBUG_SKIP=8 + VELO_DEBUG_EACH: sections 0 set-up, 1 intervals, 2 run list, 3 staging, 4 sweep, 5 barrier behind
the sweep, 6 merge, 7 finish; summed over the round's workgroups, shader cycles)."""
import os, sys
os.environ["VELO_DEBUG_SKIP"] = "8"; os.environ["VELO_DEBUG_EACH"] = "1"; os.environ["VELO_ASSOC_VARIANT"] = "5"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702,F401
import velo_amd  # noqa: F401
from velo_amd import api, synth
d = synth.scan_pair()
c = api.Context(0, icp_skip=1)
c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
c.frame_to_frame(d["x0"])
c.close()
