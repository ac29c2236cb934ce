#!/usr/bin/env python3
"""Dev tool: workgroup duration distribution of every association round of a scan-to-map registration (2M-point map, production
parameters), with the slowest groups named.  Replays frame_to_frame's loop by hand (associate + solve) so that round k can be the
last launch of a context -- the diagnostics build prints its statistics when the context goes (VELO_DEBUG_SKIP=32)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import velo_amd
    from velo_amd import api, synth
    d = synth.scan_to_map()
    for last in range(1, 7):
        c = api.Context(0, icp_skip=1)
        c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
        x = d["x0"].copy()
        for r in range(last):
            c.associate(x, 1 if r < 3 else 2)
            if r + 1 < last:
                x, _ = c.solve(x)
        sys.stderr.write(f"--- round {last}\n"); sys.stderr.flush()
        c.close()
    sys.exit(0)
from velo_amd import build
lib = build.build_hip(diagnostics=True)
env = dict(os.environ, VELO_DEBUG_SKIP=os.environ.get("VELO_DEBUG_SKIP", "40"), VELO_LIB_PATH=lib)
out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True).stderr
for line in out.splitlines():
    if "velo dbg" in line or line.startswith("--- round"):
        print(line)
