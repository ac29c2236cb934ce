#!/usr/bin/env python3
"""Dev tool: counters of a COLD and of a WARM association round of the tube kernel (VELO_DEBUG_SKIP=16 accumulates per context;
the warm round is the difference between a two-round and a one-round context)."""
import os, sys, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _diag  # noqa: E702  the A/B switches exist in the diagnostics build only
    import velo_amd
    from velo_amd import api, synth
    d = synth.scan_pair()
    it = int(sys.argv[3])
    x0, x1 = d["x0"], d["x_true"]
    xa = x0 if it == 1 else x1
    xb = x0 + 0.5 * (x1 - x0) if it == 1 else x1 + 1e-3
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    c.associate(xa, it)
    if sys.argv[2] == "2": c.associate(xb, it)
    c.close()
    sys.exit(0)
env = dict(os.environ, VELO_DEBUG_SKIP="16", VELO_ASSOC_VARIANT="5")
for it in (1, 2):
    vals = []
    for rounds in ("1", "2"):
        out = subprocess.run([sys.executable, __file__, "--child", rounds, str(it)], env=env, capture_output=True, text=True).stderr
        m = re.search(r"setup (\d+) cluster (\d+) runlist (\d+) stage (\d+) sweep (\d+)", out)
        vals.append([int(v) for v in m.groups()])
    names = ["clusters", "row chunks", "staged", "staged phase 2", "rows"]
    print(f"iter {it}  cold:", dict(zip(names, vals[0])), "\n        warm:", dict(zip(names, [b - a for a, b in zip(*vals)])))
