#!/usr/bin/env python3
"""Dev tool: where a 64-query group of the association kernel spends its cycles, round by round along frame_to_frame's real pose
sequence on C2 (diagnostic instantiation, VELO_DEBUG_SKIP=8: wave-0 cycle stamps per section; =32: workgroup start/end times).
A context prints its totals when it is destroyed, so round k is the difference of the contexts that ran k and k-1 rounds."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
XS = [[0.0, 0.0, 0.0, 0.0, 0.0, 1.0],
      [-0.0022185673765433398, -0.01320751722503497, 0.0011630384070612238, -0.01822384240595095, -0.009697824384943644, 0.999731078292073],
      [-0.00235036892604132, -0.018652350621762518, 0.0017034178513706116, -0.02417041069789526, -0.012049963021747635, 0.9990775463586585],
      [-0.002325648599613658, -0.019726646953725255, 0.0019322317905635137, -0.025238286869366126, -0.01244598455515236, 0.9982641886663832],
      [-0.0021390492748765886, -0.019928092057226374, 0.002070537052540992, -0.02547107724038336, -0.01064705708617851, 0.9983976752028932],
      [-0.0020959990910373962, -0.019976930496238523, 0.0021406826256700853, -0.025498945008825564, -0.010253558573431909, 0.9985500856943826]]
ITER = [1, 1, 1, 2, 2, 2]


if len(sys.argv) > 1 and sys.argv[1] == "child":
    import velo_amd
    from velo_amd import api, synth
    d = synth.scan_pair()
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    for rep in range(2):                       # second pass = steady state (buffers allocated); seeds reset by the new source
        if rep: c.set_source(d["src_xyz"], d["src_off"])
        for k in range(6):
            c.associate(XS[k], ITER[k])
            if os.environ.get("VELO_DEBUG_SKIP") == "32":
                c.close(); c = api.Context(0, icp_skip=1)      # the per-workgroup times are printed when the context goes
                c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
                for j in range(k + 1):
                    c.associate(XS[j], ITER[j])
    c.close()
    sys.exit(0)

from velo_amd import build
lib = build.build_hip(diagnostics=True)
names = ["setup", "cluster", "runlist", "stage", "sweep", "sweepbar", "merge", "finish"]
env = dict(os.environ, VELO_DEBUG_SKIP="8", VELO_DEBUG_EACH="1", VELO_ASSOC_VARIANT="5", VELO_LIB_PATH=lib)
out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True).stderr
rows = re.findall(r"\[velo dbg launch\] groups (\d+) iter (\d+): (.*)", out)
for k, (g, it, vals) in enumerate(rows[6:]):
    v = [int(x) for x in vals.split()]
    print(f"round {k + 1} (iter {it}): wave-0 cycles per group: " + " ".join(f"{nm} {x / int(g):.0f}" for nm, x in zip(names, v)) + f" | total {sum(v) / int(g):.0f}", flush=True)
if not rows:
    print(out[-2000:])
