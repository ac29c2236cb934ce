#!/usr/bin/env python3
"""Dev tool: time line of the LM chain of ONE pair in flight (C2), from the stage stamps of the diagnostics build
(VELO_LM_TRACE=1: s_memrealtime, 100 MHz).  Per evaluation, microseconds since the first evaluation's first workgroup started:
stage first/last over the workgroups --
  0 sweep entered | 1 state read | 2 rotation constants ready | 3 rows done | 4 partial row stored
  5 step entered  | 6 sums ready | 7 state written
and the mean duration of every hop of the chain."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import velo_amd
    from velo_amd import api, synth
    d = synth.scan_pair()
    c = api.Context(0, icp_skip=1)
    c.set_target(d["tgt_xyz"], d["tgt_off"]); c.set_source(d["src_xyz"], d["src_off"])
    for _ in range(3):
        c.frame_to_frame(d["x0"])
    c.close()
    sys.exit(0)
from velo_amd import build
lib = build.build_hip(diagnostics=True)
env = dict(os.environ, VELO_LM_TRACE="1", VELO_LIB_PATH=lib)
out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True).stderr
solves, cur = [], []
for line in out.splitlines():
    m = re.match(r"\[velo lm trace\] eval +(\d+): (.*)", line)
    if not m:
        continue
    if int(m.group(1)) == 0 and cur:
        solves.append(cur); cur = []
    cur.append([tuple(float(v) for v in tok.split("/")) if tok != "-" else None for tok in m.group(2).split()])
if cur:
    solves.append(cur)
solves = solves[-6:]                      # the last call's six solves
names = ["sweep entered", "state read", "rotation ready", "rows done", "partials stored", "step entered", "sums ready", "state written"]
hops = {}
for sv in solves:
    for e, st in enumerate(sv):
        if any(s is None for s in st):
            continue
        seq = [("launch->first WG in", None), ]
        h = {"sweep: first WG in -> last WG in": st[0][1] - st[0][0],
             "sweep: entered -> state read (last WG)": st[1][1] - st[0][1],
             "sweep: state read -> rotation constants (last WG)": st[2][1] - st[1][1],
             "sweep: rotation -> rows done (last WG)": st[3][1] - st[2][1],
             "sweep: rows done -> partial stored (last WG)": st[4][1] - st[3][1],
             "boundary: last partial stored -> step entered": st[5][0] - st[4][1],
             "step: entered -> sums ready": st[6][0] - st[5][0],
             "step: sums -> state written": st[7][0] - st[6][0]}
        if e + 1 < len(sv) and sv[e + 1][0] is not None:
            h["boundary: state written -> next sweep's first WG in"] = sv[e + 1][0][0] - st[7][0]
            h["whole iteration (sweep in -> next sweep in)"] = sv[e + 1][0][0] - st[0][0]
        for k, v in h.items():
            hops.setdefault(k, []).append(v)
print(f"{len(solves)} solves, {sum(len(s) for s in solves)} evaluations")
for k, v in hops.items():
    v = sorted(v)
    print(f"  {k:58s} mean {sum(v) / len(v):6.2f} us   median {v[len(v) // 2]:6.2f}   max {v[-1]:6.2f}")
if not solves:
    print(out[-3000:])
