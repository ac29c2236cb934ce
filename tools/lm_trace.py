#!/usr/bin/env python3
"""Dev tool: time line of the LM chain of ONE pair in flight (C2), from the stage stamps of the diagnostics build
(VELO_LM_TRACE=1: s_memrealtime, 100 MHz).  Per evaluation, microseconds since the first evaluation's first workgroup started:
stage first/last over the workgroups, and the mean duration of every hop of the chain."""
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
hops = {}
merged = False
for sv in solves:
    for e, st in enumerate(sv):
        if st[4] is None and st[0] is not None:              # one launch per iteration (lm_iter_kernel)
            merged = True
            if any(st[k] is None for k in (0, 5, 6, 7, 8, 2, 3)):
                continue
            h = {"first WG in -> last WG in": st[0][1] - st[0][0],
                 "entered -> previous partial rows + state in LDS (last WG)": st[5][1] - st[0][1],
                 "-> 28 sums ready (last WG)": st[6][1] - st[5][1],
                 "-> transition done (last WG)": st[7][1] - st[6][1],
                 "-> eval point built (last WG)": st[8][1] - st[7][1],
                 "-> rows done (last WG)": st[2][1] - st[8][1],
                 "-> partial row stored (last WG)": st[3][1] - st[2][1]}
            if e + 1 < len(sv) and sv[e + 1][0] is not None:
                h["boundary: last partial stored -> next launch's first WG in"] = sv[e + 1][0][0] - st[3][1]
                h["whole iteration (first WG in -> next launch's first WG in)"] = sv[e + 1][0][0] - st[0][0]
        else:
            if any(st[k] is None for k in range(10)):
                continue
            h = {"sweep: first WG in -> last WG in": st[0][1] - st[0][0],
                 "sweep: entered -> eval point in LDS (last WG)": st[1][1] - st[0][1],
                 "sweep: eval point -> rows done (last WG)": st[2][1] - st[1][1],
                 "sweep: rows done -> partial row stored (last WG)": st[3][1] - st[2][1],
                 "boundary: last partial stored -> step entered": st[4][0] - st[3][1],
                 "step: entered -> partials + state in LDS": st[5][0] - st[4][0],
                 "step: -> 28 sums ready": st[6][0] - st[5][0],
                 "step: -> transition done": st[7][0] - st[6][0],
                 "step: -> next eval point built": st[8][0] - st[7][0],
                 "step: -> stores issued": st[9][0] - st[8][0]}
            if e + 1 < len(sv) and sv[e + 1][0] is not None:
                h["boundary: stores issued -> next sweep's first WG in"] = sv[e + 1][0][0] - st[9][0]
                h["whole iteration (sweep in -> next sweep in)"] = sv[e + 1][0][0] - st[0][0]
        for k, v in h.items():
            hops.setdefault(k, []).append(v)
print(f"{len(solves)} solves, {sum(len(s) for s in solves)} evaluations, {'one launch' if merged else 'two launches'} per iteration")
for k, v in hops.items():
    v = sorted(v)
    print(f"  {k:66s} mean {sum(v) / len(v):6.2f} us   median {v[len(v) // 2]:6.2f}   max {v[-1]:6.2f}")
if not solves:
    print(out[-3000:])
