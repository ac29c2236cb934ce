#!/usr/bin/env python3
"""Parity budget, part 1: census of EXACT float-distance ties in the association rounds of BASELINE configs C1-C4 (CPU only, the oracle as counter).

Why: FLANN's answer among exactly equidistant points of one ring depends on its traversal order, which the reference does not pin
(SURVEY.md Appendix B4); the oracle and the HIP path both take the lowest index.  If no such tie occurs on a winning ring, the choice
cannot be observed in any table or pose of that workload.  Cross-ring ties are decided by the reference's own strict '<'
(velo.h:836,843) and are counted for completeness.

  python tools/parity_budget.py [c1 c2 c3 c4] [--out profiles/r04_parity_budget.json]
Part 2 (unless --no-variants): the pose of the whole call under each alternative reading of un-pinned third-party behaviour
(oracle switches qr / ftol_apply / tie_high, see tests/oracle_lib.py::Oracle.set_variant) against the default restatement.
Every round is taken at the pose the oracle's own frame-to-frame loop holds when it associates (6 rounds per pair)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import velo_amd  # noqa: F401,E402
from velo_amd import synth  # noqa: E402
import oracle_lib as ol  # noqa: E402


def workload(name):
    if name == "c4":
        d = synth.scan_to_map()
        return d, 1, None
    d = synth.scan_pair()
    if name == "c1":
        return d, 200, None
    if name == "c3":
        return d, 1, synth.stereo_matches(1000, x_true=d["x_true"])
    return d, 1, None


def census(d, skip, vis, threads):
    o = ol.Oracle(threads=threads, icp_skip=skip)
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    if vis is not None:
        o.set_visual(vis)
    x = np.array(d["x0"], dtype=np.float64)
    rounds = []
    P = o.params
    for it in range(1, P.f2f_iterations + 1):
        o.build_visual(x, it)
        for _ in range(P.icp_iterations):
            c = o.tie_census(x, it)
            o.associate(x, it)
            x, _s = o.solve(x)
            rounds.append(c)
    return rounds, x


def variant_budget(d, skip, vis, threads):
    """Pose of the whole frame-to-frame call under each alternative third-party behaviour, against the default restatement."""
    def run(normal=None, **variant):
        o = ol.Oracle(threads=threads, icp_skip=skip)
        o.set_variant(**variant)
        if normal:
            o.set_variant_normal(**normal)
        o.set_target(d["tgt_xyz"], d["tgt_off"])
        o.set_source(d["src_xyz"], d["src_off"])
        if vis is not None:
            o.set_visual(vis)
        x, _T, s = o.frame_to_frame(d["x0"])
        skips = o.set_variant_normal(**(normal or {}))           # (reads and resets the counter of ||N|| < 1e-5 skips, velo.h:873)
        return x, [int(s.solves[i].evaluations) for i in range(s.n_solves)], skips, [int(s.solves[i].n_icp_valid) for i in range(s.n_solves)]
    x0, e0, k0, v0 = run()
    out = {}
    # the plane normal's float arithmetic under the other readings of Eigen (velo.h:868-874): the 3-element reduction order of norm(), an
    # FMA-contracted cross product -- effect on the pose, on the evaluation counts, on the rows each solve holds and on the ||N|| skips
    for name, normal in (("norm_split", dict(norm_split=True)), ("cross_fma", dict(cross_fma=True))):
        x, e, k, v = run(normal=normal)
        out[name] = dict(dt_m=float(np.linalg.norm(x[3:] - x0[3:])), dw_rad=float(np.linalg.norm(x[:3] - x0[:3])), evaluations=e, same_evaluation_counts=(e == e0),
                         norm_skips=k, default_norm_skips=k0, same_valid_counts=(v == v0))
    for name in ("qr", "ftol_apply", "tie_high"):
        x, e, _k, _v = run(**{name: True})
        out[name] = dict(dt_m=float(np.linalg.norm(x[3:] - x0[3:])), dw_rad=float(np.linalg.norm(x[:3] - x0[:3])),
                         evaluations=e, same_evaluation_counts=(e == e0))
    out["default_evaluations"] = e0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c1", "c2", "c3", "c4"])
    ap.add_argument("--threads", type=int, default=ol.max_threads())
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_parity_budget.json"))
    ap.add_argument("--no-variants", action="store_true", help="census only")
    a = ap.parse_args()
    out = {}
    for name in a.configs:
        t = time.time()
        d, skip, vis = workload(name)
        rounds, x = census(d, skip, vis, a.threads)
        tot = {k: int(sum(r[k] for r in rounds)) for k in rounds[0]}
        out[name] = dict(icp_skip=skip, n_target=int(d["tgt_off"][-1]), rounds=rounds, total=tot, final_pose=[float(v) for v in x])
        print(f"{name}: {tot}  ({time.time() - t:.1f} s)", flush=True)
        if not a.no_variants:
            out[name]["variants"] = variant_budget(d, skip, vis, a.threads)
            print(f"{name}: {out[name]['variants']}  ({time.time() - t:.1f} s)", flush=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
