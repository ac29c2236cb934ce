"""Rows R1-R5 of SURVEY.md 8(a) (and the triangulation functors of 8(f) row 4) against vectors produced FROM THE REFERENCE'S OWN FUNCTOR
TEXT: tests/golden/functors_ref.npz, made by tests/golden/make_functor_ref.py, which reads /root/reference/costfunctions.h at run time,
turns every operator() body into Python statement by statement and evaluates it in double (Jacobians by complex step).  This is the one
place where the oracle and the HIP path are held to something the reference itself wrote, not to this repository's reading of it;
ceres::AngleAxisRotatePoint stays [3P] (restated from Ceres' published rotation.h in the generating script).

  CPU, everywhere        the oracle's functors (values + dual-number Jacobians) equal the fixture;
  CPU, authoring box     the committed fixture is bit for bit what the script produces from the reference today;
  GPU                    velo_evaluate_functors (seam 2) equals the fixture."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
import velo_amd  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "functors_ref.npz")
DIMS = {0: 3, 1: 2, 2: 2, 3: 1, 4: 1}
TRI_3D, TRI_2D = 0, 1                      # VELO_TRI_OBS_3D / VELO_TRI_OBS_2D (include/velo_hip.h:161-162)


def _rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1.0, float(np.max(np.abs(b)))))


def test_oracle_functors_equal_the_reference_derived_vectors():
    z = np.load(FIX)
    assert len(z["kinds"]) == 60 and set(z["kinds"].tolist()) == {0, 1, 2, 3, 4}
    for k, c, x, r, J in zip(z["kinds"], z["consts"], z["x"], z["r"], z["J"]):
        d = DIMS[int(k)]
        ro, Jo = ol.functor(int(k), c, x)
        assert _rel(np.asarray(ro)[:d], r[:d]) <= 1e-14, (int(k), x)
        assert _rel(np.asarray(Jo).reshape(-1, 6)[:d], J[:d]) <= 1e-13, (int(k), x)
    # every branch of the rotation is in there: zero, below and above the first-order switch, large angles
    th2 = np.sum(z["x"][:, :3] ** 2, axis=1)
    assert np.any(th2 == 0) and np.any((th2 > 0) & (th2 < np.finfo(float).eps)) and np.any(th2 > 1.0)


def test_oracle_triangulation_functors_equal_the_reference_derived_vectors():
    z = np.load(FIX)
    assert z["tri_is3d"].sum() == 10 and len(z["tri_is3d"]) == 20
    for is3d, cam, s, t, x, r, J in zip(z["tri_is3d"], z["tri_cam"], z["tri_s"], z["tri_t"], z["tri_x"], z["tri_r"], z["tri_J"]):
        d = 3 if is3d else 2
        ro, Jo = ol.tri_functor(TRI_3D if is3d else TRI_2D, cam, s, t, x)
        assert _rel(ro, r[:d]) <= 1e-14 and _rel(Jo, J[:d]) <= 1e-13, (bool(is3d), cam)


@pytest.mark.skipif(not os.path.exists("/root/reference/costfunctions.h"), reason="the reference checkout is not on this box")
def test_committed_fixture_is_what_the_reference_text_yields_today():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_functor_ref as M
    fresh = M.generate()
    z = np.load(FIX)
    assert set(fresh) == set(z.files)
    for k in z.files:
        assert np.array_equal(np.asarray(fresh[k]), z[k]), k
    # the transpiler saw what it should: seven functors, the frame ones with the parameter counts of costfunctions.h:19-28,58-64,89-97,128-136,170-177
    parsed = M.parse_functors(open(M.REF_HEADER).read())
    assert {n: len(parsed[n][0]) for n in M.FRAME_FUNCTORS} == {"cost3D3D": 6, "cost3D2D": 8, "cost2D3D": 8, "cost2D2D": 7, "cost3DPD": 9}
    assert len(parsed["triangulation2D"][0]) == 11 and len(parsed["triangulation3D"][0]) == 9


def test_transpiler_refuses_anything_but_arithmetic():
    """the reference is untrusted text: a functor body that held anything but plain arithmetic is refused before it runs"""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_functor_ref as M
    ok = "M = [0.0, 0.0, 0.0]\n_rot(x, m, M)\nM[0] += x[3] - s_x * M[2]\nresidual[0] = _sqrt(M[0] * M[0]) / 2.0"
    M.check_arithmetic_only(ok, "<ok>")
    for bad in ("residual[0] = __import__('os').system('true')", "residual[0] = x.__class__", "import os", "residual[0] = open('f')",
                "residual[0] = [v for v in x]", "residual[0] = (lambda: 1)()", "residual[0] = 'text'", "residual[0] = _rot.__globals__",
                "residual[0] = __builtins__", "residual[0] = x if x else 0", "residual[0] = _sqrt(v=1)"):
        with pytest.raises((ValueError, SyntaxError)):
            M.check_arithmetic_only(bad, "<bad>")
    # and a body that passed the check still runs without builtins
    f = M.make_callable("f", {"f": (["s_x"], ["x", "residual"], "residual[0] = x[0] * s_x")})
    assert f([2.0], [3.0], [0.0])[0] == 6.0


@pytest.mark.gpu
def test_hip_functor_batch_equals_the_reference_derived_vectors(hip_lib):
    from velo_amd import api
    z = np.load(FIX)
    c = api.Context(0)
    xs = np.unique(z["x"], axis=0)
    checked = 0
    for x in xs:
        sel = np.nonzero(np.all(z["x"] == x, axis=1))[0]
        r, J = c.evaluate_functors(z["kinds"][sel], z["consts"][sel], x)
        for i, j in enumerate(sel):
            d = DIMS[int(z["kinds"][j])]
            assert _rel(r[i, :d], z["r"][j, :d]) <= 1e-13, (int(z["kinds"][j]), x)
            assert _rel(J[i, :d], z["J"][j, :d]) <= 1e-12, (int(z["kinds"][j]), x)
            checked += 1
    assert checked == 60
    c.close()
