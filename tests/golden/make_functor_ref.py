#!/usr/bin/env python3
"""tests/golden/functors_ref.npz -- functor vectors produced FROM THE REFERENCE'S OWN FUNCTOR TEXT.

The reference cannot be built here (no Ceres, no network) and ships no vectors, but the bodies of its residual functors
(costfunctions.h:17-220: cost3DPD, cost3D3D, cost3D2D, cost2D3D, cost2D2D; costfunctions.h:288-375: triangulation2D / 3D) are a
dozen lines of plain arithmetic each.  This script READS /root/reference/costfunctions.h at run time (the authoring container
only), turns every `operator()` body into a Python function statement by statement (declarations, assignments, `T(...)` casts, the
one library call), evaluates it in IEEE double on seeded inputs and stores inputs + residuals + Jacobians.  Nothing of the
reference's text is stored -- the fixture is numbers.

What is and is not pinned by it:
  * pinned: the functors' arithmetic as the reference wrote it -- operand order, signs, which offsets are added where, the
    un-normalised epipolar form of cost2D2D, the inverse-direction form of cost2D3D;
  * [3P] ceres::AngleAxisRotatePoint is not under /root/reference: it is restated here from Ceres' published rotation.h (Rodrigues
    for theta^2 > DBL_EPSILON, first order below) -- the same statement the oracle makes;
  * Jacobians: by complex-step differentiation of the SAME transpiled code (h = 1e-30: exact to rounding for these analytic bodies) --
    the derivative Ceres' autodiff computes, obtained without a dual-number type.

    python tests/golden/make_functor_ref.py            # writes tests/golden/functors_ref.npz
tests/test_functor_ref.py checks the oracle against the fixture everywhere, the HIP path on the GPU box, and -- where the reference is
present -- that the committed fixture is what this script produces from it."""
import cmath
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_HEADER = "/root/reference/costfunctions.h"
FRAME_FUNCTORS = ("cost3D3D", "cost3D2D", "cost2D3D", "cost2D2D", "cost3DPD")        # ResidualType order + the point-to-plane block
TRI_FUNCTORS = ("triangulation2D", "triangulation3D")
DBL_EPSILON = float(np.finfo(np.float64).eps)


# ---- [3P] ceres/rotation.h, restated (works on complex numbers for the complex-step derivative) -----------------------------------
def angle_axis_rotate_point(w, p, out):
    theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
    if complex(theta2).real > DBL_EPSILON:
        theta = cmath.sqrt(theta2)
        costheta, sintheta = cmath.cos(theta), cmath.sin(theta)
        ti = 1.0 / theta
        a = [w[0] * ti, w[1] * ti, w[2] * ti]
        cx = [a[1] * p[2] - a[2] * p[1], a[2] * p[0] - a[0] * p[2], a[0] * p[1] - a[1] * p[0]]
        tmp = (a[0] * p[0] + a[1] * p[1] + a[2] * p[2]) * (1.0 - costheta)
        for i in range(3):
            out[i] = p[i] * costheta + cx[i] * sintheta + a[i] * tmp
    else:
        cx = [w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]]
        for i in range(3):
            out[i] = p[i] + cx[i]


# ---- the reference's functor text -> Python ----------------------------------------------------------------------------------------
def _split_top(s, sep):
    """split at `sep` outside (), [], {}"""
    out, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    out.append("".join(cur))
    return [p.strip() for p in out if p.strip()]


def _expr(e):
    e = re.sub(r"\bT\s*\(", "(", e)                                   # T(value) casts
    e = re.sub(r"\bsqrt\s*\(", "_sqrt(", e)
    return " ".join(e.split())


def parse_functors(text):
    """{name: (constructor parameter names, python source of operator())}"""
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"struct\s+(\w+)\s*\{", text):
        name = m.group(1)
        rest = text[m.end():]
        ctor = re.search(name + r"\s*\(([^)]*)\)\s*:", rest)
        op = re.search(r"bool\s+operator\(\)\s*\(([^)]*)\)\s*const\s*\{", rest)
        if not ctor or not op:
            continue
        params = [p.split()[-1] for p in _split_top(ctor.group(1), ",")]
        args = [re.sub(r"[\s*&]|const|T", "", a.split()[-1]) if a.split() else "" for a in _split_top(op.group(1), ",")]
        args = [a.split()[-1].lstrip("*") for a in _split_top(op.group(1), ",")]
        # the body up to the matching brace
        depth, i = 1, op.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(rest[i], 0)
            i += 1
        body = rest[op.end():i - 1]
        lines = []
        for st in _split_top(body, ";"):
            st = " ".join(st.split())
            if st.startswith("return"):
                continue
            if st.startswith("T "):                                    # declarations: arrays, initialised arrays, initialised scalars
                for item in _split_top(st[2:], ","):
                    arr = re.match(r"(\w+)\s*\[\s*(\d+)\s*\]\s*(?:=\s*\{(.*)\})?$", item)
                    if arr:
                        init = [_expr(v) for v in _split_top(arr.group(3), ",")] if arr.group(3) else ["0.0"] * int(arr.group(2))
                        lines.append(f"{arr.group(1)} = [{', '.join(init)}]")
                    else:
                        nm, val = item.split("=", 1)
                        lines.append(f"{nm.strip()} = {_expr(val)}")
                continue
            call = re.match(r"ceres::AngleAxisRotatePoint\((.*)\)$", st)
            if call:
                a, b, c = _split_top(call.group(1), ",")
                lines.append(f"_rot({a}, {b}, {c})")
                continue
            asg = re.match(r"(\w+(?:\[\d+\])?)\s*(=|\+=|-=|\*=|/=)\s*(.*)$", st)
            assert asg, (name, st)
            lines.append(f"{asg.group(1)} {asg.group(2)} {_expr(asg.group(3))}")
        out[name] = (params, args, "\n".join(lines))
    return out


_ALLOWED_NODES = ("Module", "Assign", "AugAssign", "Expr", "BinOp", "UnaryOp", "Name", "Subscript", "Constant", "List", "Call", "Load", "Store",
                  "Add", "Sub", "Mult", "Div", "USub", "UAdd", "Index")
_ALLOWED_CALLS = ("_rot", "_sqrt")


def check_arithmetic_only(src, where):
    """The reference is untrusted text: the transpiled body may contain ONLY plain arithmetic -- assignments, + - * /, names, constant
    subscripts, numeric constants, list displays and calls of the two helpers this script supplies.  Anything else (attributes,
    imports, other calls, comprehensions, strings ...) is refused before a single statement runs."""
    import ast
    tree = ast.parse(src, filename=where, mode="exec")
    for node in ast.walk(tree):
        kind = type(node).__name__
        if kind not in _ALLOWED_NODES:
            raise ValueError(f"{where}: refusing to evaluate a {kind} node taken from the reference's text")
        if isinstance(node, ast.Call):
            if not (isinstance(node.func, ast.Name) and node.func.id in _ALLOWED_CALLS) or node.keywords:
                raise ValueError(f"{where}: refusing a call other than {_ALLOWED_CALLS}")
        if isinstance(node, ast.Constant) and not isinstance(node.value, (int, float)):
            raise ValueError(f"{where}: refusing a non-numeric constant")
        if isinstance(node, ast.Name) and node.id.startswith("__"):
            raise ValueError(f"{where}: refusing the name {node.id}")
    return tree


def make_callable(name, parsed):
    params, args, src = parsed[name]
    where = f"<{name}::operator()>"
    code = compile(check_arithmetic_only(src, where), where, "exec")

    def f(consts, *call_args):
        """call_args in the order of the reference's operator() parameters; the last one is the residual array (filled)"""
        ns = {p: float(v) for p, v in zip(params, consts)}
        ns.update(_rot=angle_axis_rotate_point, _sqrt=cmath.sqrt, __builtins__={})   # whitelisted arithmetic, no builtins
        for a, v in zip(args, call_args):
            ns[a] = v
        exec(code, ns)
        return ns[args[-1]]
    f.n_params = len(params)
    f.args = args
    return f


def residual_and_jacobian(f, consts, inputs, dim, wrt=0):
    """inputs: list of the functor's input arrays (floats); derivative w.r.t. inputs[wrt] by complex step"""
    r = f(consts, *[list(map(complex, a)) for a in inputs], [0j] * dim)
    res = np.array([complex(v).real for v in r[:dim]])
    n = len(inputs[wrt])
    J = np.zeros((dim, n))
    h = 1e-30
    for i in range(n):
        args = [list(map(complex, a)) for a in inputs]
        args[wrt][i] += 1j * h
        ri = f(consts, *args, [0j] * dim)
        J[:, i] = [complex(v).imag / h for v in ri[:dim]]
    return res, J


DIMS = {"cost3D3D": 3, "cost3D2D": 2, "cost2D3D": 2, "cost2D2D": 1, "cost3DPD": 1, "triangulation2D": 2, "triangulation3D": 3}


def poses(rng, n):
    """pose vectors that visit every branch: zero, below / above the first-order switch, a registration's few hundredths, large"""
    xs = [np.zeros(6), np.array([1e-9, -2e-9, 1.5e-9, 0.3, -0.2, 1.0]), np.array([1.2e-8, 0.0, 0.0, 0.0, 0.0, 0.0]),
          np.array([0.002, -0.02, 0.003, -0.03, -0.01, 1.0]), np.array([0.7, -1.1, 0.4, 2.0, -3.0, 5.0]), np.array([0.0, 3.1, 0.0, 0.1, 0.2, 0.3])]
    while len(xs) < n:
        xs.append(np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 1.0, 3)]))
    return np.array(xs[:n])


def generate(header=REF_HEADER):
    parsed = parse_functors(open(header).read())
    rng = np.random.default_rng(20261003)
    out = {}
    kinds, consts, xs, rs, Js = [], [], [], [], []
    for kind, name in enumerate(FRAME_FUNCTORS):
        f = make_callable(name, parsed)
        assert f.args[0] == "x" and len(f.args) == 2, (name, f.args)
        X = poses(rng, 10)
        for k in range(12):
            if name == "cost3DPD":                                      # point, unit normal, plane offset
                nrm = rng.normal(size=3)
                c = np.concatenate([rng.normal(0, 8, 3), nrm / np.linalg.norm(nrm), rng.normal(0, 8, 3)])
            elif name == "cost2D2D":                                    # canonical 2-D points, camera translation
                c = np.concatenate([rng.uniform(-0.8, 0.8, 2), rng.uniform(-0.8, 0.8, 2), [[0.0, 0.0, 0.0], [-0.537, 0.0, 0.0]][k % 2]])
            elif name == "cost3D3D":
                c = np.concatenate([rng.normal(0, 10, 3), rng.normal(0, 10, 3)])
            else:                                                       # 3-D point, canonical 2-D observation, camera translation
                c = np.concatenate([rng.uniform(-10, 10, 2), [rng.uniform(4, 40)], rng.uniform(-0.8, 0.8, 2), [[0.0, 0.0, 0.0], [-0.537, 0.0, 0.0]][k % 2]])
            assert len(c) == f.n_params, (name, len(c), f.n_params)
            # (the epipolar block divides by the length of the relative translation: poses without any are left to the others)
            x = X[3 + k % (len(X) - 3)] if name == "cost2D2D" else X[k % len(X)]
            r, J = residual_and_jacobian(f, c, [x], DIMS[name])
            kinds.append(kind)
            consts.append(np.pad(c, (0, 9 - len(c))))
            xs.append(x)
            rs.append(np.pad(r, (0, 3 - len(r))))
            Js.append(np.pad(J, ((0, 3 - J.shape[0]), (0, 0))))
    out.update(kinds=np.array(kinds, dtype=np.int32), consts=np.array(consts), x=np.array(xs), r=np.array(rs), J=np.array(Js))
    # the triangulation functors (costfunctions.h:288-375): the unknown is the 3-D point; observation, camera pose (a 6-vector) and
    # camera translation are constructor constants.  triangulation2D(s_x, s_y, cam_0..5, t_x, t_y, t_z), triangulation3D(s_x, s_y, s_z, cam_0..5)
    tk, tcam, ts, tt, tx, tr, tJ = [], [], [], [], [], [], []
    for name in TRI_FUNCTORS:
        f = make_callable(name, parsed)
        assert f.args[0] == "point" and len(f.args) == 2, (name, f.args)
        cams = poses(rng, 10)
        for k in range(10):
            cam = cams[k]
            point = np.concatenate([rng.uniform(-10, 10, 2), [rng.uniform(4, 40)]])
            if name == "triangulation2D":
                s_obs = np.concatenate([rng.uniform(-0.8, 0.8, 2), [0.0]])
                t_cam = np.array([[0.0, 0.0, 0.0], [-0.537, 0.0, 0.0]][k % 2])
                c = np.concatenate([s_obs[:2], cam, t_cam])
            else:
                s_obs = rng.normal(0, 10, 3)
                t_cam = np.zeros(3)
                c = np.concatenate([s_obs, cam])
            assert len(c) == f.n_params, (name, len(c), f.n_params)
            r, J = residual_and_jacobian(f, c, [point], DIMS[name])
            tk.append(name == "triangulation3D"); tcam.append(cam); ts.append(s_obs); tt.append(t_cam); tx.append(point)
            tr.append(np.pad(r, (0, 3 - len(r)))); tJ.append(np.pad(J, ((0, 3 - J.shape[0]), (0, 0))))
    out.update(tri_is3d=np.array(tk), tri_cam=np.array(tcam), tri_s=np.array(ts), tri_t=np.array(tt), tri_x=np.array(tx), tri_r=np.array(tr), tri_J=np.array(tJ))
    return out


if __name__ == "__main__":
    if not os.path.exists(REF_HEADER):
        sys.exit(f"{REF_HEADER} is not on this box: the fixture can only be (re)generated where the reference checkout is")
    vec = generate()
    np.savez(os.path.join(HERE, "functors_ref.npz"), **vec)
    print("wrote functors_ref.npz:", {k: v.shape for k, v in vec.items()})
