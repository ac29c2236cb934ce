"""A SECOND restatement of one `ceres::Solve` (reference call site velo.h:897-902), written from the text of SURVEY.md
Appendix B1-B3 and from costfunctions.h -- NOT from oracle/velo_oracle.cpp -- so that the oracle's LM iterate sequence has
something independent to be compared with (tests/test_parity_budget.py).

What is deliberately different from the oracle, so that a shared slip cannot hide:
  * derivatives: numpy forward-mode duals over ALL blocks of a kind at once (the oracle: a C++ Jet per block);
  * the rotation: the matrix form  p cos(t) + (w x p) sin(t)/t + w (w.p) (1 - cos(t))/t^2  on the un-normalised vector
    (the oracle follows ceres/rotation.h: normalise the axis first);
  * the linear algebra: the step is the least-squares solution of the AUGMENTED system [J; D] (numpy.linalg.lstsq, an SVD),
    never the 6x6 normal equations; the model change is taken row by row, -(J s).(r + J s / 2);
  * the bookkeeping: one flat loop over the appendix's pseudo-code, its own names, its own termination labels.
Nothing here is used by the product or by the oracle.  Test infrastructure only.
"""
from __future__ import annotations

import numpy as np

K_3D3D, K_3D2D, K_2D3D, K_2D2D, K_3DPD = 0, 1, 2, 3, 4      # ResidualType order of velo.h + the point-to-plane block
DIMS = {K_3D3D: 3, K_3D2D: 2, K_2D3D: 2, K_2D2D: 1, K_3DPD: 1}
EPS = np.finfo(np.float64).eps
DBL_MIN = np.finfo(np.float64).tiny


class Dual:
    """value a: (n,) or (1,), partials v: (n, 6) or (1, 6); broadcasting does the rest."""
    __slots__ = ("a", "v")
    __array_ufunc__ = None          # ndarray <op> Dual defers to Dual.__r<op>__ instead of looping over elements

    def __init__(self, a, v=None):
        self.a = np.atleast_1d(np.asarray(a, dtype=np.float64))
        self.v = np.zeros(self.a.shape + (6,)) if v is None else v

    @staticmethod
    def lift(o):
        return o if isinstance(o, Dual) else Dual(o)

    def __add__(self, o):
        o = Dual.lift(o)
        return Dual(self.a + o.a, self.v + o.v)
    __radd__ = __add__

    def __sub__(self, o):
        o = Dual.lift(o)
        return Dual(self.a - o.a, self.v - o.v)

    def __rsub__(self, o):
        return Dual.lift(o) - self

    def __neg__(self):
        return Dual(-self.a, -self.v)

    def __mul__(self, o):
        o = Dual.lift(o)
        return Dual(self.a * o.a, self.a[..., None] * o.v + o.a[..., None] * self.v)
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Dual.lift(o)
        q = self.a / o.a
        return Dual(q, (self.v - q[..., None] * o.v) / o.a[..., None])

    def sqrt(self):
        r = np.sqrt(self.a)
        return Dual(r, self.v / (2.0 * r)[..., None])

    def sin(self):
        return Dual(np.sin(self.a), np.cos(self.a)[..., None] * self.v)

    def cos(self):
        return Dual(np.cos(self.a), -np.sin(self.a)[..., None] * self.v)


def pose_duals(x):
    out = []
    for i in range(6):
        v = np.zeros((1, 6))
        v[0, i] = 1.0
        out.append(Dual([x[i]], v))
    return out


def rotate(w, p):
    """AngleAxisRotatePoint on duals.  w: three duals, p: three duals or arrays."""
    p = [Dual.lift(c) for c in p]
    t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
    cx = w[1] * p[2] - w[2] * p[1]
    cy = w[2] * p[0] - w[0] * p[2]
    cz = w[0] * p[1] - w[1] * p[0]
    if float(t2.a[0]) > EPS:
        t = t2.sqrt()
        c, s = t.cos(), t.sin()
        k1 = s / t
        k2 = (1.0 - c) / t2
        dot = w[0] * p[0] + w[1] * p[1] + w[2] * p[2]
        return [p[0] * c + cx * k1 + w[0] * dot * k2,
                p[1] * c + cy * k1 + w[1] * dot * k2,
                p[2] * c + cz * k1 + w[2] * dot * k2]
    return [p[0] + cx, p[1] + cy, p[2] + cz]        # first-order branch: keeps the derivative finite at w = 0


def residuals_of_kind(kind, c, x):
    """c: (n, 9) constructor arguments in the reference's order; returns a list of dim duals of n entries each."""
    X = pose_duals(x)
    w, t = X[:3], X[3:]
    if kind == K_3DPD:                                            # costfunctions.h:39-54
        M = rotate(w, [c[:, 0], c[:, 1], c[:, 2]])
        M = [M[i] + (t[i] - c[:, 6 + i]) for i in range(3)]
        return [M[0] * c[:, 3] + M[1] * c[:, 4] + M[2] * c[:, 5]]
    if kind == K_3D3D:                                            # costfunctions.h:76-87
        M = rotate(w, [c[:, 0], c[:, 1], c[:, 2]])
        return [M[i] + t[i] - c[:, 3 + i] for i in range(3)]
    if kind == K_3D2D:                                            # costfunctions.h:111-126
        M = rotate(w, [c[:, 0], c[:, 1], c[:, 2]])
        M = [M[i] + t[i] + c[:, 5 + i] for i in range(3)]
        return [M[0] - M[2] * c[:, 3], M[1] - M[2] * c[:, 4]]
    if kind == K_2D3D:                                            # costfunctions.h:151-168
        M = rotate([-w[0], -w[1], -w[2]], [c[:, i] - t[i] for i in range(3)])
        M = [M[i] + c[:, 5 + i] for i in range(3)]
        return [M[0] - M[2] * c[:, 3], M[1] - M[2] * c[:, 4]]
    if kind == K_2D2D:                                            # costfunctions.h:192-216
        n = len(c)
        M = rotate(w, [c[:, 0], c[:, 1], np.ones(n)])
        Rt = rotate(w, [c[:, 4], c[:, 5], c[:, 6]])
        tt = [t[i] + c[:, 4 + i] - Rt[i] for i in range(3)]
        tn = (tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]).sqrt()
        tt = [q / tn for q in tt]
        sx, sy = c[:, 2], c[:, 3]
        return [M[0] * (tt[1] - tt[2] * sy) + M[1] * (tt[2] * sx - tt[0]) + M[2] * (tt[0] * sy - tt[1] * sx)]
    raise ValueError(kind)


def loss(loss_type, a, w, s):
    """(rho, rho') of CauchyLoss(a) / ArctanLoss(a), scaled by w; type 0 = trivial."""
    if loss_type == 1:
        b = a * a
        rho, d1 = b * np.log1p(s / b), np.maximum(DBL_MIN, 1.0 / (1.0 + s / b))
    elif loss_type == 2:
        rho, d1 = a * np.arctan2(s, a), np.maximum(DBL_MIN, 1.0 / (1.0 + s * s / (a * a)))
    else:
        rho, d1 = s, np.ones_like(s)
    return w * rho, w * d1


def evaluate(blocks, x):
    """One evaluation (B1 'init'): cost = sum rho/2 and the robustified rows r, J in BLOCK ORDER."""
    n = len(blocks)
    dims = np.array([DIMS[int(k)] for k in blocks["kind"]], dtype=np.int64)
    first = np.concatenate([[0], np.cumsum(dims)])
    r = np.zeros(first[-1])
    J = np.zeros((first[-1], 6))
    cost_terms = np.zeros(n)
    for kind in np.unique(blocks["kind"]):
        sel = np.nonzero(blocks["kind"] == kind)[0]
        b = blocks[sel]
        res = residuals_of_kind(int(kind), b["c"], x)
        vals = np.stack([np.broadcast_to(q.a, (len(sel),)) for q in res], axis=1)            # (n_k, dim)
        jac = np.stack([np.broadcast_to(q.v, (len(sel), 6)) for q in res], axis=1)           # (n_k, dim, 6)
        s = np.sum(vals * vals, axis=1)
        rho = np.zeros(len(sel))
        d1 = np.zeros(len(sel))
        for lt in np.unique(b["loss_type"]):
            m = b["loss_type"] == lt
            rho[m], d1[m] = loss(int(lt), b["loss_a"][m], b["loss_w"][m], s[m])
        cost_terms[sel] = 0.5 * rho
        sq = np.sqrt(d1)                                          # rho'' <= 0: corrector = scale rows by sqrt(rho')
        for q in range(DIMS[int(kind)]):
            r[first[sel] + q] = vals[:, q] * sq
            J[first[sel] + q] = jac[:, q, :] * sq[:, None]
    return float(np.sum(cost_terms)), r, J


def solve(blocks, x0, *, max_iterations=50, radius0=1e4, eta=1e-3, f_tol=1e-6, g_tol=1e-10, p_tol=1e-8,
          d_lo=1e-6, d_hi=1e32, radius_min=1e-32, radius_max=1e16, max_invalid=5):
    """SURVEY.md Appendix B1.  Returns (x, label, trace); trace = list of dicts, one per iteration."""
    x = np.array(x0, dtype=np.float64)
    cost, r, J = evaluate(blocks, x)
    evaluations = 1
    trace = []
    g = J.T @ r
    if np.max(np.abs(g)) <= g_tol:
        return x, "gradient", trace, evaluations
    c = 1.0 / (1.0 + np.sqrt(np.sum(J * J, axis=0)))              # Jacobi scaling, ONCE
    mu, nu, bad = radius0, 2.0, 0
    d = None
    k = 0
    while True:
        k += 1
        if k > max_iterations:
            return x, "max_iterations", trace, evaluations
        if mu < radius_min:
            return x, "radius", trace, evaluations
        Js = J * c
        if d is None:
            d = np.clip(np.sum(Js * Js, axis=0), d_lo, d_hi)
        A = np.vstack([Js, np.diag(np.sqrt(d / mu))])
        b = np.concatenate([-r, np.zeros(6)])
        step, *_ = np.linalg.lstsq(A, b, rcond=None)
        Jd = Js @ step
        model = -float(np.dot(Jd, r + 0.5 * Jd))
        row = dict(iteration=k, cost=cost, radius=mu, model_change=model, gradient_max=float(np.max(np.abs(J.T @ r))))
        if not (np.all(np.isfinite(step)) and model > 0.0):
            bad += 1
            row.update(status="invalid")
            trace.append(row)
            if bad >= max_invalid:
                return x, "failure", trace, evaluations
            mu, nu = mu / nu, 2.0 * nu
            continue
        bad = 0
        delta = step * c
        x_c = x + delta
        cost_c, r_c, J_c = evaluate(blocks, x_c)
        evaluations += 1
        row.update(candidate_cost=cost_c, step_norm=float(np.linalg.norm(delta)))
        if np.linalg.norm(delta) <= p_tol * (np.linalg.norm(x) + p_tol):
            row.update(status="parameter")
            trace.append(row)
            return x, "parameter", trace, evaluations
        q = (cost - cost_c) / model
        row.update(relative_decrease=q)
        if abs(cost - cost_c) <= f_tol * cost:
            row.update(status="function")
            trace.append(row)
            return x, "function", trace, evaluations
        if q > eta:
            x, cost, r, J = x_c, cost_c, r_c, J_c
            row.update(status="accepted")
            trace.append(row)
            if np.max(np.abs(J.T @ r)) <= g_tol:
                row.update(status="gradient")
                return x, "gradient", trace, evaluations
            mu = min(radius_max, mu / max(1.0 / 3.0, 1.0 - (2.0 * q - 1.0) ** 3))
            nu = 2.0
            d = None
        else:
            row.update(status="rejected")
            trace.append(row)
            mu, nu = mu / nu, 2.0 * nu
