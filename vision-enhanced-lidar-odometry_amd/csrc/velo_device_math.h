// velo_device_math.h -- device-side arithmetic of the residual path (gfx950 only).
//
// Rows R1-R5, L1 of SURVEY.md section 8(a): the five residual functors of the reference
// (costfunctions.h:17-220) with the Jacobians ceres::AutoDiffCostFunction<F,k,6> [3P] produces, and the
// Ceres loss functions the reference attaches at velo.h:688,714-717,748-751,781-784,887-890.
//
// Forward-mode dual numbers as in ceres::Jet, with one economy: every quantity that depends on the pose only
// through the rotation (theta, sin, cos, unit axis, the rotated point) carries 3 partials (d/d omega); the
// translation partials are attached when the residual is assembled.  A Jet<6> evaluation produces exactly the
// same numbers -- its omitted partials are the zeros and ones that drop out here.
#pragma once
#include <hip/hip_runtime.h>

namespace velo {

// ---- sin / cos of the rotation angle, PINNED -------------------------------------------------------------------------------------
// The reference takes sin and cos of theta = |omega| from whatever libm it is linked against ([3P], unpinned: ceres::AngleAxisRotatePoint
// calls sin() / cos()), and the float coordinates of the transformed queries (utility.h:97-103) depend on their last
// bit.  To let the DEVICE compute the pose scalars of an association round -- so that a whole frame_to_frame runs without a
// host round trip per round -- while the tables stay bit-identical to the CPU restatement, both sides use THIS function: the
// fdlibm kernels (__kernel_sin / __kernel_cos polynomials, Cody-Waite reduction by pi/2 in three parts), written with plain IEEE
// double operations in a fixed order (no FMA: the file is built with -ffp-contract=off, the oracle too), so host, device and
// oracle produce the same bits.  Accuracy ~1 ulp for |x| < 1e5 -- the accuracy class of the libm calls it replaces.
__host__ __device__ inline void velo_sincos(double x, double* s, double* c) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_2 = 6.07710050630396597660e-11,
                 pio2_3 = 2.02226624871116645580e-21;
    if (!(x == x) || x - x != 0.0) { *s = x - x; *c = x - x; return; }     // NaN / infinity -> NaN
    double r = x;
    long long n = 0;
    if (x > 0.78539816339744830962 || x < -0.78539816339744830962) {
        const double fn = floor(x * invpio2 + 0.5);
        n = (long long)fn;
        r = ((x - fn * pio2_1) - fn * pio2_2) - fn * pio2_3;
    }
    const double z = r * r;
    const double ks = r + (z * r) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double kc = 1.0 - (0.5 * z - z * (z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))));
    switch ((int)(n & 3)) {
        case 0: *s = ks; *c = kc; break;
        case 1: *s = kc; *c = -ks; break;
        case 2: *s = -ks; *c = -kc; break;
        default: *s = -kc; *c = ks; break;
    }
}

template <int N>
struct Dual {
    double a;
    double v[N];
};
typedef Dual<3> D3;
typedef Dual<6> D6;

template <int N> __device__ __forceinline__ Dual<N> dconst(double s) {
    Dual<N> h; h.a = s;
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = 0.0;
    return h;
}
template <int N> __device__ __forceinline__ Dual<N> dvar(double s, int k) {
    Dual<N> h; h.a = s;
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = (i == k) ? 1.0 : 0.0;
    return h;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h; h.a = f.a + g.a;
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = f.v[i] + g.v[i];
    return h;
}
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h; h.a = f.a - g.a;
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = f.v[i] - g.v[i];
    return h;
}
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& f) {
    Dual<N> h; h.a = -f.a;
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = -f.v[i];
    return h;
}
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h; h.a = f.a * g.a;
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
    return h;
}
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h; const double gi = 1.0 / g.a; const double q = f.a * gi; h.a = q;
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = (f.v[i] - q * g.v[i]) * gi;
    return h;
}
// dual (op) constant: what Jet arithmetic yields when the other operand has zero partials
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& f, double s) {
    Dual<N> h; h.a = f.a * s;
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = f.v[i] * s;
    return h;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& f, double s) { Dual<N> h = f; h.a = f.a + s; return h; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& f, double s) { Dual<N> h = f; h.a = f.a - s; return h; }
template <int N> __device__ __forceinline__ Dual<N> dsqrt(const Dual<N>& f) {
    Dual<N> h; h.a = sqrt(f.a); const double d = 1.0 / (2.0 * h.a);
#pragma unroll
    for (int i = 0; i < N; i++) h.v[i] = f.v[i] * d;
    return h;
}

// ---- the point-independent part of ceres::AngleAxisRotatePoint [3P rotation.h] ------------------
// Rodrigues for theta^2 > DBL_EPSILON, first order p + w x p otherwise.  Values and d/d omega.
struct PoseRot {
    bool small;
    double w[3];   // omega (the first-order branch differentiates to the unit vectors)
    D3 c, s;       // cos(theta), sin(theta)
    D3 u[3];       // omega / theta
    D3 omc;        // 1 - cos(theta)
};

__device__ __forceinline__ void pose_rot_init(const double w[3], PoseRot* R) {
    R->w[0] = w[0]; R->w[1] = w[1]; R->w[2] = w[2];
    const D3 w0 = dvar<3>(w[0], 0), w1 = dvar<3>(w[1], 1), w2 = dvar<3>(w[2], 2);
    const D3 theta2 = w0 * w0 + w1 * w1 + w2 * w2;
    R->small = !(theta2.a > 2.220446049250313e-16);
    if (!R->small) {
        const D3 theta = dsqrt(theta2);
        double sv, cv;
        velo_sincos(theta.a, &sv, &cv);
        R->s.a = sv; R->c.a = cv;
#pragma unroll
        for (int i = 0; i < 3; i++) { R->s.v[i] = cv * theta.v[i]; R->c.v[i] = -sv * theta.v[i]; }
        const D3 ti = dconst<3>(1.0) / theta;
        R->u[0] = w0 * ti; R->u[1] = w1 * ti; R->u[2] = w2 * ti;
        R->omc = dconst<3>(1.0) - R->c;
    }
}

// R(omega) p for a constant point p: value + d/d omega
__device__ __forceinline__ void rotate_point(const PoseRot& R, const double p[3], D3 out[3]) {
    if (!R.small) {
        const D3 uxp0 = R.u[1] * p[2] - R.u[2] * p[1];
        const D3 uxp1 = R.u[2] * p[0] - R.u[0] * p[2];
        const D3 uxp2 = R.u[0] * p[1] - R.u[1] * p[0];
        const D3 tmp = (R.u[0] * p[0] + R.u[1] * p[1] + R.u[2] * p[2]) * R.omc;
        out[0] = R.c * p[0] + uxp0 * R.s + R.u[0] * tmp;
        out[1] = R.c * p[1] + uxp1 * R.s + R.u[1] * tmp;
        out[2] = R.c * p[2] + uxp2 * R.s + R.u[2] * tmp;
    } else {
        // p + w x p with w the independent variables: d(w x p)/dw_j = e_j x p
        out[0].a = p[0] + (R.w[1] * p[2] - R.w[2] * p[1]);
        out[1].a = p[1] + (R.w[2] * p[0] - R.w[0] * p[2]);
        out[2].a = p[2] + (R.w[0] * p[1] - R.w[1] * p[0]);
        out[0].v[0] = 0.0;   out[0].v[1] = p[2];  out[0].v[2] = -p[1];
        out[1].v[0] = -p[2]; out[1].v[1] = 0.0;   out[1].v[2] = p[0];
        out[2].v[0] = p[1];  out[2].v[1] = -p[0]; out[2].v[2] = 0.0;
    }
}
// value only (used for d/dt of cost2D3D, where the rotated point itself carries the translation partials)
__device__ __forceinline__ void rotate_point_value(const PoseRot& R, const double p[3], double out[3]) {
    if (!R.small) {
        const double c0 = R.u[1].a * p[2] - R.u[2].a * p[1], c1 = R.u[2].a * p[0] - R.u[0].a * p[2], c2 = R.u[0].a * p[1] - R.u[1].a * p[0];
        const double tmp = (R.u[0].a * p[0] + R.u[1].a * p[1] + R.u[2].a * p[2]) * R.omc.a;
        out[0] = p[0] * R.c.a + c0 * R.s.a + R.u[0].a * tmp;
        out[1] = p[1] * R.c.a + c1 * R.s.a + R.u[1].a * tmp;
        out[2] = p[2] * R.c.a + c2 * R.s.a + R.u[2].a * tmp;
    } else {
        out[0] = p[0] + (R.w[1] * p[2] - R.w[2] * p[1]);
        out[1] = p[1] + (R.w[2] * p[0] - R.w[0] * p[2]);
        out[2] = p[2] + (R.w[0] * p[1] - R.w[1] * p[0]);
    }
}

// widen a rotation-only dual to the 6 pose parameters (translation partials zero)
__device__ __forceinline__ D6 widen(const D3& f) {
    D6 h; h.a = f.a;
    h.v[0] = f.v[0]; h.v[1] = f.v[1]; h.v[2] = f.v[2]; h.v[3] = 0.0; h.v[4] = 0.0; h.v[5] = 0.0;
    return h;
}

// ---- residual functors: value r[k] and row-major Jacobian J[k][6] ------------------------------------------------
// R1 cost3DPD (costfunctions.h:39-54): r = N . (R p + t - v0)
__device__ __forceinline__ void res_3dpd(const PoseRot& R, const double t[3], const double p[3], const double n[3], const double v0[3],
                                         double* r, double J[6]) {
    D3 m[3];
    rotate_point(R, p, m);
    const double m0 = m[0].a + (t[0] - v0[0]), m1 = m[1].a + (t[1] - v0[1]), m2 = m[2].a + (t[2] - v0[2]);
    *r = m0 * n[0] + m1 * n[1] + m2 * n[2];
#pragma unroll
    for (int j = 0; j < 3; j++) J[j] = m[0].v[j] * n[0] + m[1].v[j] * n[1] + m[2].v[j] * n[2];
    J[3] = n[0]; J[4] = n[1]; J[5] = n[2];
}
// R4 cost3D3D (costfunctions.h:76-87): r = R m + t - s
__device__ __forceinline__ void res_3d3d(const PoseRot& R, const double t[3], const double mm[3], const double s[3], double r[3], double J[18]) {
    D3 m[3];
    rotate_point(R, mm, m);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r[k] = m[k].a + t[k] - s[k];
#pragma unroll
        for (int j = 0; j < 3; j++) { J[k * 6 + j] = m[k].v[j]; J[k * 6 + 3 + j] = (j == k) ? 1.0 : 0.0; }
    }
}
// R2 cost3D2D (costfunctions.h:111-126): M = R m + t + t_cam; r = (Mx - sx Mz, My - sy Mz)
__device__ __forceinline__ void res_3d2d(const PoseRot& R, const double t[3], const double mm[3], const double s[2], const double tc[3],
                                         double r[2], double J[12]) {
    D3 m[3];
    rotate_point(R, mm, m);
    const double M0 = m[0].a + (t[0] + tc[0]), M1 = m[1].a + (t[1] + tc[1]), M2 = m[2].a + (t[2] + tc[2]);
    r[0] = M0 - s[0] * M2;
    r[1] = M1 - s[1] * M2;
#pragma unroll
    for (int j = 0; j < 3; j++) { J[j] = m[0].v[j] - s[0] * m[2].v[j]; J[6 + j] = m[1].v[j] - s[1] * m[2].v[j]; }
    J[3] = 1.0; J[4] = 0.0; J[5] = -s[0];
    J[9] = 0.0; J[10] = 1.0; J[11] = -s[1];
}
// R3 cost2D3D (costfunctions.h:151-168): M = R(-omega)(m - t) + t_cam.  Rinv is PoseRot of -omega, whose
// partials are w.r.t. (-omega): d/d omega flips their sign.  d/dt_k = R(-omega)(-e_k), the Jet propagation of the
// point's own partial through the same expression.
__device__ __forceinline__ void res_2d3d(const PoseRot& Rinv, const double t[3], const double mm[3], const double s[2], const double tc[3],
                                         double r[2], double J[12]) {
    const double q[3] = {mm[0] - t[0], mm[1] - t[1], mm[2] - t[2]};
    D3 m[3];
    rotate_point(Rinv, q, m);
    const double M0 = m[0].a + tc[0], M1 = m[1].a + tc[1], M2 = m[2].a + tc[2];
    r[0] = M0 - s[0] * M2;
    r[1] = M1 - s[1] * M2;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const double d0 = -m[0].v[j], d1 = -m[1].v[j], d2 = -m[2].v[j];
        J[j] = d0 - s[0] * d2;
        J[6 + j] = d1 - s[1] * d2;
    }
    // d/dt_k = R(-omega)(-e_k): minus the k-th column of R(-omega), each from the value-only rotation of a unit vector
    {
        const double e0[3] = {-1.0, 0.0, 0.0}, e1[3] = {0.0, -1.0, 0.0}, e2[3] = {0.0, 0.0, -1.0};
        double c0[3], c1[3], c2[3];
        rotate_point_value(Rinv, e0, c0);
        rotate_point_value(Rinv, e1, c1);
        rotate_point_value(Rinv, e2, c2);
        J[3] = c0[0] - s[0] * c0[2]; J[9] = c0[1] - s[1] * c0[2];
        J[4] = c1[0] - s[0] * c1[2]; J[10] = c1[1] - s[1] * c1[2];
        J[5] = c2[0] - s[0] * c2[2]; J[11] = c2[1] - s[1] * c2[2];
    }
}
// R5 cost2D2D (costfunctions.h:192-216): epipolar residual, 6-wide duals after the two rotations
__device__ __forceinline__ void res_2d2d(const PoseRot& R, const double t[3], const double mm[2], const double s[2], const double tc[3],
                                         double r[1], double J[6]) {
    const double p[3] = {mm[0], mm[1], 1.0};
    D3 m3[3], rt3[3];
    rotate_point(R, p, m3);
    rotate_point(R, tc, rt3);
    const D6 m0 = widen(m3[0]), m1 = widen(m3[1]), m2 = widen(m3[2]);
    D6 tx = -widen(rt3[0]) + dvar<6>(t[0], 3) + tc[0];
    D6 ty = -widen(rt3[1]) + dvar<6>(t[1], 4) + tc[1];
    D6 tz = -widen(rt3[2]) + dvar<6>(t[2], 5) + tc[2];
    const D6 tn = dsqrt(tx * tx + ty * ty + tz * tz);
    tx = tx / tn;
    ty = ty / tn;
    tz = tz / tn;
    const double sx = s[0], sy = s[1];
    const D6 res = m0 * (tz * (-sy) + ty) + m1 * (tz * sx - tx) + m2 * (ty * (-sx) + tx * sy);
    r[0] = res.a;
#pragma unroll
    for (int j = 0; j < 6; j++) J[j] = res.v[j];
}

// The same residual with the pose parameters J0 .. J0 + W - 1 only: every partial of a dual number is computed from the values and
// the SAME partial of its operands, so a window of the six gives the bits the 6-wide evaluation gives for those columns (and the same
// value).  Three passes of width 2 need a third of the registers of one pass of width 6 -- what lets the visual sweep ride in the lean
// sweep + step kernel (velo_kernels.h, visual_sweep_one).
template <int W, int J0> __device__ __forceinline__ Dual<W> widen_win(const D3& f) {
    Dual<W> h; h.a = f.a;
#pragma unroll
    for (int i = 0; i < W; i++) h.v[i] = (J0 + i < 3) ? f.v[(J0 + i < 3) ? J0 + i : 0] : 0.0;
    return h;
}
template <int W, int J0> __device__ __forceinline__ Dual<W> dvar_win(double s, int param) {
    Dual<W> h; h.a = s;
#pragma unroll
    for (int i = 0; i < W; i++) h.v[i] = (J0 + i == param) ? 1.0 : 0.0;
    return h;
}
template <int W, int J0>
__device__ __forceinline__ void res_2d2d_win(const PoseRot& R, const double t[3], const double mm[2], const double s[2], const double tc[3],
                                             double* r, double* Jw /* [W] */) {
    typedef Dual<W> DW;
    const double p[3] = {mm[0], mm[1], 1.0};
    D3 m3[3], rt3[3];
    rotate_point(R, p, m3);
    rotate_point(R, tc, rt3);
    const DW m0 = widen_win<W, J0>(m3[0]), m1 = widen_win<W, J0>(m3[1]), m2 = widen_win<W, J0>(m3[2]);
    DW tx = -widen_win<W, J0>(rt3[0]) + dvar_win<W, J0>(t[0], 3) + tc[0];
    DW ty = -widen_win<W, J0>(rt3[1]) + dvar_win<W, J0>(t[1], 4) + tc[1];
    DW tz = -widen_win<W, J0>(rt3[2]) + dvar_win<W, J0>(t[2], 5) + tc[2];
    const DW tn = dsqrt(tx * tx + ty * ty + tz * tz);
    tx = tx / tn;
    ty = ty / tn;
    tz = tz / tn;
    const double sx = s[0], sy = s[1];
    const DW res = m0 * (tz * (-sy) + ty) + m1 * (tz * sx - tx) + m2 * (ty * (-sx) + tx * sy);
    *r = res.a;
#pragma unroll
    for (int j = 0; j < W; j++) Jw[j] = res.v[j];
}

// ---- loss functions [3P loss_function.cc] ---------------------------------------------------
// Cauchy(a) scaled by w: rho = w b ln(1 + s/b), rho' = w / (1 + s/b), b = a^2
__device__ __forceinline__ void loss_cauchy(double a, double w, double s, double* rho0, double* rho1) {
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + s * c, inv = 1.0 / sum;
    *rho0 = w * (b * log(sum));
    *rho1 = w * fmax(2.2250738585072014e-308, inv);
}
// Arctan(a) scaled by w: rho = w a atan2(s, a), rho' = w / (1 + s^2/a^2)
__device__ __forceinline__ void loss_arctan(double a, double w, double s, double* rho0, double* rho1) {
    const double b = 1.0 / (a * a);
    const double sum = 1.0 + s * s * b, inv = 1.0 / sum;
    *rho0 = w * (a * atan2(s, a));
    *rho1 = w * fmax(2.2250738585072014e-308, inv);
}

}  // namespace velo
