// velo_tri_kernels.h -- SURVEY.md 8(f) row 4: batched triangulatePoint (reference velo.h:1027-1130, functors
// costfunctions.h:288-375).  Included by velo_hip.hip after velo_kernels.h (LMParams, loss functions); gfx950 only.
//
// Every landmark is an independent 3-unknown Levenberg-Marquardt problem over a handful of observations (3..30), thousands per
// frame.  Two kernels with identical results: one WAVE per landmark (default, see triangulate_wave_kernel below) and one thread
// per landmark (the plain restatement, kept for A/B).  The unknown is the POINT and the camera poses are constants, so each
// residual is affine in it:
//     M = R(-omega_f) (x - c_f) [+ t_cam],   3-D: r = M - s                 d r/dx = R
//                                             2-D: r = (M.x - s.x M.z, M.y - s.y M.z)   d r/dx = R[0,:] - s.x R[2,:], ...
// R(-omega_f) and the Rodrigues scalars of every frame are computed once on the host in double (the same libm the CPU
// restatement uses); the value M still goes through the Rodrigues form so that it rounds like ceres::AngleAxisRotatePoint.
// The columns of R are that same form applied to the unit vectors -- exactly what the dual-number evaluation yields.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/velo_hip.h"
#include "velo_device_math.h"

namespace velo {

struct TriFrame {
    double w[3];        // -omega of the frame's pose (rot of costfunctions.h:318-320,361-363)
    double u[3];        // w / theta
    double c, s, omc;   // cos, sin, 1 - cos of theta
    double R[9];        // row-major R(w): column j = rotation of e_j
    double center[3];   // camera_poses[f][3..5]
    int small;          // theta^2 <= DBL_EPSILON: first-order branch
    int pad;
};

struct TriParams { LMParams lm; double loss_a, loss_w; };

__device__ __forceinline__ void tri_rotate(const TriFrame& F, const double p[3], double out[3]) {
    if (!F.small) {
        const double c0 = F.u[1] * p[2] - F.u[2] * p[1], c1 = F.u[2] * p[0] - F.u[0] * p[2], c2 = F.u[0] * p[1] - F.u[1] * p[0];
        const double tmp = (F.u[0] * p[0] + F.u[1] * p[1] + F.u[2] * p[2]) * F.omc;
        out[0] = p[0] * F.c + c0 * F.s + F.u[0] * tmp;
        out[1] = p[1] * F.c + c1 * F.s + F.u[1] * tmp;
        out[2] = p[2] * F.c + c2 * F.s + F.u[2] * tmp;
    } else {
        out[0] = p[0] + (F.w[1] * p[2] - F.w[2] * p[1]);
        out[1] = p[1] + (F.w[2] * p[0] - F.w[0] * p[2]);
        out[2] = p[2] + (F.w[0] * p[1] - F.w[1] * p[0]);
    }
}

struct TriEval { double cost; double H[9]; double g[3]; };

// residual blocks in the reference's order: 3-D observations first (velo.h:1049-1085), then 2-D (velo.h:1087-1122);
// first_3d_only = the problem of the early solve at velo.h:1080-1083
__device__ __forceinline__ void tri_evaluate(const TriFrame* __restrict__ frames, const double* __restrict__ cam_t, const velo_tri_obs* __restrict__ obs,
                                             int n_obs, bool first_3d_only, const TriParams& P, const double x[3], TriEval* E) {
    E->cost = 0.0;
#pragma unroll
    for (int i = 0; i < 9; i++) E->H[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) E->g[i] = 0.0;
    for (int pass = 0; pass < 2; pass++) {
        for (int k = 0; k < n_obs; k++) {
            const velo_tri_obs o = obs[k];
            if ((o.kind == VELO_TRI_OBS_2D) != (pass == 1)) continue;
            const TriFrame& F = frames[o.frame];
            const double m0[3] = {x[0] - F.center[0], x[1] - F.center[1], x[2] - F.center[2]};
            double m[3], r[3], J[3][3];
            tri_rotate(F, m0, m);
            int d;
            double rho0, rho1;
            if (pass == 0) {
                d = 3;
#pragma unroll
                for (int q = 0; q < 3; q++) { r[q] = m[q] - (double)o.s[q]; J[q][0] = F.R[q * 3]; J[q][1] = F.R[q * 3 + 1]; J[q][2] = F.R[q * 3 + 2]; }
                const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                rho0 = sq; rho1 = 1.0;                               // TrivialLoss
            } else {
                d = 2;
                m[0] += cam_t[3 * o.cam]; m[1] += cam_t[3 * o.cam + 1]; m[2] += cam_t[3 * o.cam + 2];
                const double sx = (double)o.s[0], sy = (double)o.s[1];
                r[0] = m[0] - sx * m[2];
                r[1] = m[1] - sy * m[2];
                r[2] = 0.0;
#pragma unroll
                for (int i = 0; i < 3; i++) { J[0][i] = F.R[i] - sx * F.R[6 + i]; J[1][i] = F.R[3 + i] - sy * F.R[6 + i]; J[2][i] = 0.0; }
                const double sq = r[0] * r[0] + r[1] * r[1];
                loss_cauchy(P.loss_a, P.loss_w, sq, &rho0, &rho1);
            }
            E->cost += 0.5 * rho0;
            const double sr = sqrt(rho1);
            for (int q = 0; q < d; q++) {
                const double rk = r[q] * sr;
                const double Jk[3] = {J[q][0] * sr, J[q][1] * sr, J[q][2] * sr};
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    E->g[i] += Jk[i] * rk;
#pragma unroll
                    for (int j = 0; j < 3; j++) E->H[i * 3 + j] += Jk[i] * Jk[j];
                }
            }
            if (first_3d_only) return;
        }
        if (first_3d_only) return;
    }
}

__device__ __forceinline__ bool tri_chol3(const double A[9], const double b[3], double y[3]) {
    double L[9];
#pragma unroll
    for (int i = 0; i < 9; i++) L[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double d = A[j * 3 + j];
        for (int k = 0; k < j; k++) d -= L[j * 3 + k] * L[j * 3 + k];
        if (!(d > 0.0) || !isfinite(d)) return false;
        L[j * 3 + j] = sqrt(d);
        for (int i = j + 1; i < 3; i++) {
            double s = A[i * 3 + j];
            for (int k = 0; k < j; k++) s -= L[i * 3 + k] * L[j * 3 + k];
            L[i * 3 + j] = s / L[j * 3 + j];
        }
    }
    double z[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * 3 + k] * z[k]; z[i] = s / L[i * 3 + i]; }
#pragma unroll
    for (int i = 2; i >= 0; i--) { double s = z[i]; for (int k = i + 1; k < 3; k++) s -= L[k * 3 + i] * y[k]; y[i] = s / L[i * 3 + i]; }
    return isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2]);
}

// Ceres' trust-region LM with default options, 3 unknowns -- the loop of the main path's lm_step.
// `eval(x, &E)` fills cost, J^T J, J^T r at x; with a wave-wide evaluation every lane runs this bookkeeping identically.
template <typename Eval>
__device__ __forceinline__ void tri_solve(Eval&& eval, const TriParams& P, double x[3], velo_tri_result* S) {
    const LMParams& Q = P.lm;
    TriEval E;
    eval(x, &E);
    S->evaluations++;
    S->n_solves++;
    S->lm_iterations = 0;
    double cost = E.cost;
    S->final_cost = cost;
    double x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (fmax(fmax(fabs(E.g[0]), fabs(E.g[1])), fabs(E.g[2])) <= Q.gradient_tolerance) { S->termination = VELO_CONVERGENCE; return; }
    double scale[3], diag[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 3; j++) scale[j] = 1.0 / (1.0 + sqrt(E.H[j * 3 + j]));
    double radius = Q.initial_radius, decrease = 2.0;
    bool reuse_diag = false;
    int invalid = 0;
    S->termination = VELO_NO_CONVERGENCE;
    for (int it = 1;; it++) {
        if (it > Q.max_num_iterations) { S->termination = VELO_NO_CONVERGENCE; break; }
        if (radius < Q.min_radius) { S->termination = VELO_CONVERGENCE; break; }
        S->lm_iterations = it;
        double Hs[9], gs[3], A[9];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            gs[i] = E.g[i] * scale[i];
#pragma unroll
            for (int j = 0; j < 3; j++) Hs[i * 3 + j] = E.H[i * 3 + j] * scale[i] * scale[j];
        }
        if (!reuse_diag) {
#pragma unroll
            for (int j = 0; j < 3; j++) diag[j] = fmin(fmax(Hs[j * 3 + j], Q.min_diag), Q.max_diag);
        }
#pragma unroll
        for (int i = 0; i < 9; i++) A[i] = Hs[i];
#pragma unroll
        for (int j = 0; j < 3; j++) { const double l = sqrt(diag[j] / radius); A[j * 3 + j] += l * l; }
        double y[3], step[3] = {0.0, 0.0, 0.0};
        bool ok = tri_chol3(A, gs, y);
        reuse_diag = true;
        double model_change = 0.0;
        if (ok) {
            double gd = 0.0, dHd = 0.0;
#pragma unroll
            for (int i = 0; i < 3; i++) step[i] = -y[i];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                gd += gs[i] * step[i];
#pragma unroll
                for (int j = 0; j < 3; j++) dHd += step[i] * Hs[i * 3 + j] * step[j];
            }
            model_change = -(gd + 0.5 * dHd);
            if (!(model_change > 0.0)) ok = false;
        }
        if (!ok) {
            if (++invalid >= Q.max_invalid) { S->termination = VELO_FAILURE; break; }
            radius = radius / decrease; decrease *= 2.0; reuse_diag = true;
            continue;
        }
        invalid = 0;
        double xc[3], dn = 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) { const double d = step[i] * scale[i]; xc[i] = x[i] + d; dn += d * d; }
        dn = sqrt(dn);
        TriEval Ec;
        eval(xc, &Ec);
        S->evaluations++;
        if (dn <= Q.parameter_tolerance * (x_norm + Q.parameter_tolerance)) { S->termination = VELO_CONVERGENCE; break; }
        const double cost_change = cost - Ec.cost;
        if (fabs(cost_change) <= Q.function_tolerance * cost) { S->termination = VELO_CONVERGENCE; break; }
        const double q = cost_change / model_change;
        if (q > Q.min_relative_decrease) {
            x[0] = xc[0]; x[1] = xc[1]; x[2] = xc[2];
            cost = Ec.cost; E = Ec;
            x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            if (fmax(fmax(fabs(E.g[0]), fabs(E.g[1])), fabs(E.g[2])) <= Q.gradient_tolerance) { S->termination = VELO_CONVERGENCE; break; }
            const double t = 2.0 * q - 1.0;
            radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            radius = fmin(Q.max_radius, radius);
            decrease = 2.0; reuse_diag = false;
        } else {
            radius = radius / decrease; decrease *= 2.0; reuse_diag = true;
        }
    }
    S->final_cost = cost;
}

__global__ void __launch_bounds__(64)
triangulate_kernel(const TriFrame* __restrict__ frames, const double* __restrict__ cam_t, const velo_tri_obs* __restrict__ obs,
                   const int* __restrict__ off, int n, TriParams P, float* __restrict__ pts, const unsigned char* __restrict__ init,
                   velo_tri_result* __restrict__ results)
#if VELO_DEF_LOAD
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= n) return;
    const int b = off[l], n_obs = off[l + 1] - b;
    const bool guess = init != nullptr && init[l] != 0;
    double x[3] = {0.0, 0.0, 10.0};                                   // velo.h:1043
    if (guess) { x[0] = pts[3 * l]; x[1] = pts[3 * l + 1]; x[2] = pts[3 * l + 2]; }   // velo.h:1044-1049
    velo_tri_result S;
    S.n_solves = 0; S.termination = VELO_CONVERGENCE; S.lm_iterations = 0; S.evaluations = 0; S.final_cost = 0.0;
    bool any3d = false;
    for (int k = 0; k < n_obs; k++) any3d = any3d || obs[b + k].kind == VELO_TRI_OBS_3D;
    if (!guess && any3d)                                              // velo.h:1080-1083
        tri_solve([&](const double* xx, TriEval* E) { tri_evaluate(frames, cam_t, obs + b, n_obs, true, P, xx, E); }, P, x, &S);
    if (n_obs > 0)                                                    // velo.h:1123
        tri_solve([&](const double* xx, TriEval* E) { tri_evaluate(frames, cam_t, obs + b, n_obs, false, P, xx, E); }, P, x, &S);
    pts[3 * l] = (float)x[0]; pts[3 * l + 1] = (float)x[1]; pts[3 * l + 2] = (float)x[2];   // velo.h:1124-1126
    if (results) results[l] = S;
}
#else
;
#endif

// ---- one WAVE per landmark (default) ----------------------------------------------------------------------------------------
// The thread-per-landmark kernel above is a single dependent chain per landmark: every evaluation walks the observations one
// after the other, each paying two dependent global loads (observation -> its frame) and ~150 double operations, 50 evaluations
// deep for the hard landmarks -- 2.2 ms for the slowest of 3,000 landmarks while most lanes idle.  Here the 64 lanes of a wave
// take one observation each (registers keep its frame constants across all evaluations), write the robustified rows
// {J_q * sr, r_q * sr} into LDS in block order, and lanes 0..9 each own ONE accumulator (6 entries of J^T J, 3 of J^T r, the
// cost) which they sum over the rows IN BLOCK ORDER -- every accumulator sees exactly the additions of the sequential loop, so
// the result is bit-identical to it, only 10-way parallel and with the per-observation arithmetic 64-way parallel.  The LM step
// itself (3x3) is computed redundantly by all lanes (wave-uniform, no divergence).  Observations must arrive with the 3-D ones
// first (the host entry point partitions them, stable).
struct TriLane {         // one observation held in registers
    int kind;
    double s[3], t[3];
    TriFrame F;
};

__device__ __forceinline__ void tri_load_obs(const TriFrame* __restrict__ frames, const double* __restrict__ cam_t, const velo_tri_obs& o, TriLane* L) {
    L->kind = o.kind;
    L->s[0] = (double)o.s[0]; L->s[1] = (double)o.s[1]; L->s[2] = (double)o.s[2];
    L->F = frames[o.frame];
    if (o.kind == VELO_TRI_OBS_2D) { L->t[0] = cam_t[3 * o.cam]; L->t[1] = cam_t[3 * o.cam + 1]; L->t[2] = cam_t[3 * o.cam + 2]; }
    else { L->t[0] = 0.0; L->t[1] = 0.0; L->t[2] = 0.0; }
}

// rows of one block at x, robustified: V[q] = {J_q0, J_q1, J_q2, r_q} * sqrt(rho'), *half_rho = 0.5 rho; returns the block dimension
__device__ __forceinline__ int tri_block_rows(const TriLane& L, const TriParams& P, const double x[3], double V[3][4], double* half_rho) {
    const TriFrame& F = L.F;
    const double m0[3] = {x[0] - F.center[0], x[1] - F.center[1], x[2] - F.center[2]};
    double m[3], r[3], J[3][3];
    tri_rotate(F, m0, m);
    int d;
    double rho0, rho1;
    if (L.kind == VELO_TRI_OBS_3D) {
        d = 3;
#pragma unroll
        for (int q = 0; q < 3; q++) { r[q] = m[q] - L.s[q]; J[q][0] = F.R[q * 3]; J[q][1] = F.R[q * 3 + 1]; J[q][2] = F.R[q * 3 + 2]; }
        rho0 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2]; rho1 = 1.0;
    } else {
        d = 2;
        m[0] += L.t[0]; m[1] += L.t[1]; m[2] += L.t[2];
        r[0] = m[0] - L.s[0] * m[2];
        r[1] = m[1] - L.s[1] * m[2];
        r[2] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) { J[0][i] = F.R[i] - L.s[0] * F.R[6 + i]; J[1][i] = F.R[3 + i] - L.s[1] * F.R[6 + i]; J[2][i] = 0.0; }
        loss_cauchy(P.loss_a, P.loss_w, r[0] * r[0] + r[1] * r[1], &rho0, &rho1);
    }
    *half_rho = 0.5 * rho0;
    const double sr = sqrt(rho1);
#pragma unroll
    for (int q = 0; q < 3; q++) { V[q][0] = J[q][0] * sr; V[q][1] = J[q][1] * sr; V[q][2] = J[q][2] * sr; V[q][3] = r[q] * sr; }
    return d;
}

struct TriShared {
    double V[64 * 3][4];     // robustified rows of up to 64 blocks, dense: 3-D blocks (3 rows each) first, then 2-D blocks (2 rows each)
    double half_rho[64];
};

// n3d = number of 3-D observations of the landmark (they come first); rows of block p start at row_of(p)
__device__ __forceinline__ int tri_row_of(int p, int n3d) { return p < n3d ? 3 * p : 3 * n3d + 2 * (p - n3d); }

__device__ __forceinline__ void tri_evaluate_wave(TriShared& sh, const TriFrame* __restrict__ frames, const double* __restrict__ cam_t,
                                                  const velo_tri_obs* __restrict__ obs, int n_obs, int n3d, bool first_3d_only, const TriLane& mine,
                                                  const TriParams& P, const double x[3], TriEval* E) {
    const int lane = threadIdx.x;
    // accumulator owned by this lane: 0..5 = H(ia, ja) upper triangle, 6..8 = g(ia) (column 3 of a row holds the residual), 9 = cost
    const int ia = lane < 3 ? 0 : lane < 5 ? 1 : lane == 5 ? 2 : lane < 9 ? lane - 6 : 0;
    const int ja = lane < 3 ? lane : lane < 5 ? lane - 2 : lane == 5 ? 2 : 3;
    double acc = 0.0;
    const int n_eff = first_3d_only ? 1 : n_obs;
    for (int base = 0; base < n_eff; base += 64) {
        const int k = base + lane;
        const int row0 = tri_row_of(base, n3d);                       // first row of this chunk
        if (k < n_eff) {
            double V[3][4], hr;
            int d;
            if (base == 0) d = tri_block_rows(mine, P, x, V, &hr);
            else { TriLane L; tri_load_obs(frames, cam_t, obs[k], &L); d = tri_block_rows(L, P, x, V, &hr); }
            const int r = tri_row_of(k, n3d) - row0;
            for (int q = 0; q < d; q++) {
#pragma unroll
                for (int i = 0; i < 4; i++) sh.V[r + q][i] = V[q][i];
            }
            sh.half_rho[lane] = hr;
        }
        __syncthreads();
        const int cnt = min(64, n_eff - base);
        const int n_rows = tri_row_of(base + cnt, n3d) - row0;
        if (lane < 9) {
            // rows in block order; the loads of several rows are in flight while the (ordered) additions retire
            int r = 0;
            for (; r + 4 <= n_rows; r += 4) {
                const double a0 = sh.V[r][ia], b0 = sh.V[r][ja], a1 = sh.V[r + 1][ia], b1 = sh.V[r + 1][ja];
                const double a2 = sh.V[r + 2][ia], b2 = sh.V[r + 2][ja], a3 = sh.V[r + 3][ia], b3 = sh.V[r + 3][ja];
                acc += a0 * b0; acc += a1 * b1; acc += a2 * b2; acc += a3 * b3;
            }
            for (; r < n_rows; r++) acc += sh.V[r][ia] * sh.V[r][ja];
        } else if (lane == 9) {
            for (int p = 0; p < cnt; p++) acc += sh.half_rho[p];
        }
        __syncthreads();
    }
    const double h00 = __shfl(acc, 0), h01 = __shfl(acc, 1), h02 = __shfl(acc, 2), h11 = __shfl(acc, 3), h12 = __shfl(acc, 4), h22 = __shfl(acc, 5);
    E->H[0] = h00; E->H[1] = h01; E->H[2] = h02; E->H[3] = h01; E->H[4] = h11; E->H[5] = h12; E->H[6] = h02; E->H[7] = h12; E->H[8] = h22;
    E->g[0] = __shfl(acc, 6); E->g[1] = __shfl(acc, 7); E->g[2] = __shfl(acc, 8);
    E->cost = __shfl(acc, 9);
}

__global__ void __launch_bounds__(64)
triangulate_wave_kernel(const TriFrame* __restrict__ frames, const double* __restrict__ cam_t, const velo_tri_obs* __restrict__ obs,
                        const int* __restrict__ off, int n, TriParams P, float* __restrict__ pts, const unsigned char* __restrict__ init,
                        velo_tri_result* __restrict__ results)
#if VELO_DEF_LOAD
{
    __shared__ TriShared sh;
    const int l = blockIdx.x;
    if (l >= n) return;
    const int lane = threadIdx.x;
    const int b = off[l], n_obs = off[l + 1] - b;
    const bool guess = init != nullptr && init[l] != 0;
    double x[3] = {0.0, 0.0, 10.0};                                   // velo.h:1043
    if (guess) { x[0] = pts[3 * l]; x[1] = pts[3 * l + 1]; x[2] = pts[3 * l + 2]; }   // velo.h:1044-1049
    velo_tri_result S;
    S.n_solves = 0; S.termination = VELO_CONVERGENCE; S.lm_iterations = 0; S.evaluations = 0; S.final_cost = 0.0;
    TriLane mine;
    mine.kind = VELO_TRI_OBS_3D;
    if (lane < n_obs) tri_load_obs(frames, cam_t, obs[b + lane], &mine);
    int n3d = 0;                                                      // 3-D observations come first (host partition)
    for (int k0 = 0; k0 < n_obs; k0 += 64) {
        const int k = k0 + lane;
        n3d += (int)__popcll(__ballot(k < n_obs && obs[b + k].kind == VELO_TRI_OBS_3D));
    }
    if (!guess && n3d > 0)                                            // velo.h:1080-1083
        tri_solve([&](const double* xx, TriEval* E) { tri_evaluate_wave(sh, frames, cam_t, obs + b, n_obs, n3d, true, mine, P, xx, E); }, P, x, &S);
    if (n_obs > 0)                                                    // velo.h:1123
        tri_solve([&](const double* xx, TriEval* E) { tri_evaluate_wave(sh, frames, cam_t, obs + b, n_obs, n3d, false, mine, P, xx, E); }, P, x, &S);
    if (lane == 0) {
        pts[3 * l] = (float)x[0]; pts[3 * l + 1] = (float)x[1]; pts[3 * l + 2] = (float)x[2];        // velo.h:1124-1126
        if (results) results[l] = S;
    }
}
#else
;
#endif

}  // namespace velo
