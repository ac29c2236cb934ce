// velo_kernels.h -- device data layout + HIP kernels of the scan-matching core (gfx950 / CDNA4 only).
//
// Kernel map (SURVEY.md section 8(a) rows in brackets):
//   pack_points_kernel, bbox_kernel, grid_count/scan/scatter  [T1]  target index: uniform grid, cell >= gate radius
//   assoc_search_kernel                                      [A1-A6] query transform, exact per-ring 1-NN via the
//                                                                    27-cell neighbourhood, top-2 rings, ring neighbour,
//                                                                    triangle normal -> correspondence table
//   visual_gate_kernel                                       [G1]   visual block choice + outlier gate
//   eval_kernel                                              [R1-R5,L1] residuals + dual-number Jacobians + robust
//                                                                    weights -> per-workgroup 28-double partial sums
//   lm_step_kernel                                           [S1]   deterministic final reduction + one trust-region
//                                                                    Levenberg-Marquardt state transition, on device
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "velo_device_math.h"

// ---- translation units (round 6) ---------------------------------------------------------------------------------------------------------
// The library is built from several units compiled side by side (build.py): velo_hip.hip holds the host side of the C-ABI and DEFINES NO
// KERNEL; each kernel family is defined -- its device code generated -- in exactly one unit:
//     VELO_DEF_LOAD   velo_unit_load.hip    scan ingestion, index build, the group loaders, the widened rows (projection, depth, triangulation)
//     VELO_DEF_ASSOC  velo_unit_assoc.hip   the association family (tube / direct / asker / lane / box-walk kernels, seeds, merge)
//     VELO_DEF_LMA    velo_unit_lm_a.hip    visual gate, residual statistics, the sweep and step kernels, one-launch iterations, peer exchange
//     VELO_DEF_LMB    velo_unit_lm_b.hip    the fused sweep + step family of the lock-step groups, single-launch solves, chain finish, functors
// Every other unit sees a kernel's DECLARATION (the `#else ;` branch behind its signature) and launches it through the host stub the
// defining unit exports; template kernels are instantiated explicitly in their unit and declared `extern template` elsewhere (the list
// at the end of this header).  Device functions and data layouts stay visible to every unit.
#ifndef VELO_DEF_LOAD
#define VELO_DEF_LOAD 0
#endif
#ifndef VELO_DEF_ASSOC
#define VELO_DEF_ASSOC 0
#endif
#ifndef VELO_DEF_LMA
#define VELO_DEF_LMA 0
#endif
#ifndef VELO_DEF_LMB
#define VELO_DEF_LMB 0
#endif

namespace velo {

constexpr int kWave = 64;
constexpr int kNumAcc = 28;           // 21 upper-triangular JtJ + 6 Jtr + cost
constexpr int kEvalThreads = 256;
constexpr int kMaxEvalBlocks = 512;
constexpr int kAssocThreads = 256;

// ---- uniform grid over the target cloud (one per distinct gate radius) ------------------------------------
struct GridDesc {
    float ox, oy, oz;       // origin = bbox min
    float inv_h;            // 1 / cell size (one fine grid, cell ~ 1.01 x the smallest gate radius, serves every gate)
    int nx, ny, nz;
    int ncells;
};

// Cell-sorted copy of the target: float4 {x, y, z, bits(global ring-major target index)} + the point's ring, so that a
// workgroup can stage whole cell runs into LDS with one coalesced 16-byte load per candidate.  kGridPad sentinel
// entries (x = +inf) follow the last real point.
constexpr int kGridPad = 16;
struct GridView {
    GridDesc d;
    const int* __restrict__ cell_start;      // dense: [ncells + 1] start of every cell; compressed: start of every OCCUPIED cell, [n_occupied + 1]
    const float4* __restrict__ sorted;       // [n_finite + kGridPad]
    const int* __restrict__ sring;           // ring of sorted[j]
    // Compressed table (large grids: an accumulated map's millions of cells, of which a few per cent hold points): per row (y, z) of
    // the grid `wpr` 64-bit words, bit x & 63 of word x >> 6 = "cell x of this row is occupied", and the number of occupied cells before
    // each word.  start(row, x) = cell_start[wprefix[w] + popcount(wmask[w] below bit x)] -- 0.19 bytes per cell instead of 4, so the
    // table of the 2M-point map (11 M cells) is 2 MB and stays in every XCD's L2, at the price of one dependent load per look-up.
    const unsigned long long* __restrict__ wmask;   // null: dense table
    const int* __restrict__ wprefix;
    int wpr;                                  // words per row = ceil(nx / 64)
};
// first sorted point of cell x (0..nx: nx = one past the row's last cell) of grid row `row` = z * ny + y
__device__ __forceinline__ int grid_start(const GridView& G, int row, int x) {
    if (G.wmask == nullptr) return G.cell_start[row * G.d.nx + x];
    const int w = row * G.wpr + (x >> 6);
    const unsigned long long m = G.wmask[w];
    const int k = G.wprefix[w] + __popcll(m & ((1ull << (x & 63)) - 1ull));
    return G.cell_start[k];
}

// ---- pose scalars of one association round, computed on the HOST in double with the same libm the CPU
// restatement uses, so that the float coordinates of the transformed queries are bit-identical (row A1) -----
struct PoseScalars {
    double w[3];     // omega
    double t[3];
    double c, s;     // cos / sin(theta)
    double u[3];     // omega / theta
    double omc;      // 1 - cos(theta)
    int small;       // theta^2 <= DBL_EPSILON -> first-order branch
};

// the same for host and device: with the pinned sin / cos (velo_device_math.h) and correctly rounded sqrt and division the device
// computes a round's pose scalars with the bits the host (and the oracle) gets, so a frame_to_frame can run as ONE chain of launches
__host__ __device__ inline void pose_scalars_compute(const double x[6], PoseScalars* S) {
    for (int k = 0; k < 3; k++) { S->w[k] = x[k]; S->t[k] = x[3 + k]; S->u[k] = 0.0; }
    S->c = 0.0; S->s = 0.0; S->omc = 0.0;
    const double theta2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    if (theta2 > 2.220446049250313e-16) {
        const double theta = sqrt(theta2);
        velo_sincos(theta, &S->s, &S->c);
        const double ti = 1.0 / theta;
        S->u[0] = x[0] * ti; S->u[1] = x[1] * ti; S->u[2] = x[2] * ti;
        S->omc = 1.0 - S->c;
        S->small = 0;
    } else {
        S->small = 1;
    }
}
// Device-driven rounds ("chain mode"): the LM step that finishes a solve leaves the pose scalars of its result here, and the next
// round's association kernel reads them -- no host round trip between a solve and the next association.  ready = 0 while a solve
// is running: an association kernel that finds it so was enqueued behind a solve that needed more launches than the host had
// predicted; it raises the chain's failure flag and everything behind it returns at once (the host then repeats the call).
struct PoseRecord {
    PoseScalars P;
    int ready;
    int pad;
};
// what the host wants to know about a finished solve (velo_solve_summary), left by the step that finished it
struct SolveLog {
    double x[6];
    double initial_cost, final_cost;
    int termination, iter, evals, n_valid;
};

__device__ __forceinline__ int cell_coord(float p, float o, float inv_h, int n) {
    float f = (p - o) * inv_h;
    f = fminf(fmaxf(f, -2.0f), (float)(n + 1));
    return (int)floorf(f);
}

// (velo_lm_ag.hip, the translation unit of the all-gather solve, defines VELO_UNIT_LM_ONLY: it needs the solver half of this header only, and every
//  kernel it would compile here -- ingest, index build, the association family -- would be a second, unused copy in the shared library)
#ifndef VELO_UNIT_LM_ONLY
// ---- point packing: (stride-addressed xyz) -> float4 ---------------------------------------------------------
// records from page-locked host memory into a device buffer, as a launch: the runtime's copy of a few tens of KB (its DMA path) was seen to
// block the submitting thread for 6-14 ms once in a hundred steps with four busy queues; a launch never does
__global__ void __launch_bounds__(256) upload_words_kernel(const int* __restrict__ src, int* __restrict__ dst, int n_words)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_words) dst[i] = __builtin_nontemporal_load(src + i);
}
#else
;
#endif
__global__ void pack_points_kernel(const char* __restrict__ src, int64_t stride, int n, float4* __restrict__ dst)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = (const float*)(src + (int64_t)i * stride);
    dst[i] = make_float4(p[0], p[1], p[2], 0.0f);
}
#else
;
#endif

// ring id per point (binary search in ring_offsets) -- the target's gidx -> ring map
// (a context may hold only a block of whole rings of the target -- target-sharded mode -- so ring ids and point
//  indices that leave the kernels are GLOBAL: local ring + first_ring, local index + first_point)
// (bbox_init: the six keys bbox_kernel folds into -- min keys all ones, max keys zero -- are initialised here, one launch ahead of it,
//  instead of by a copy operation of their own)
__global__ void ring_of_kernel(const int* __restrict__ off, int n_rings, int n, int first_ring, int* __restrict__ ring_of, unsigned* __restrict__ bbox_init)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (bbox_init && i < 6) bbox_init[i] = i < 3 ? 0xffffffffu : 0u;
    if (i >= n) return;
    int lo = 0, hi = n_rings;   // find r with off[r] <= i < off[r+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
    ring_of[i] = lo + first_ring;
}
#else
;
#endif

// Ring-major copy of the target with one wrap-around sentinel on either side of every ring:
//     ring r (n_r points at [off[r], off[r+1])) -> pad[off[r] + 2 r] = its LAST point, then its points, then its FIRST point,
// so the cyclic ring neighbours (velo.h:852-854) of point i of local ring r are pad[i + 2 r] and pad[i + 2 r + 2], whatever i:
// the finish of the association reads three adjacent records and needs neither the ring bounds nor a second trip for the wrap.
__global__ void pad_rings_kernel(const float4* __restrict__ tgt, const int* __restrict__ off, const int* __restrict__ ring_of, int n, int first_ring,
                                 float4* __restrict__ pad)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = ring_of[i] - first_ring, base = off[r], end = off[r + 1];
    const float4 p = tgt[i];
    pad[i + 2 * r + 1] = p;
    if (i == base) pad[end + 2 * r + 1] = p;          // trailing sentinel = first point
    if (i == end - 1) pad[base + 2 * r] = p;          // leading sentinel = last point
}
#else
;
#endif

// compact query points: qpts[i] = src[q_src[i]] (icp_skip > 1; with icp_skip == 1 the source cloud itself is the list)
__global__ void gather_queries_kernel(const float4* __restrict__ src, const int* __restrict__ q_src, int nq, float4* __restrict__ qpts)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) qpts[i] = src[q_src[i]];
}
#else
;
#endif

// ---- patch order of the queries ---------------------------------------------------------------------------------------------------
// The tube kernel gives 64 consecutive queries to a workgroup.  In ring order that is a 2 m long line of one ring (a tube of ~125
// cells); the same 64 queries taken as 8 neighbouring rings x 8 consecutive points are a patch of ~0.5 m x 0.25 m (~60 cells):
// fewer rows, fewer staged candidates, tighter bounds.  So the query LIST is laid out in patch order: bands of kPatchRings source
// rings, each ring of a band cut into the same number M of segments (M = ceil(longest ring of the band / kPatchLen)); the list runs
// band by band, segment by segment, ring by ring, point by point.  Everything indexed by query (seeds, table, LM rows) follows the
// list; the results per query are what they were, and the calls that hand tables out restore ring order (patch_position is the map,
// the same integer arithmetic on host and device).  Closed form, no sort: segment j of a ring with n queries starts at ceil(j n / M).
constexpr int kPatchRingsDefault = 8, kPatchLenDefault = 8;      // (4 x 16, 16 x 4, 2 x 32, 8 x 4, 4 x 8 measured within 3 % or worse)
__host__ __device__ inline int patch_first(int j, int n, int M) { return (int)(((long long)j * n + M - 1) / M); }
__host__ __device__ inline int patch_position(const int* q_off, int n_rings, int r, int k, int kPatchRings = 8, int kPatchLen = 8) {
    const int r0 = (r / kPatchRings) * kPatchRings, r1 = (r0 + kPatchRings < n_rings) ? r0 + kPatchRings : n_rings;
    int nmax = 0;
    for (int rr = r0; rr < r1; rr++) { const int nn = q_off[rr + 1] - q_off[rr]; nmax = nn > nmax ? nn : nmax; }
    const int M = (nmax + kPatchLen - 1) / kPatchLen > 1 ? (nmax + kPatchLen - 1) / kPatchLen : 1;
    const int n = q_off[r + 1] - q_off[r];
    const int j = (int)((long long)k * M / n);                        // segment of this query: first(j) <= k < first(j + 1)
    int pos = q_off[r0];                                               // the earlier bands
    for (int rr = r0; rr < r1; rr++) {
        const int nn = q_off[rr + 1] - q_off[rr];
        const int f = patch_first(j, nn, M);
        pos += f;                                                      // the earlier segments of every ring of the band
        if (rr < r) pos += patch_first(j + 1, nn, M) - f;              // this segment of the earlier rings
    }
    return pos + (k - patch_first(j, n, M));
}
// query list: position -> global source index.  Ring r contributes ceil(n_r / skip) queries (velo.h:806-807); patch != 0: the list is
// in patch order, else in the reference's order (ring by ring).
__global__ void query_list_kernel(const int* __restrict__ src_off, const int* __restrict__ q_off, int n_rings,
                                  int skip, int nq, int patch, int patch_rings, int patch_len, int* __restrict__ q_src)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int lo = 0, hi = n_rings;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (q_off[mid] <= i) lo = mid; else hi = mid; }
    const int k = i - q_off[lo];
    q_src[patch ? patch_position(q_off, n_rings, lo, k, patch_rings, patch_len) : i] = src_off[lo] + k * skip;
}
#else
;
#endif

// ---- scan ingestion on the device: KITTI records -> camera-0-frame rings (kitti.h:121-185), "next" row 1 of SURVEY 8(f) ----
// 1. ring_break_kernel: flag[i] = i > 0 && x_i > 0 && (y_i > 0) != (y_{i-1} > 0)            (kitti.h:164-168, velodyne frame)
// 2. exclusive scan of the flags (scan_tiles/sums/add above) -> ring id per point, ring count
// 3. ring_offsets_kernel: off[ring] = first point of that ring (the flagged points), off[n_rings] = n
// 4. ring_reorder_kernel: camera-frame copy  q = velo_to_cam * p  in float, stored at  new[i] = old[n-1-((i + n/2) % n)]  (kitti.h:178-183)
__global__ void ring_break_kernel(const char* __restrict__ rec, int64_t stride, int n, int* __restrict__ flag)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = (const float*)(rec + (int64_t)i * stride);
    int f = 0;
    if (i > 0) {
        const float* q = (const float*)(rec + (int64_t)(i - 1) * stride);
        f = (p[0] > 0.f && ((p[1] > 0.f) != (q[1] > 0.f))) ? 1 : 0;
    }
    flag[i] = f;
}
#else
;
#endif
// ring_id[i] = (exclusive scan of flag)[i] + flag[i]; the flagged points start rings 1.., point 0 starts ring 0
__global__ void ring_offsets_kernel(const int* __restrict__ excl, const int* __restrict__ flag, int n, int* __restrict__ ring_id, int* __restrict__ off,
                                    int* __restrict__ n_rings_out)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = excl[i] + flag[i];
    ring_id[i] = r;
    if (i == 0 || flag[i]) off[r] = i;
    if (i == n - 1) { off[r + 1] = n; *n_rings_out = r + 1; }
}
#else
;
#endif
struct Mat34f { float m[12]; };   // rows of velo_to_cam (kitti.h:100-107)
__global__ void ring_reorder_kernel(const char* __restrict__ rec, int64_t stride, int n, const int* __restrict__ ring_id, const int* __restrict__ off,
                                    Mat34f M, float4* __restrict__ dst)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = (const float*)(rec + (int64_t)i * stride);
    const float x = p[0], y = p[1], z = p[2];
    // pcl::transformPointCloud<float> [3P]: linear part column by column, then the translation, all in float
    const float cx = ((M.m[0] * x + M.m[1] * y) + M.m[2] * z) + M.m[3];
    const float cy = ((M.m[4] * x + M.m[5] * y) + M.m[6] * z) + M.m[7];
    const float cz = ((M.m[8] * x + M.m[9] * y) + M.m[10] * z) + M.m[11];
    const int r = ring_id[i], base = off[r], m = off[r + 1] - base, j = i - base;
    // new[i'] = old[m-1-((i' + m/2) % m)]  <=>  i' = (m-1-j - m/2) mod m
    int ip = (m - 1 - j - m / 2) % m;
    if (ip < 0) ip += m;
    dst[base + ip] = make_float4(cx, cy, cz, 0.f);
}
#else
;
#endif

// ---- bounding box of the finite points: per-block min/max then atomics on order-preserving integer keys ----
__device__ __forceinline__ unsigned f2key(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float key2f(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; __builtin_memcpy(&f, &u, 4); return f;
#endif
}
__global__ void __launch_bounds__(256)
bbox_kernel(const float4* __restrict__ pts, int n, unsigned* __restrict__ mnmx /* [6]: min xyz, max xyz keys */)
#if VELO_DEF_LOAD
{
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
        }
    }
    __shared__ float red[4][6];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { for (int k = 0; k < 3; k++) { red[wid][k] = mn[k]; red[wid][3 + k] = mx[k]; } }
    __syncthreads();
    if (threadIdx.x < 6) {     // one atomic per block and component (same-address atomics serialise)
        const int k = threadIdx.x;
        float v = red[0][k];
        for (int w = 1; w < 4; w++) v = (k < 3) ? fminf(v, red[w][k]) : fmaxf(v, red[w][k]);
        if (k < 3) atomicMin(&mnmx[k], f2key(v)); else atomicMax(&mnmx[k], f2key(v));
    }
}
#else
;
#endif

// ---- a scan enters a context in ONE launch per side -----------------------------------------------------------------------------------
// With several registrations in flight a queue operation costs 5-10 us of hand-over whatever it does, and loading a pair used to take
// sixteen of them (pack, ring ids, padded rings, bounding box, two clears, ... each a launch or a copy of its own) against ~30 for the
// registration itself.  target_ingest_kernel is pack_points + ring_of + pad_rings + bbox (and clears the scan's status words);
// source_ingest_kernel is pack_points + query_list + gather_queries (the query blocks read the caller's records themselves, so they
// do not wait for the packed copy).  Same arithmetic, same tables as the separate kernels, which the other entries keep using.
constexpr int kIngestPerThread = 4;                                    // points per thread: 1,024 per workgroup, 6 bounding-box atomics per workgroup at most
// IN-PLACE CONTRACT: a promotion (promote_begin: the source scan becomes the target) runs this kernel with src == tgt and stride 16 -- the
// records are already packed.  src and tgt therefore carry NO __restrict__, and the body must keep to "thread i reads record i of src, then
// writes record i of tgt": no staging of other threads' records, no reads of a neighbour's record from src (pad sentinels take the thread's
// OWN point p), no vector loads across records.  tests/test_gpu_next_rows.py compares a promoted target with a freshly loaded one bit for bit.
__global__ void __launch_bounds__(256)
target_ingest_kernel(const char* src, int64_t stride, int n, const int* __restrict__ off, int n_rings, int first_ring,
                     float4* tgt, int* __restrict__ ring_of, float4* __restrict__ pad, unsigned* __restrict__ mnmx,
                     unsigned long long* __restrict__ lb_status, int lb_words)
#if VELO_DEF_LOAD
{
    for (int j = blockIdx.x * 256 + threadIdx.x; j < lb_words; j += gridDim.x * 256) lb_status[j] = 0ull;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int u = 0; u < kIngestPerThread; u++) {
        const int i = (blockIdx.x * kIngestPerThread + u) * 256 + threadIdx.x;
        if (i >= n) continue;
        const float* q = (const float*)(src + (int64_t)i * stride);
        const float4 p = make_float4(q[0], q[1], q[2], 0.0f);
        tgt[i] = p;
        int lo = 0, hi = n_rings;   // find r with off[r] <= i < off[r+1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
        ring_of[i] = lo + first_ring;
        const int base = off[lo], end = off[lo + 1];
        pad[i + 2 * lo + 1] = p;
        if (i == base) pad[end + 2 * lo + 1] = p;          // trailing sentinel = first point
        if (i == end - 1) pad[base + 2 * lo] = p;          // leading sentinel = last point
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], o));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o));
        }
    }
    __shared__ float red[4][6];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { for (int k = 0; k < 3; k++) { red[wid][k] = mn[k]; red[wid][3 + k] = mx[k]; } }
    __syncthreads();
    if (threadIdx.x < 6) {     // one atomic per workgroup and component, and only where this workgroup moves the box (a stale look is harmless)
        const int k = threadIdx.x;
        float v = red[0][k];
        for (int w = 1; w < 4; w++) v = (k < 3) ? fminf(v, red[w][k]) : fmaxf(v, red[w][k]);
        const unsigned key = f2key(v), cur = mnmx[k];
        if (k < 3) { if (v < 3.0e38f && key < cur) atomicMin(&mnmx[k], key); } else { if (v > -3.0e38f && key > cur) atomicMax(&mnmx[k], key); }
    }
}
#else
;
#endif
__global__ void __launch_bounds__(256)
source_ingest_kernel(const char* __restrict__ raw, int64_t stride, int n, float4* __restrict__ src, int nb_pack,
                     const int* __restrict__ src_off, const int* __restrict__ q_off, int n_rings, int skip, int nq, int patch, int patch_rings, int patch_len,
                     int* __restrict__ q_src, float4* __restrict__ qpts, unsigned* __restrict__ mnmx)
#if VELO_DEF_LOAD
{
    if ((int)blockIdx.x < nb_pack) {
        // (the pack blocks also take the cloud's bounding box -- the keys target_ingest_kernel would compute for the same points: a drive
        //  promotes this scan to target one frame later, and with the box already on the host the index build needs no host wait)
        const int i = blockIdx.x * 256 + threadIdx.x;
        float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        if (i < n) {
            const float* p = (const float*)(raw + (int64_t)i * stride);
            const float4 v = make_float4(p[0], p[1], p[2], 0.0f);
            src[i] = v;
            if (isfinite(v.x) && isfinite(v.y) && isfinite(v.z)) { mn[0] = mx[0] = v.x; mn[1] = mx[1] = v.y; mn[2] = mx[2] = v.z; }
        }
        if (!mnmx) return;
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
        }
        __shared__ float red[4][6];
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        if (lane == 0) { for (int k = 0; k < 3; k++) { red[wid][k] = mn[k]; red[wid][3 + k] = mx[k]; } }
        __syncthreads();
        if (threadIdx.x < 6) {
            const int k = threadIdx.x;
            float v = red[0][k];
            for (int w = 1; w < 4; w++) v = (k < 3) ? fminf(v, red[w][k]) : fmaxf(v, red[w][k]);
            const unsigned key = f2key(v), cur = mnmx[k];
            if (k < 3) { if (v < 3.0e38f && key < cur) atomicMin(&mnmx[k], key); } else { if (v > -3.0e38f && key > cur) atomicMax(&mnmx[k], key); }
        }
        return;
    }
    const int i = ((int)blockIdx.x - nb_pack) * 256 + threadIdx.x;
    if (i >= nq) return;
    int lo = 0, hi = n_rings;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (q_off[mid] <= i) lo = mid; else hi = mid; }
    const int k = i - q_off[lo];
    const int pos = patch ? patch_position(q_off, n_rings, lo, k, patch_rings, patch_len) : i, si = src_off[lo] + k * skip;
    q_src[pos] = si;
    if (qpts) { const float* p = (const float*)(raw + (int64_t)si * stride); qpts[pos] = make_float4(p[0], p[1], p[2], 0.0f); }
}
#else
;
#endif

// ---- grid build: count, exclusive scan (3 kernels), scatter ----------------------------------------------------
__device__ __forceinline__ int cell_of_point(const GridDesc& g, const float4& p) {
    int cx = cell_coord(p.x, g.ox, g.inv_h, g.nx), cy = cell_coord(p.y, g.oy, g.inv_h, g.ny), cz = cell_coord(p.z, g.oz, g.inv_h, g.nz);
    cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
    return (cz * g.ny + cy) * g.nx + cx;
}
// The table is built IN PLACE with an offset of one: cell c counts into table[c + 1]; the exclusive scan of table[1..ncells] leaves
// table[c + 1] = start(c); the scatter takes its slots with atomicAdd(&table[c + 1], 1), which leaves table[c + 1] = start(c + 1).
// table[0] stays 0, so in the end table[k] = start(k) for k = 0..ncells -- no cursor copy of the table (a 134 MB write per build
// of the 2M-point map's 33 M cells), no second pass over it.
// Consecutive points of a ring are neighbours in space, on an accumulated map often several per cell: the lanes of a wave that hold a RUN
// of equal cell ids send ONE atomic for the run (count) / take one block of slots for it (scatter).  The atomics on a cell's word come
// from workgroups on different XCDs -- every one moves the line to another L2 -- so fewer of them is what shortens both kernels.
// run_of_lane: `head` lanes start a run; returns the run's first lane and its length (valid on every lane with c >= 0).
__device__ __forceinline__ void run_of_lane(int c, int lane, bool* head, int* first, int* len) {
    const int prev = __shfl_up(c, 1);
    const bool h = c >= 0 && (lane == 0 || prev != c);
    const unsigned long long hm = __ballot(h), vm = __ballot(c >= 0);
    const unsigned long long upto = (2ull << lane) - 1ull;                       // lanes 0..lane
    const int f = 63 - __clzll((long long)((hm & upto) | 1ull));                 // (| 1: keeps clz defined on lanes before the first run)
    const unsigned long long stop = (hm | ~vm) & ~upto;                          // the next run or the next lane without a cell
    *head = h; *first = f; *len = (stop ? (int)__ffsll((long long)stop) - 1 : 64) - f;
#ifdef VELO_NO_RUN_AGG                                                   // A/B build: one atomic per point, as before round 3
    *head = c >= 0; *first = lane; *len = 1;
#endif
}
__global__ void grid_count_kernel(GridDesc g, const float4* __restrict__ pts, int n, int* __restrict__ cell_of, int* __restrict__ table)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1;
    if (i < n) {
        const float4 p = pts[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) c = cell_of_point(g, p);
        cell_of[i] = c;
    }
    bool head; int first, len;
    run_of_lane(c, threadIdx.x & 63, &head, &first, &len);
    if (head) atomicAdd(&table[c + 1], len);
}
#else
;
#endif

// ---- compressed table (large grids) ----------------------------------------------------------------------------------------------------
// mark: cell id per point (cell_of) + one bit per occupied cell (a run of equal cells sends one atomicOr); word_popc: occupied cells per
// word, ready for the one-pass scan; ccount: every point's COMPACT cell number (its rank among the occupied cells, overwriting cell_of)
// counted into table[k + 1] like grid_count_kernel does for the dense table.  Scan and scatter are the dense path's kernels on the
// compact table: ~8 B per point of table traffic instead of 4 B per CELL read and written twice.
__global__ void grid_mark_kernel(GridDesc g, const float4* __restrict__ pts, int n, int* __restrict__ cell_of, unsigned long long* __restrict__ wmask, int wpr)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1;
    if (i < n) {
        const float4 p = pts[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) c = cell_of_point(g, p);
        cell_of[i] = c;
    }
    bool head; int first, len;
    run_of_lane(c, threadIdx.x & 63, &head, &first, &len);
    if (head) {
        const int row = c / g.nx, x = c - row * g.nx;
        const unsigned long long bit = 1ull << (x & 63);
        unsigned long long* w = &wmask[(size_t)row * wpr + (x >> 6)];
        if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(w, bit);   // (a set bit stays set: most runs find theirs set already)
    }
}
#else
;
#endif
__global__ void word_popc_kernel(const unsigned long long* __restrict__ wmask, int n_words, int* __restrict__ wprefix)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) wprefix[i] = __popcll(wmask[i]);
    else if (i == n_words) wprefix[i] = 0;                              // the sentinel word behind the last row
}
#else
;
#endif
__global__ void grid_ccount_kernel(GridDesc g, int* __restrict__ cell_of, int n, const unsigned long long* __restrict__ wmask, const int* __restrict__ wprefix, int wpr,
                                   int* __restrict__ table)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int k = -1;
    if (i < n) {
        const int c = cell_of[i];
        if (c >= 0) {
            const int row = c / g.nx, x = c - row * g.nx;
            const int w = row * wpr + (x >> 6);
            k = wprefix[w] + __popcll(wmask[w] & ((1ull << (x & 63)) - 1ull));
        }
        cell_of[i] = k;
    }
    bool head; int first, len;
    run_of_lane(k, threadIdx.x & 63, &head, &first, &len);
    if (head) atomicAdd(&table[k + 1], len);
}
#else
;
#endif

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;                      // per thread
constexpr int kScanTile = kScanThreads * kScanItems;

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    __shared__ int wave_sums[kScanThreads / kWave];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    if (lane == 63) wave_sums[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / kWave; w++) { const int s = wave_sums[w]; if (w < wid) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
// Exclusive scan in ONE pass over the data (decoupled look-back): a workgroup takes the next tile by ticket (so every tile before
// it is already running), scans it locally, publishes its total, and wave 0 looks back over the tiles before it -- 64 status
// words at a time, each {flag, value} in one 64-bit word read with a device-scope atomic -- adding totals until it meets a tile
// whose whole prefix is known.  The atomics are RELAXED on purpose: a status word carries everything its reader needs, and an
// acquire / release at device scope costs an L2 invalidate / write-back per spin (measured: the whole chip slows down 2x).  Read once, write once: 2 x 134 MB for the map's table instead of 5 x (three-kernel scan + cursor).
// Elements per thread and tile: 16 (4,096-element tiles) for a scan's table, 64 (16,384) for a map's -- the one-pass scan is bound by its
// NUMBER of tiles (ticket, status word, look-back per tile), not by bytes: on the 23 M cells of the 2M-point map 1,024-element tiles take
// 272 us, 4,096: 96 us, 8,192: 67 us, 16,384: 56 us (185 MB: 3.3 TB/s); on the 0.87 M cells of a scan 16,384-element tiles leave 53
// workgroups for the chip (11.0 against 7.9 us).
constexpr int kLbItemsSmall = 16, kLbItemsLarge = 64;
constexpr int kLbLargeFrom = 1 << 22;              // table entries from which the large tiles are used
__host__ __device__ constexpr int lb_tile(int items) { return kScanThreads * items; }
constexpr unsigned long long kLbAgg = 1ull << 62, kLbPrefix = 2ull << 62, kLbFlags = 3ull << 62;
template <int kLbItems>
__device__ __forceinline__ void
scan_lookback_body(int* __restrict__ data, int n, unsigned long long* __restrict__ status, int* __restrict__ ticket, int* __restrict__ grand_total) {
    // A tile = kLbItems / 4 sub-tiles of kScanThreads x 4 ints; thread t holds the int4 number t of EVERY sub-tile, so a wave's loads and
    // stores are 1 KB of consecutive memory.  The sub-tiles' prefixes come out of ONE round of wave scans (all sub-tiles at once) and
    // one barrier.
    constexpr int kSub = kLbItems / 4, kWaves = kScanThreads / kWave, kLbTile = lb_tile(kLbItems);
    static_assert(kLbItems % 4 == 0, "int4 loads");
    __shared__ int s_tile, s_prefix;
    __shared__ int s_wsum[kSub][kWaves];
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1);
    __syncthreads();
    const int tile = s_tile;
    const int tbase = tile * kLbTile, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int4 v[kSub];
    int inc[kSub];
#pragma unroll
    for (int k = 0; k < kSub; k++) {
        const int i = tbase + k * (kScanThreads * 4) + threadIdx.x * 4;
        if (i + 3 < n) v[k] = *reinterpret_cast<const int4*>(data + i);
        else { v[k].x = i < n ? data[i] : 0; v[k].y = i + 1 < n ? data[i + 1] : 0; v[k].z = i + 2 < n ? data[i + 2] : 0; v[k].w = 0; }
        inc[k] = (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < kSub; k++) { const int t = __shfl_up(inc[k], off); if (lane >= off) inc[k] += t; }
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < kSub; k++) s_wsum[k][wid] = inc[k];
    }
    __syncthreads();
    int ex[kSub], run = 0;                                              // exclusive prefix of this thread's int4 inside the tile
#pragma unroll
    for (int k = 0; k < kSub; k++) {
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) { const int sw = s_wsum[k][w]; if (w < wid) before += sw; all += sw; }
        ex[k] = run + before + inc[k] - ((v[k].x + v[k].y) + (v[k].z + v[k].w));
        run += all;
    }
    const int total = run;
    if (threadIdx.x == 0) {
        s_prefix = 0;
        __hip_atomic_store(&status[tile], (tile == 0 ? kLbPrefix : kLbAgg) | (unsigned long long)(unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tile > 0 && threadIdx.x < 64) {
        int acc = 0;
        for (int idx = tile - 1;; idx -= 64) {
            const int j = idx - lane;                                  // lane 0 = the nearest tile before this one
            unsigned long long w = kLbPrefix;                          // tiles before the first: prefix 0
            if (j >= 0) { do { w = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((w & kLbFlags) == 0ull); }
            const unsigned long long pm = __ballot((w & kLbFlags) == kLbPrefix);
            const int first = (int)__ffsll((long long)pm) - 1;         // pm != 0 at the latest when j < 0 appears
            int val = (pm == 0ull || lane <= first) ? (int)(unsigned)(w & 0xffffffffull) : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off);
            acc += val;
            if (pm != 0ull) break;
        }
        if (lane == 0) {
            s_prefix = acc;
            __hip_atomic_store(&status[tile], kLbPrefix | (unsigned long long)(unsigned)(acc + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const int pre = s_prefix;
#pragma unroll
    for (int k = 0; k < kSub; k++) {
        const int i = tbase + k * (kScanThreads * 4) + threadIdx.x * 4;
        int4 q;
        q.x = pre + ex[k]; q.y = q.x + v[k].x; q.z = q.y + v[k].y; q.w = q.z + v[k].z;
        if (i + 3 < n) *reinterpret_cast<int4*>(data + i) = q;
        else { if (i < n) data[i] = q.x; if (i + 1 < n) data[i + 1] = q.y; if (i + 2 < n) data[i + 2] = q.z; }
        if (i <= n - 1 && n - 1 < i + 4) *grand_total = q.w + v[k].w;  // the thread that holds the last element: its running sum is the total (elements beyond n are zero)
    }
}
template <int kLbItems>
__global__ void __launch_bounds__(kScanThreads)
scan_lookback_kernel(int* __restrict__ data, int n, unsigned long long* __restrict__ status, int* __restrict__ ticket, int* __restrict__ grand_total) {
    scan_lookback_body<kLbItems>(data, n, status, ticket, grand_total);
}
// (three-kernel scan, kept for the device ring segmenter's small arrays)
// pass 1: tile-local exclusive scan in place (counts -> local offsets), tile totals out
__global__ void scan_tiles_kernel(int* __restrict__ data, int n, int* __restrict__ tile_sums)
#if VELO_DEF_LOAD
{
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int v[kScanItems], s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { v[k] = (base + k < n) ? data[base + k] : 0; s += v[k]; }
    int total;
    int ex = block_exclusive_scan(s, &total);
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { if (base + k < n) data[base + k] = ex; ex += v[k]; }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
#else
;
#endif
// pass 2: one workgroup scans the tile totals in place (exclusive) and writes the grand total to data_total
__global__ void scan_sums_kernel(int* __restrict__ tile_sums, int n_tiles, int* __restrict__ grand_total)
#if VELO_DEF_LOAD
{
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int start = 0; start < n_tiles; start += kScanThreads) {
        const int i = start + threadIdx.x;
        const int v = (i < n_tiles) ? tile_sums[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total);
        const int carry = carry_s;
        if (i < n_tiles) tile_sums[i] = ex + carry;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *grand_total = carry_s;
}
#else
;
#endif
// pass 3: add tile offsets; also seed the scatter cursor; element n receives the grand total
__global__ void scan_add_kernel(int* __restrict__ data, int n, const int* __restrict__ tile_sums, const int* __restrict__ grand_total,
                                int* __restrict__ cursor)
#if VELO_DEF_LOAD
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = data[i] + tile_sums[i / kScanTile]; data[i] = v; cursor[i] = v; }
    else if (i == n) data[n] = *grand_total;
}
#else
;
#endif
__device__ __forceinline__ void grid_scatter_body(const float4* __restrict__ pts, const int* __restrict__ cell_of, const int* __restrict__ ring_of, int n,
                                                  int* __restrict__ cursor, const int* __restrict__ n_finite, int first_point,
                                                  float4* __restrict__ sorted, int* __restrict__ sring, const int bx) {
    const int i = bx * 256 + threadIdx.x;
    if (i < kGridPad) {   // sentinels behind the last real point: +inf coordinates can never pass the gate
        const int j = *n_finite + i;
        sorted[j] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __int_as_float(0x7fffffff));
        sring[j] = 0x7fffffff;
    }
    const int c = i < n ? cell_of[i] : -1;
    bool head; int first, len;
    const int lane = threadIdx.x & 63;
    run_of_lane(c, lane, &head, &first, &len);
    int slot = head ? atomicAdd(&cursor[c], len) : 0;                  // cursor = table + 1 (see grid_count_kernel); one block of slots per run
    slot = __shfl(slot, first) + (lane - first);
    if (c < 0) return;
    const float4 p = pts[i];
    sorted[slot] = make_float4(p.x, p.y, p.z, __int_as_float(i + first_point));
    sring[slot] = ring_of[i];
}
__global__ void __launch_bounds__(256)
grid_scatter_kernel(const float4* __restrict__ pts, const int* __restrict__ cell_of, const int* __restrict__ ring_of, int n,
                    int* __restrict__ cursor, const int* __restrict__ n_finite, int first_point, float4* __restrict__ sorted, int* __restrict__ sring)
#if VELO_DEF_LOAD
{
    grid_scatter_body(pts, cell_of, ring_of, n, cursor, n_finite, first_point, sorted, sring, (int)blockIdx.x);
}
#else
;
#endif

// ---- the next frame of a drive in three launches for a whole lock-step group (velo_hint_next_frame) -------------------------------------
// A drive's step loads a frame on both sides: the scan the context holds as source becomes the target (ring ids, padded rings, index), the
// announced scan becomes the source (packed copy, query list, bounding box).  Through the general loaders that is thirteen queue operations
// per context -- two ring-table copies, ingest, count, scan, scatter, source ingest, the box's way back, fills -- at 5-10 us of hand-over
// each with four busy queues: 170-290 us on the stream for a group of two, as long as the host takes to turn a step around.  Here the jobs
// of ALL contexts of a group ride in the kernel arguments (ring tables included: up to kAdvRings rings), and the stages that do not depend
// on each other share a launch:
//   advance_ingest_kernel   target side: ring ids + padded rings of the (already packed) promoted cloud, cell ids and the cells' counts
//                           (the box is known from the frame's time as source, so the grid is sized before anything runs);
//                           source side: source_ingest_kernel's work; both sides' ring tables written for the kernels that follow
//   advance_scan_kernel     the one-pass scan of every context's table
//   advance_scatter_kernel  the cell-sorted copies
// Same arithmetic and the same tables as the general loaders, which every other entry keeps using (tests compare the two bit for bit).
constexpr int kAdvRings = 64, kAdvJobs = 4;
struct AdvJob {
    // target side
    const float4* tgt; int* tgt_off_dev; int* ring_of; float4* pad; unsigned long long* lb_status; int* cell_of; int* table;
    float4* sorted; int* sring; int* scan_total; int* lb_ticket; int* clear; int n_clear;   // clear: the index table's words, zeroed ahead of the counts
    GridDesc g;
    int n_t, n_rings_t, first_ring, first_point, lb_words, nb_t, nc, n_tiles, nb_sc;
    // source side
    const char* raw; long long stride; float4* src; int* src_off_dev; int* q_src; float4* qpts; unsigned* keys; unsigned* keys_next;
    unsigned* h_keys;                                                   // page-locked host memory: the source's box keys ride back with the last launch
    float4* seed_fill;                                                  // both winners' seed arrays (2 nq entries) set to "none" here, or null
    int n_s, n_rings_s, nb_pack, nb_q, skip, nq, patch, patch_rings, patch_len;
    int off_t[kAdvRings + 1], off_s[kAdvRings + 1];
};
struct AdvBatch { AdvJob job[kAdvJobs]; };

// every context's index table zeroed in one launch (the runtime's fill is a queue operation per table)
__global__ void __launch_bounds__(256)
advance_clear_kernel(AdvBatch B)
#if VELO_DEF_LOAD
{
    const AdvJob& J = B.job[blockIdx.y];
    int4* p = reinterpret_cast<int4*>(J.clear);                        // (hipMalloc'd: 256-byte aligned; n_clear rounded up to whole int4s by the host, inside the allocation)
    const int n4 = J.n_clear >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) p[i] = make_int4(0, 0, 0, 0);
}
#else
;
#endif
__global__ void __launch_bounds__(256)
advance_ingest_kernel(AdvBatch B)
#if VELO_DEF_LOAD
{
    const AdvJob& J = B.job[blockIdx.y];
    __shared__ int s_off[kAdvRings + 2], s_qoff[kAdvRings + 2];
    __shared__ float red[4][6];
    const int tid = threadIdx.x, lane = tid & 63;
    int bx = blockIdx.x;
    if (bx < J.nb_t) {
        // ---- target side: target_ingest_kernel on a packed cloud in place (nothing to pack, box known) + grid_count_kernel ----
        const int R = J.n_rings_t, n = J.n_t;
        if (tid <= R) s_off[tid] = J.off_t[tid];
        for (int j = bx * 256 + tid; j < J.lb_words; j += J.nb_t * 256) J.lb_status[j] = 0ull;
        __syncthreads();
        if (bx == 0 && tid <= R) J.tgt_off_dev[tid] = s_off[tid];
#pragma unroll
        for (int u = 0; u < kIngestPerThread; u++) {
            const int i = (bx * kIngestPerThread + u) * 256 + tid;
            int c = -1;
            if (i < n) {
                const float4 p = J.tgt[i];
                int lo = 0, hi = R;   // find r with off[r] <= i < off[r+1]
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
                J.ring_of[i] = lo + J.first_ring;
                const int base = s_off[lo], end = s_off[lo + 1];
                J.pad[i + 2 * lo + 1] = p;
                if (i == base) J.pad[end + 2 * lo + 1] = p;          // trailing sentinel = first point
                if (i == end - 1) J.pad[base + 2 * lo] = p;          // leading sentinel = last point
                if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) c = cell_of_point(J.g, p);
                J.cell_of[i] = c;
            }
            bool head; int first, len;
            run_of_lane(c, lane, &head, &first, &len);
            if (head) atomicAdd(&J.table[c + 1], len);
        }
        return;
    }
    bx -= J.nb_t;
    if (bx >= J.nb_pack) return;
    // ---- source side: source_ingest_kernel, ring and query tables from the arguments ----
    const int R = J.n_rings_s;
    if (tid <= R) s_off[tid] = J.off_s[tid];
    if (tid < 64) {                                                    // query offsets: smi = 0, skip, 2 skip, ... < n per ring (velo.h:807)
        int cnt = tid < R ? (J.off_s[tid + 1] - J.off_s[tid] + J.skip - 1) / J.skip : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(cnt, o); if (lane >= o) cnt += t; }
        if (tid == 0) s_qoff[0] = 0;
        if (tid < R) s_qoff[tid + 1] = cnt;
    }
    __syncthreads();
    if (bx == 0) {
        if (tid <= R) { J.src_off_dev[tid] = s_off[tid]; J.src_off_dev[R + 1 + tid] = s_qoff[tid]; }
        if (tid < 6) J.keys_next[tid] = tid < 3 ? 0xffffffffu : 0u;     // the keys of the frame after this one (the two slots alternate)
    }
    if (bx >= J.nb_pack) return;
    // ONE pass over the caller's records: the workgroup's 256 records come in with 16-byte loads where the layout allows (stride 12: 3 KB
    // of consecutive bytes through LDS; stride 16: one load per record) -- the records may live in page-locked HOST memory, where every
    // load instruction is a trip over the bus -- and the thread that packs point i also emits its query: ring r, every skip-th point
    // (velo.h:807) -> position in the (patch-ordered) list, the same map source_ingest_kernel's query blocks walk from the other side.
    __shared__ float s_raw[256 * 3];
    const int i = bx * 256 + tid;
    const bool have = i < J.n_s;
    const bool aligned = (reinterpret_cast<unsigned long long>(J.raw) & 15ull) == 0ull;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (J.stride == 12 && aligned) {
        const char* base = J.raw + (long long)bx * 3072;
        const int nbytes = min(3072, (J.n_s - bx * 256) * 12);
        if (tid < 192) {
            if (tid * 16 + 16 <= nbytes) *reinterpret_cast<uint4*>(&s_raw[tid * 4]) = *reinterpret_cast<const uint4*>(base + tid * 16);
            else for (int f = 0; f < 4; f++) if (tid * 16 + 4 * f + 4 <= nbytes) s_raw[tid * 4 + f] = *reinterpret_cast<const float*>(base + tid * 16 + 4 * f);
        }
        __syncthreads();
        if (have) v = make_float4(s_raw[3 * tid], s_raw[3 * tid + 1], s_raw[3 * tid + 2], 0.0f);
    } else if (have) {
        const char* q = J.raw + (long long)i * J.stride;
        if (J.stride == 16 && aligned) { const float4 w = *reinterpret_cast<const float4*>(q); v = make_float4(w.x, w.y, w.z, 0.0f); }
        else { const float* p = reinterpret_cast<const float*>(q); v = make_float4(p[0], p[1], p[2], 0.0f); }
    }
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (have) {
        J.src[i] = v;
        if (isfinite(v.x) && isfinite(v.y) && isfinite(v.z)) { mn[0] = mx[0] = v.x; mn[1] = mx[1] = v.y; mn[2] = mx[2] = v.z; }
        if (J.nq > 0) {
            int lo = 0, hi = R;                                            // ring of point i
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
            const int j = i - s_off[lo];
            if (j % J.skip == 0) {
                const int k = j / J.skip;
                const int pos = J.patch ? patch_position(s_qoff, R, lo, k, J.patch_rings, J.patch_len) : s_qoff[lo] + k;
                J.q_src[pos] = i;
                if (J.qpts) J.qpts[pos] = v;
                if (J.seed_fill) {                                     // (what attach_seeds' fill of 0xff bytes writes: no previous winner)
                    const float4 none = make_float4(__int_as_float(-1), __int_as_float(-1), __int_as_float(-1), __int_as_float(-1));
                    J.seed_fill[pos] = none; J.seed_fill[J.nq + pos] = none;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
    }
    const int wid = tid >> 6;
    if (lane == 0) { for (int k = 0; k < 3; k++) { red[wid][k] = mn[k]; red[wid][3 + k] = mx[k]; } }
    __syncthreads();
    if (tid < 6) {
        const int k = tid;
        float vv = red[0][k];
        for (int w = 1; w < 4; w++) vv = (k < 3) ? fminf(vv, red[w][k]) : fmaxf(vv, red[w][k]);
        const unsigned key = f2key(vv), cur = J.keys[k];
        if (k < 3) { if (vv < 3.0e38f && key < cur) atomicMin(&J.keys[k], key); } else { if (vv > -3.0e38f && key > cur) atomicMax(&J.keys[k], key); }
    }
}
#else
;
#endif
template <int kLbItems>
__global__ void __launch_bounds__(kScanThreads)
advance_scan_kernel(AdvBatch B) {
    const AdvJob& J = B.job[blockIdx.y];
    if ((int)blockIdx.x >= J.n_tiles) return;
    scan_lookback_body<kLbItems>(J.table + 1, J.nc, J.lb_status, J.lb_ticket, J.scan_total);
}
__global__ void __launch_bounds__(256)
advance_scatter_kernel(AdvBatch B)
#if VELO_DEF_LOAD
{
    const AdvJob& J = B.job[blockIdx.y];
    if ((int)blockIdx.x >= J.nb_sc) return;
    // (the source's bounding box, complete since the ingest launch ended, written where the host reads it before the scan's promotion one
    //  step later: two copies per group and step less)
    if (blockIdx.x == 0 && threadIdx.x < 6 && J.h_keys) J.h_keys[threadIdx.x] = J.keys[threadIdx.x];
    grid_scatter_body(J.tgt, J.cell_of, J.ring_of, J.n_t, J.table + 1, J.scan_total, J.first_point, J.sorted, J.sring, (int)blockIdx.x);
}
#else
;
#endif

// ---- association ------------------------------------------------------------------------------------------------
// Candidate order inside the scan is irrelevant: candidates are compared through the total order
//     key = (float bits of d2) << 32 | global target index        (d2 >= 0, so the bit pattern is monotone)
// and the global index is ring-major, so "lower key" == "closer, ties -> lower ring, then lower index" -- the
// outcome of the reference's sequential strict-'<' scan over rings (velo.h:825-848) on exact per-ring minima.
// best1 = min key overall; best2 = min key among candidates whose ring differs from best1's ring.
struct PartialRec;
struct AssocOut {
    float4* __restrict__ p;      // xyz of the untransformed source point, w = bits(valid)
    float4* __restrict__ n;      // unit normal
    float4* __restrict__ v0;     // plane offset
    int4* __restrict__ aux0;     // ring_i, idx_i, ring_j, idx_j      (only when want_aux)
    float4* __restrict__ aux1;   // bits(idx_k), dist_i, dist_j, -
    int* __restrict__ n_valid;   // atomic counter
    int* __restrict__ n_valid_next;   // tube kernel: the counter of the NEXT round, cleared here (or null)
    unsigned long long* __restrict__ dbg;   // [8] diagnostic cycle totals (VELO_DEBUG_SKIP & 8)
    unsigned long long* __restrict__ wg_times;   // [2 * groups] start/end s_memrealtime per workgroup (VELO_DEBUG_SKIP & 32)
    int first_ring, first_point;                  // global ids of this context's first target ring / point
    struct PartialRec* __restrict__ partial;      // target-sharded mode: per-query top-2 record instead of the table
    // tube kernel, warm start: the two winners of the last round WITH their coordinates -- {x, y, z, bits(local index)} of best1 /
    // best2 (index -1: none) and their rings -- read as seeds with one coalesced load each (no gather behind an index), rewritten
    // at the end (same entry, same workgroup); prev_a == null: no warm start
    float4* __restrict__ prev_a;
    float4* __restrict__ prev_b;
    int2* __restrict__ prev_r;
    // tube kernel on a density-shrunk grid: queries that still need cells after phase 1 ("askers": a box of up to 31 x 31 rows each) are
    // not searched inside their group's workgroup but appended here and searched by assoc_asker_kernel, a wave per kAskChunk of them
    // (ask_list == null: searched in place).  ask_count_next: the other round's counter, cleared by the tube launch.
    int* __restrict__ ask_count;
    int* __restrict__ ask_count_next;
    int* __restrict__ ask_list;                  // [n_q] query indices
    unsigned long long* __restrict__ ask_keys;   // [2 n_q] best1 / best2 keys so far, by query
    int2* __restrict__ ask_rings;                // [n_q] their rings
    int ask_map;                                 // which list entries a wave of the asker kernel takes: 0 strided, 1 a contiguous block, XCD-chunked
};

// Target-sharded mode (BASELINE config 5): what one rank knows about a query after searching ITS rings.
// Rings are disjoint across ranks, so the global winners are: best1 = min key1 over ranks (rank w*), best2 = the smaller
// of { min key1 over the other ranks, key2 of rank w* }.  The record carries the points the plane needs, so the owner of
// the query can finish without touching any other rank's cloud.  Same layout as velo_partial in include/velo_hip.h.
struct PartialRec {
    unsigned long long key1, key2;     // (d^2 bits << 32 | global index); >= key_inf when absent
    int ring1, ring2;                  // global ring ids, -1 when absent
    int idx1, idx_k, idx2, pad;        // np_i, np_k (in ring1), np_j (in ring2)
    float v0[3], v2[3], v1[3];         // best point, its chosen ring neighbour, second-best point
    float pad2;
};

__device__ __forceinline__ float dist2_f(float qx, float qy, float qz, float sx, float sy, float sz) {
    // float L2, accumulated x -> y -> z with separate multiplies and adds (utility.h:51-53, FLANN L2_Simple [3P]);
    // this file is compiled with -ffp-contract=off so no FMA is formed.
    const float dx = qx - sx, dy = qy - sy, dz = qz - sz;
    float r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

__device__ __forceinline__ void transform_query(const PoseScalars& P, const float4& p, float* qx, float* qy, float* qz) {
    // util::transform_point (utility.h:97-103) = ceres::AngleAxisRotatePoint in double, + t, rounded to float
    const double x = p.x, y = p.y, z = p.z;
    double r0, r1, r2;
    if (!P.small) {
        const double c0 = P.u[1] * z - P.u[2] * y, c1 = P.u[2] * x - P.u[0] * z, c2 = P.u[0] * y - P.u[1] * x;
        const double tmp = (P.u[0] * x + P.u[1] * y + P.u[2] * z) * P.omc;
        r0 = x * P.c + c0 * P.s + P.u[0] * tmp;
        r1 = y * P.c + c1 * P.s + P.u[1] * tmp;
        r2 = z * P.c + c2 * P.s + P.u[2] * tmp;
    } else {
        r0 = x + (P.w[1] * z - P.w[2] * y);
        r1 = y + (P.w[2] * x - P.w[0] * z);
        r2 = z + (P.w[0] * y - P.w[1] * x);
    }
    *qx = (float)(r0 + P.t[0]);
    *qy = (float)(r1 + P.t[1]);
    *qz = (float)(r2 + P.t[2]);
}

// velo.h:872-874: unit normal of the triangle (v0, v1, v2) in float, or invalid when degenerate (velo.h:873)
__device__ __forceinline__ int plane_from_points(const float v0[3], const float v1[3], const float v2[3], double norm_cond, float n[3]) {
    const float ax = v1[0] - v0[0], ay = v1[1] - v0[1], az = v1[2] - v0[2];
    const float bx = v2[0] - v0[0], by = v2[1] - v0[1], bz = v2[2] - v0[2];
    const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;   // velo.h:872
    const float nn = sqrtf(cx * cx + cy * cy + cz * cz);            // Eigen norm(): sqrt of the x,y,z sum
    if ((double)nn < norm_cond) return 0;                           // velo.h:873 (float norm vs double constant)
    n[0] = cx / nn; n[1] = cy / nn; n[2] = cz / nn;                 // velo.h:874
    return 1;
}

__device__ __forceinline__ void write_correspondence(int qi, const float4& psrc, int valid, const float n[3], const float v0[3],
                                                     int ring_i, int idx_i, int ring_j, int idx_j, int idx_k, float di, float dj,
                                                     const AssocOut& out, bool want_aux) {
    out.p[qi] = make_float4(psrc.x, psrc.y, psrc.z, __int_as_float(valid));
    out.n[qi] = valid ? make_float4(n[0], n[1], n[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    out.v0[qi] = valid ? make_float4(v0[0], v0[1], v0[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (want_aux) {
        out.aux0[qi] = make_int4(ring_i, idx_i, ring_j, idx_j);
        out.aux1[qi] = make_float4(__int_as_float(idx_k), di, dj, 0.f);
    }
    // one atomic per wave, issued by the first lane that is executing this call (lane 0 may be masked off)
    const unsigned long long m = __ballot(valid);
    const unsigned long long act = __ballot(1);
    if ((int)(threadIdx.x & 63) == (int)__ffsll((long long)act) - 1 && m) atomicAdd(out.n_valid, (int)__popcll(m));
}

// rows A3 (tail) - A6: from the two winners to (N, v0, valid) -- or, in target-sharded mode, to the partial record
__device__ __forceinline__ void finish_correspondence(
    int qi, const float4& psrc, float qx, float qy, float qz, unsigned long long b1, unsigned long long b2, unsigned long long key_inf,
    const float4* __restrict__ tgt, const int* __restrict__ tgt_off, const int* __restrict__ ring_of, double norm_cond,
    const AssocOut& out, bool want_aux) {
    int ring_i = -1, idx_i = 0, ring_j = -1, idx_j = 0, idx_k = 0;
    float di = 1e18f, dj = 1e18f;
    float v0[3] = {0.f, 0.f, 0.f}, v1[3] = {0.f, 0.f, 0.f}, v2[3] = {0.f, 0.f, 0.f};
    if (b1 < key_inf) {
        const int gi = (int)(unsigned)(b1 & 0xffffffffull) - out.first_point;      // local index
        ring_i = ring_of[gi];                                                      // global ring id
        const int base = tgt_off[ring_i - out.first_ring];
        const int n = tgt_off[ring_i - out.first_ring + 1] - base;
        idx_i = gi - base;
        di = __uint_as_float((unsigned)(b1 >> 32));
        const int k1 = (idx_i + 1) % n, k2 = (idx_i - 1 + n) % n;       // velo.h:852-854
        const float4 a1 = tgt[base + k1], a2 = tgt[base + k2];
        const float d1 = dist2_f(a1.x, a1.y, a1.z, qx, qy, qz);         // (np - pointM), same squares
        const float d2 = dist2_f(a2.x, a2.y, a2.z, qx, qy, qz);
        idx_k = (d1 < d2) ? k1 : k2;                                    // velo.h:859-863
        const float4 p0 = tgt[base + idx_i];
        const float4 p2 = (idx_k == k1) ? a1 : a2;
        v0[0] = p0.x; v0[1] = p0.y; v0[2] = p0.z; v2[0] = p2.x; v2[1] = p2.y; v2[2] = p2.z;
    }
    if (b2 < key_inf) {
        const int gj = (int)(unsigned)(b2 & 0xffffffffull) - out.first_point;
        ring_j = ring_of[gj];
        idx_j = gj - tgt_off[ring_j - out.first_ring];
        dj = __uint_as_float((unsigned)(b2 >> 32));
        const float4 p1 = tgt[gj];
        v1[0] = p1.x; v1[1] = p1.y; v1[2] = p1.z;
    }
    if (out.partial) {
        PartialRec r;
        r.key1 = b1; r.key2 = b2; r.ring1 = ring_i; r.ring2 = ring_j; r.idx1 = idx_i; r.idx_k = idx_k; r.idx2 = idx_j; r.pad = 0;
        for (int k = 0; k < 3; k++) { r.v0[k] = v0[k]; r.v2[k] = v2[k]; r.v1[k] = v1[k]; }
        r.pad2 = 0.f;
        out.partial[qi] = r;
        return;
    }
    int valid = 0;
    float n[3] = {0.f, 0.f, 0.f};
    if (ring_i >= 0 && ring_j >= 0) valid = plane_from_points(v0, v1, v2, norm_cond, n);     // velo.h:849-851,864-874
    write_correspondence(qi, psrc, valid, n, v0, ring_i, idx_i, ring_j, idx_j, idx_k, di, dj, out, want_aux);
}

// The same rows for the tube kernel, which tracks both ring ids: no ring_of look-up, and the winner with its two cyclic ring
// neighbours is three adjacent records of the padded target copy (pad_rings_kernel) -- ONE memory round trip, issued together with
// the reload of the source point; ring bounds are fetched only for the index outputs of tests / the target-sharded record.
__device__ __forceinline__ void finish_correspondence_pad(
    int qi, const float4* __restrict__ qpts, float qx, float qy, float qz, unsigned long long b1, unsigned long long b2, int ring_i_in, int ring_j_in,
    unsigned long long key_inf, const float4* __restrict__ pad, const int* __restrict__ tgt_off, double norm_cond,
    const AssocOut& out, bool want_aux) {
    const bool has1 = b1 < key_inf, has2 = b2 < key_inf;
    const int gi = has1 ? (int)(unsigned)(b1 & 0xffffffffull) - out.first_point : 0;
    const int gj = has2 ? (int)(unsigned)(b2 & 0xffffffffull) - out.first_point : 0;
    const int li = has1 ? ring_i_in - out.first_ring : 0, lj = has2 ? ring_j_in - out.first_ring : 0;
    const int pi = gi + 2 * li + 1, pj = gj + 2 * lj + 1;
    const float4 psrc = qpts[qi];
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, a1 = p0, a2 = p0;
    if (has1) { p0 = pad[pi]; a1 = pad[pi + 1]; a2 = pad[pi - 1]; }        // a1 = (np_i + 1) mod n, a2 = (np_i - 1 + n) mod n  (velo.h:852-854)
    if (has2) p1 = pad[pj];
    int ring_i = -1, idx_i = 0, ring_j = -1, idx_j = 0, idx_k = 0;
    float di = 1e18f, dj = 1e18f;
    float v0[3] = {0.f, 0.f, 0.f}, v1[3] = {0.f, 0.f, 0.f}, v2[3] = {0.f, 0.f, 0.f};
    bool next_closer = false;
    if (has1) {
        ring_i = ring_i_in;
        di = __uint_as_float((unsigned)(b1 >> 32));
        const float d1 = dist2_f(a1.x, a1.y, a1.z, qx, qy, qz);
        const float d2 = dist2_f(a2.x, a2.y, a2.z, qx, qy, qz);
        next_closer = d1 < d2;                                          // velo.h:859-863 (tie -> the -1 neighbour)
        const float4 p2 = next_closer ? a1 : a2;
        v0[0] = p0.x; v0[1] = p0.y; v0[2] = p0.z; v2[0] = p2.x; v2[1] = p2.y; v2[2] = p2.z;
    }
    if (has2) {
        ring_j = ring_j_in;
        dj = __uint_as_float((unsigned)(b2 >> 32));
        v1[0] = p1.x; v1[1] = p1.y; v1[2] = p1.z;
    }
    if (out.prev_a) {                                                   // seeds of the next round
        out.prev_a[qi] = make_float4(p0.x, p0.y, p0.z, __int_as_float(has1 ? gi : -1));
        out.prev_b[qi] = make_float4(p1.x, p1.y, p1.z, __int_as_float(has2 ? gj : -1));
        out.prev_r[qi] = make_int2(ring_i, ring_j);
    }
    if (want_aux || out.partial) {                                      // in-ring indices: tests and the target-sharded record only
        if (has1) {
            const int base = tgt_off[li], n = tgt_off[li + 1] - base;
            idx_i = gi - base;
            const int k1 = (idx_i + 1 == n) ? 0 : idx_i + 1, k2 = (idx_i == 0) ? n - 1 : idx_i - 1;
            idx_k = next_closer ? k1 : k2;
        }
        if (has2) idx_j = gj - tgt_off[lj];
    }
    if (out.partial) {
        PartialRec r;
        r.key1 = b1; r.key2 = b2; r.ring1 = ring_i; r.ring2 = ring_j; r.idx1 = idx_i; r.idx_k = idx_k; r.idx2 = idx_j; r.pad = 0;
        for (int k = 0; k < 3; k++) { r.v0[k] = v0[k]; r.v2[k] = v2[k]; r.v1[k] = v1[k]; }
        r.pad2 = 0.f;
        out.partial[qi] = r;
        return;
    }
    int valid = 0;
    float n[3] = {0.f, 0.f, 0.f};
    if (ring_i >= 0 && ring_j >= 0) valid = plane_from_points(v0, v1, v2, norm_cond, n);     // velo.h:849-851,864-874
    write_correspondence(qi, psrc, valid, n, v0, ring_i, idx_i, ring_j, idx_j, idx_k, di, dj, out, want_aux);
}

// Owner-side merge of `world` partial tables (one per target shard) for the queries [q_begin, q_end): table w holds
// records for those queries in order at tables + w * table_stride.  Ring ownership is disjoint, so (see PartialRec):
// best1 = min key1; best2 = min( key1 of the other ranks, key2 of the winning rank ).
__global__ void merge_partials_kernel(const PartialRec* __restrict__ tables, int world, int table_stride, int q_begin, int q_end,
                                      const float4* __restrict__ src, const int* __restrict__ q_src, unsigned long long key_inf,
                                      double norm_cond, AssocOut out, int want_aux)
#if VELO_DEF_ASSOC
{
    const int qi = q_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= q_end) return;
    const int li = qi - q_begin;
    unsigned long long k1 = key_inf, k2 = key_inf;
    int w1 = -1, w2 = -1, second_is_key1 = 0;
    for (int w = 0; w < world; w++) {
        const unsigned long long c = tables[(size_t)w * table_stride + li].key1;
        if (c < k1) { k2 = k1; w2 = w1; second_is_key1 = 1; k1 = c; w1 = w; }
        else if (c < k2) { k2 = c; w2 = w; second_is_key1 = 1; }
    }
    if (w1 >= 0) {
        const unsigned long long c = tables[(size_t)w1 * table_stride + li].key2;
        if (c < k2) { k2 = c; w2 = w1; second_is_key1 = 0; }
    }
    int ring_i = -1, idx_i = 0, ring_j = -1, idx_j = 0, idx_k = 0, valid = 0;
    float di = 1e18f, dj = 1e18f, v0[3] = {0.f, 0.f, 0.f}, v1[3] = {0.f, 0.f, 0.f}, v2[3] = {0.f, 0.f, 0.f}, n[3] = {0.f, 0.f, 0.f};
    if (w1 >= 0 && k1 < key_inf) {
        const PartialRec r = tables[(size_t)w1 * table_stride + li];
        ring_i = r.ring1; idx_i = r.idx1; idx_k = r.idx_k; di = __uint_as_float((unsigned)(k1 >> 32));
        for (int k = 0; k < 3; k++) { v0[k] = r.v0[k]; v2[k] = r.v2[k]; }
    }
    if (w2 >= 0 && k2 < key_inf) {
        const PartialRec r = tables[(size_t)w2 * table_stride + li];
        dj = __uint_as_float((unsigned)(k2 >> 32));
        if (second_is_key1) { ring_j = r.ring1; idx_j = r.idx1; for (int k = 0; k < 3; k++) v1[k] = r.v0[k]; }
        else { ring_j = r.ring2; idx_j = r.idx2; for (int k = 0; k < 3; k++) v1[k] = r.v1[k]; }
    }
    if (ring_i >= 0 && ring_j >= 0) valid = plane_from_points(v0, v1, v2, norm_cond, n);
    write_correspondence(qi, src[q_src[qi]], valid, n, v0, ring_i, idx_i, ring_j, idx_j, idx_k, di, dj, out, want_aux != 0);
}
#else
;
#endif

// Reference association search (VELO_ASSOC_VARIANT=0, kept for A/B checks): one lane per query, each lane walks the
// x-runs of its own (2 reach + 1)^3 cell neighbourhood, reach = ceil(gate radius / cell).
__global__ void __launch_bounds__(kAssocThreads)
assoc_search_kernel(PoseScalars P, GridView G, const float4* __restrict__ src, const int* __restrict__ q_src, int q_begin, int q_end,
                    const float4* __restrict__ tgt, const int* __restrict__ tgt_off, const int* __restrict__ ring_of,
                    unsigned gate_bits, double norm_cond, int reach, AssocOut out, int want_aux)
#if VELO_DEF_ASSOC
{
    const int qi = q_begin + blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = qi < q_end;
    const unsigned long long key_inf = ((unsigned long long)gate_bits + 1ull) << 32;
    float4 psrc = make_float4(0.f, 0.f, 0.f, 0.f);
    float qx = 0.f, qy = 0.f, qz = 0.f;
    unsigned long long b1 = key_inf, b2 = key_inf;
    int b1ring = -1;
    if (active) {
        psrc = src[q_src[qi]];
        transform_query(P, psrc, &qx, &qy, &qz);
        const GridDesc& g = G.d;
        const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx), cy = cell_coord(qy, g.oy, g.inv_h, g.ny), cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
        const int x0 = max(cx - reach, 0), x1 = min(cx + reach, g.nx - 1);
        const int y0 = max(cy - reach, 0), y1 = min(cy + reach, g.ny - 1);
        const int z0 = max(cz - reach, 0), z1 = min(cz + reach, g.nz - 1);
        if (x0 <= x1) {
            for (int z = z0; z <= z1; z++) {
                for (int y = y0; y <= y1; y++) {
                    const int row = (z * g.ny + y);
                    const int j0 = grid_start(G, row, x0), j1 = grid_start(G, row, x1 + 1);
                    for (int j = j0; j < j1; j++) {
                        const float4 sp = G.sorted[j];
                        const float d2 = dist2_f(qx, qy, qz, sp.x, sp.y, sp.z);
                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(sp.w);
                        if (key < b2) {                                  // implies d2 <= gate (keys start at key_inf)
                            const int ring = G.sring[j];
                            if (key < b1) {
                                if (ring != b1ring) b2 = b1;
                                b1 = key; b1ring = ring;
                            } else if (ring != b1ring) {
                                b2 = key;
                            }
                        }
                    }
                }
            }
        }
    }
    if (active) finish_correspondence(qi, psrc, qx, qy, qz, b1, b2, key_inf, tgt, tgt_off, ring_of, norm_cond, out, want_aux != 0);
}
#else
;
#endif

// ---- running (best1, best2) over distinct rings ---------------------------------------------------------------------
struct Top2 {
    unsigned long long b1, b2;
    int b1ring, b2ring; // b2ring is dead (and removed by the compiler) in the kernels that look the second ring up at the end
    float b2d;          // float view of b2's distance field: the cheap per-candidate reject threshold
};
__device__ __forceinline__ void top2_update(Top2& t, unsigned long long key, int ring) {
    if (key < t.b2) {
        if (key < t.b1) {
            if (ring != t.b1ring) { t.b2 = t.b1; t.b2ring = t.b1ring; }
            t.b1 = key; t.b1ring = ring;
        } else if (ring != t.b1ring) {
            t.b2 = key; t.b2ring = ring;
        }
        t.b2d = __uint_as_float((unsigned)(t.b2 >> 32));
    }
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

// ---- association search: LDS-staged shrinking-radius box walk (VELO_ASSOC_VARIANT=4; superseded by the tube kernel below, ----
// ---- kept for A/B and as the second independent implementation the parity tests compare against) ---------------------------
// A workgroup of NW waves owns 64 consecutive queries; lane i of EVERY wave holds query i (source scans are ring-ordered,
// so the 64 points are spatial neighbours).  The waves cluster the queries (cells within +-W of a seed lane's cell) and
// search ONE fine grid (cell ~ the smallest gate radius, shared by all outer iterations) in growing boxes around the
// cluster's cell bounding box: expansion e = 1, 2, 4, ... cells.  Per phase
//   1. one thread per grid row of the new shell reads the row's cell offsets (vector loads) -> list of candidate runs,
//      workgroup prefix sum of their lengths;
//   2. the runs are copied into an LDS tile by ALL threads with coalesced 16-byte loads (every thread finds its source
//      run by binary search in the LDS prefix array) -- hundreds of loads in flight instead of a dependent chain;
//   3. each wave sweeps a slice of the tile: the candidate is read from LDS with one broadcast ds_read_b128 and tested by
//      all 64 lanes (queries) at once;
//   4. the waves merge their per-lane (best1, best2) through LDS (top-2-distinct-rings is associative and idempotent).
// After a phase that covered expansion e every unvisited point is separated from every member query by >= e whole cells,
// so the walk stops once (e * cell)^2 > max over member lanes of b2d (second-best squared distance, or the gate when a
// lane has no second ring yet): no unvisited point can enter any lane's result.  In the dense part of a scan this ends
// after the first phase; the answer is still the exact exhaustive one.
#ifndef VELO_ASSOC_SYNC_LEAN
#define VELO_ASSOC_SYNC_LEAN 1
#endif
constexpr int kTileCap = 512;      // candidates per LDS tile (keeps the workgroup under 20 KB of LDS: 8 workgroups per CU)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// DBG = true compiles the diagnostic hooks selected at run time by `dbg` (bit 0/1/2: skip sweep / tiles / rows, 3: cycle stamps,
// 4: counters, 5: workgroup times, bits 8+: first expansion); the product instantiation has none of them -- they cost 56 spilled
// SGPRs and 9 spilled VGPRs when left in.
template <int NW, int MINW, bool DBG>
__global__ void __launch_bounds__(NW * 64, MINW)
assoc_search_v3_kernel(PoseScalars P, GridView G, const float4* __restrict__ src, const int* __restrict__ q_src, int q_begin, int q_end,
                       const float4* __restrict__ tgt, const int* __restrict__ tgt_off, const int* __restrict__ ring_of,
                       unsigned gate_bits, double norm_cond, int cluster_w, float h_safe, AssocOut out, int want_aux, int dbg, int xcd_map) {
    constexpr int NT = NW * 64;
    constexpr int NRUN = 2 * NT;                       // two run slots per row, NT rows per row chunk
    // tile: candidates stored as PAIRS -- {x0,x1,y0,y1} and {z0,z1,bits(g0),bits(g1)} -- so that the sweep handles two
    // candidates per packed-f32 instruction after two broadcast ds_read_b128
    __shared__ float4 s_xy[kTileCap / 2];
    __shared__ float4 s_zg[kTileCap / 2];
    __shared__ int s_ring[kTileCap];
    __shared__ int s_run_j0[NRUN];
    __shared__ int s_run_off[NRUN + 1];
    __shared__ int s_wave_tot[NW];
    __shared__ unsigned long long m1[NW][64], m2[NW][64];
    __shared__ int mr[NW][64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // diagnostic build only (DBG && (dbg & 8)): per-section cycle totals of wave 0, added to out.dbg[0..7]
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = (DBG && (dbg & 8)) ? (long long)__builtin_readcyclecounter() : 0;
#define VELO_STAMP(k) do { if (DBG && (dbg & 8)) { const long long now__ = (long long)__builtin_readcyclecounter(); tacc[k] += now__ - tlast; tlast = now__; } } while (0)
    // XCD-aware group mapping: workgroups are dealt round-robin over the 8 XCDs, so blockIdx % 8 selects the XCD; give
    // each XCD one CONTIGUOUS eighth of the ring-ordered groups -- spatial neighbours then share that XCD's 4 MB L2
    // (cell table rows and candidate cells are re-read by adjacent groups).  Placement affects speed only.
    const int n_groups = (q_end - q_begin + 63) >> 6;
    const int per_xcd = (n_groups + 7) >> 3;
    const int group = xcd_map ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (group >= n_groups || (xcd_map && (int)(blockIdx.x >> 3) >= per_xcd)) return;
    if ((DBG && (dbg & 32)) && threadIdx.x == 0) out.wg_times[2 * group] = __builtin_amdgcn_s_memrealtime();
    const int qi = q_begin + group * 64 + lane;
    const bool active = qi < q_end;
    const unsigned long long key_inf = ((unsigned long long)gate_bits + 1ull) << 32;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    Top2 t;
    t.b1 = key_inf; t.b2 = key_inf; t.b1ring = -1; t.b2ring = -1; t.b2d = __uint_as_float(gate_bits + 1u);
    const GridDesc g = G.d;
    int cx = 0, cy = 0, cz = 0;
    if (active) {
        const float4 psrc = src[q_src[qi]];                           // re-read at the end instead of living in 4 registers
        transform_query(P, psrc, &qx, &qy, &qz);
        cx = cell_coord(qx, g.ox, g.inv_h, g.nx); cy = cell_coord(qy, g.oy, g.inv_h, g.ny); cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
    }
    VELO_STAMP(0);
    const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
    float* s_xy_f = reinterpret_cast<float*>(s_xy);
    float* s_zg_f = reinterpret_cast<float*>(s_zg);
    bool pending = active;
    for (;;) {                                                         // clusters (identical control flow in every wave)
        const unsigned long long pm = __ballot(pending);
        if (pm == 0ull) break;
        const int leader = (int)__ffsll((long long)pm) - 1;
        const int scx = __builtin_amdgcn_readlane(cx, leader), scy = __builtin_amdgcn_readlane(cy, leader), scz = __builtin_amdgcn_readlane(cz, leader);
        const bool member = pending && abs(cx - scx) <= cluster_w && abs(cy - scy) <= cluster_w && abs(cz - scz) <= cluster_w;
        const int big = 1 << 28;
        const int bx0 = __builtin_amdgcn_readfirstlane(wave_min_i(member ? cx : big)), bx1 = __builtin_amdgcn_readfirstlane(wave_max_i(member ? cx : -big));
        const int by0 = __builtin_amdgcn_readfirstlane(wave_min_i(member ? cy : big)), by1 = __builtin_amdgcn_readfirstlane(wave_max_i(member ? cy : -big));
        const int bz0 = __builtin_amdgcn_readfirstlane(wave_min_i(member ? cz : big)), bz1 = __builtin_amdgcn_readfirstlane(wave_max_i(member ? cz : -big));
        VELO_STAMP(1);
        if ((DBG && (dbg & 16)) && tid == 0) atomicAdd(&out.dbg[0], 1ull);
        int e_prev = -1;                                               // expansion already covered (-1: nothing yet)
        int e = DBG ? max(1, dbg >> 8) : 1;                                     // first expansion (tuning knob, default 1)
        for (;;) {                                                     // phases: e = 1, then (if needed) the reach the bounds demand
            // box of this phase (clipped) and of the previous one (unclipped; empty when e_prev < 0)
            const int X0 = max(bx0 - e, 0), X1 = min(bx1 + e, g.nx - 1);
            const int Y0 = max(by0 - e, 0), Y1 = min(by1 + e, g.ny - 1);
            const int Z0 = max(bz0 - e, 0), Z1 = min(bz1 + e, g.nz - 1);
            const int px0 = bx0 - e_prev, px1 = bx1 + e_prev, py0 = by0 - e_prev, py1 = by1 + e_prev, pz0 = bz0 - e_prev, pz1 = bz1 + e_prev;
            const int nyb = Y1 - Y0 + 1, nzb = Z1 - Z0 + 1;
            const int nrows = (X0 <= X1 && nyb > 0 && nzb > 0) ? nyb * nzb : 0;
            for (int rbase = 0; rbase < ((DBG && (dbg & 4)) ? 0 : nrows); rbase += NT) {          // row chunks (one row per thread)
                // ---- 1. run list ----
                int ja0 = 0, la = 0, jb0 = 0, lb = 0;
                const int r = rbase + tid;
                if (r < nrows) {
                    const int y = Y0 + r % nyb, z = Z0 + r / nyb;
                    const int row = (z * g.ny + y);
                    const bool fresh = e_prev < 0 || y < py0 || y > py1 || z < pz0 || z > pz1;
                    if (fresh) {
                        ja0 = grid_start(G, row, X0); la = grid_start(G, row, X1 + 1) - ja0;
                    } else {                                            // old row: only the cells left of px0 and right of px1 are new
                        const int a1 = min(px0 - 1, X1), b0 = max(px1 + 1, X0);
                        if (X0 <= a1) { ja0 = grid_start(G, row, X0); la = grid_start(G, row, a1 + 1) - ja0; }
                        if (b0 <= X1) { jb0 = grid_start(G, row, b0); lb = grid_start(G, row, X1 + 1) - jb0; }
                    }
                }
                // workgroup exclusive scan of (la + lb)
                const int mine = la + lb;
                int inc = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
                if (lane == 63) s_wave_tot[wid] = inc;
                __syncthreads();
                int wbase = 0, total = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) { const int v = s_wave_tot[w]; if (w < wid) wbase += v; total += v; }
                const int ex = wbase + inc - mine;
                s_run_j0[2 * tid] = ja0; s_run_off[2 * tid] = ex;
                s_run_j0[2 * tid + 1] = jb0; s_run_off[2 * tid + 1] = ex + la;
                if (tid == 0) s_run_off[NRUN] = total;
                __syncthreads();
                VELO_STAMP(2);
                if ((DBG && (dbg & 16)) && tid == 0) { atomicAdd(&out.dbg[1], 1ull); atomicAdd(&out.dbg[2], (unsigned long long)total); if (e_prev >= 0) atomicAdd(&out.dbg[3], (unsigned long long)total); atomicAdd(&out.dbg[4], (unsigned long long)nrows); atomicAdd(&out.dbg[5], (unsigned long long)e); }
                if (DBG && (dbg & 2)) total = 0;
                // ---- 2./3. tiles ----
                for (int tbase = 0; tbase < total; tbase += kTileCap) {
                    const int tn = min(total - tbase, kTileCap);
                    const int tn2 = (tn + 1) & ~1;                     // the sweep consumes pairs
                    for (int i = tid; i < tn2; i += NT) {
                        float4 c = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __int_as_float(0x7fffffff));
                        int cr = 0x7fffffff;
                        if (i < tn) {
                            const int slot = tbase + i;
                            int lo = 0;                                // largest k with s_run_off[k] <= slot (NRUN is a power of two)
#pragma unroll
                            for (int step = NRUN / 2; step > 0; step >>= 1) {
                                if (s_run_off[lo + step] <= slot) lo += step;
                            }
                            const int j = s_run_j0[lo] + (slot - s_run_off[lo]);
                            c = G.sorted[j];
                            cr = G.sring[j];
                        }
                        const int pr = i >> 1, hb = i & 1;
                        s_xy_f[4 * pr + hb] = c.x; s_xy_f[4 * pr + 2 + hb] = c.y;
                        s_zg_f[4 * pr + hb] = c.z; s_zg_f[4 * pr + 2 + hb] = c.w;
                        s_ring[i] = cr;
                    }
                    __syncthreads();
                    VELO_STAMP(3);
                    // each wave sweeps a contiguous slice of the tile's pairs for all 64 queries
                    const int npairs = tn2 >> 1;
                    const int per = (npairs + NW - 1) / NW;
                    const int p0 = wid * per, p1 = min(p0 + per, npairs);
                    if (member && !(DBG && (dbg & 1))) {
                        // 4 pairs (8 candidates) per trip: all LDS reads first, then the packed distance math, then the
                        // (rare) updates -- keeps 8 ds_read_b128 in flight instead of one dependent read per pair
                        int pi = p0;
                        for (; pi + 4 <= p1; pi += 4) {
                            float4 a[4], bq[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) { a[u] = s_xy[pi + u]; bq[u] = s_zg[pi + u]; }
                            f32x2 d2[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const f32x2 cxp = {a[u].x, a[u].y}, cyp = {a[u].z, a[u].w}, czp = {bq[u].x, bq[u].y};
                                const f32x2 dx = qx2 - cxp, dy = qy2 - cyp, dz = qz2 - czp;
                                f32x2 d = dx * dx;                     // x -> y -> z accumulation, no FMA (-ffp-contract=off)
                                d = d + dy * dy;
                                d = d + dz * dz;
                                d2[u] = d;
                            }
                            float dmin = fminf(fminf(fminf(d2[0].x, d2[0].y), fminf(d2[1].x, d2[1].y)), fminf(fminf(d2[2].x, d2[2].y), fminf(d2[3].x, d2[3].y)));
                            if (dmin <= t.b2d) {                       // some candidate of the 8 may matter for this lane
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    if (d2[u].x <= t.b2d) {
                                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2[u].x) << 32) | (unsigned)__float_as_int(bq[u].z);
                                        top2_update(t, key, s_ring[2 * (pi + u)]);
                                    }
                                    if (d2[u].y <= t.b2d) {
                                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2[u].y) << 32) | (unsigned)__float_as_int(bq[u].w);
                                        top2_update(t, key, s_ring[2 * (pi + u) + 1]);
                                    }
                                }
                            }
                        }
                        for (; pi < p1; pi++) {
                            const float4 a = s_xy[pi], bq = s_zg[pi];
                            const f32x2 cxp = {a.x, a.y}, cyp = {a.z, a.w}, czp = {bq.x, bq.y};
                            const f32x2 dx = qx2 - cxp, dy = qy2 - cyp, dz = qz2 - czp;
                            f32x2 d2 = dx * dx;
                            d2 = d2 + dy * dy;
                            d2 = d2 + dz * dz;
                            if (d2.x <= t.b2d) {
                                const unsigned long long key = ((unsigned long long)__float_as_uint(d2.x) << 32) | (unsigned)__float_as_int(bq.z);
                                top2_update(t, key, s_ring[2 * pi]);
                            }
                            if (d2.y <= t.b2d) {
                                const unsigned long long key = ((unsigned long long)__float_as_uint(d2.y) << 32) | (unsigned)__float_as_int(bq.w);
                                top2_update(t, key, s_ring[2 * pi + 1]);
                            }
                        }
                    }
                    VELO_STAMP(4);
                    __syncthreads();
                    VELO_STAMP(5);
                }
            }
            // ---- 4. merge across waves ----
            if (NW > 1) {
                m1[wid][lane] = t.b1; m2[wid][lane] = t.b2; mr[wid][lane] = t.b1ring;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    if (w == wid) continue;
                    const unsigned long long c1 = m1[w][lane], c2 = m2[w][lane];
                    if (c1 < t.b2) top2_update(t, c1, mr[w][lane]);
                    if (c2 < t.b2) top2_update(t, c2, ring_of[(int)(unsigned)(c2 & 0xffffffffull) - out.first_point]);
                }
                __syncthreads();
            }
            // ---- stop test (identical in every wave: all hold the same merged state) ----
            // Every unvisited point is separated from every member query by >= e whole cells.  b2d only shrinks, so the
            // radius the CURRENT bounds allow is enough for one more phase to finish the cluster.
            const float rw = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(wave_max_i(member ? (int)__float_as_uint(t.b2d) : 0)));
            const float reach = (float)e * h_safe;
            VELO_STAMP(6);
            if (reach * reach > rw) break;
            e_prev = e;
            e = max(e + 1, (int)ceilf(sqrtf(rw) / h_safe));
            if ((float)e * h_safe * ((float)e * h_safe) <= rw) e++;     // rounding guard: the next test must pass
        }
        pending = pending && !member;
    }
    if (NW > 1 && wid != 0) return;
    if (active) finish_correspondence(qi, src[q_src[qi]], qx, qy, qz, t.b1, t.b2, key_inf, tgt, tgt_off, ring_of, norm_cond, out, want_aux != 0);
    VELO_STAMP(7);
    if ((DBG && (dbg & 32)) && tid == 0) out.wg_times[2 * group + 1] = __builtin_amdgcn_s_memrealtime();
    if ((DBG && (dbg & 8)) && tid == 0) { for (int k = 0; k < 8; k++) atomicAdd((unsigned long long*)&out.dbg[k], (unsigned long long)tacc[k]); }
#undef VELO_STAMP
}

// Cells that can hold a point closer than r to the query: [cell(q - r), cell(q + r)] per axis (the cell function is monotone);
// clip = additionally restricted to the query's own cell +- 1 (what phase 1 of the tube kernel visits).  Cheap enough (6 cell
// computations) to be recomputed where it is needed instead of living in 6 registers across the sweep.
struct CellBox { int x0, x1, y0, y1, z0, z1; };
__device__ __forceinline__ CellBox query_box(const GridDesc& g, float qx, float qy, float qz, int cx, int cy, int cz, float r, bool clip) {
    asm volatile("" : "+v"(r));     // opaque to CSE / loop-invariant hoisting: recomputing IS the point (6 registers less across the sweep)
    CellBox b;
    b.x0 = cell_coord(qx - r, g.ox, g.inv_h, g.nx); b.x1 = cell_coord(qx + r, g.ox, g.inv_h, g.nx);
    b.y0 = cell_coord(qy - r, g.oy, g.inv_h, g.ny); b.y1 = cell_coord(qy + r, g.oy, g.inv_h, g.ny);
    b.z0 = cell_coord(qz - r, g.oz, g.inv_h, g.nz); b.z1 = cell_coord(qz + r, g.oz, g.inv_h, g.nz);
    if (clip) {
        b.x0 = max(b.x0, cx - 1); b.x1 = min(b.x1, cx + 1);
        b.y0 = max(b.y0, cy - 1); b.y1 = min(b.y1, cy + 1);
        b.z0 = max(b.z0, cz - 1); b.z1 = min(b.z1, cz + 1);
    }
    return b;
}

__device__ __forceinline__ void top2_merge_xor(Top2& t, int mask) {
    const unsigned long long o1 = __shfl_xor(t.b1, mask), o2 = __shfl_xor(t.b2, mask);
    const int r1 = __shfl_xor(t.b1ring, mask), r2 = __shfl_xor(t.b2ring, mask);
    if (o1 < t.b2) top2_update(t, o1, r1);
    if (o2 < t.b2) top2_update(t, o2, r2);
}

// ---- association search, tube variant (VELO_ASSOC_VARIANT=5) ---------------------------------------------------------------
// Same machinery as assoc_search_v3_kernel (run list -> LDS tile -> packed sweep -> merge), different candidate set:
//   * phase 1 does not stage the whole bounding box of the cluster but, for every grid row (y, z), only the x-interval
//     [min cx - 1, max cx + 1] over the member queries whose cell lies within one row of it -- a tube around the ring segment
//     instead of its axis-aligned box (LDS atomics build the per-row intervals);
//   * phase 2 is per query: a member whose current bound reaches beyond what phase 1 visited for it asks for the cell box of
//     ITS OWN bound sphere and contributes the rows/intervals of that box only; finished members ask for nothing.  Rows
//     subtract the interval phase 1 already visited.  After phase 2 every member has seen its whole bound sphere, so there
//     is never a third phase.
// Tubes do not blow up with the length of the segment, so the cluster radius can be large (one cluster per group).
//   * rounds after the first are warm-started from the previous round's winners (AssocOut::prev).
#define kXcdChunks (reinterpret_cast<const int*>(8))
#define kXcdTiles (reinterpret_cast<const int*>(16))
template <int NW, int MINW, bool DBG, int PPT, int ASKER>
__device__ __forceinline__ void
assoc_search_v5_body(const PoseScalars& P_in, const PoseRecord* __restrict__ P_dev, int* __restrict__ chain_fail, const GridView& G, const float4* __restrict__ qpts, int q_begin, int q_end,
                     const float4* __restrict__ tgt_pad, const int* __restrict__ tgt_off,
                     unsigned gate_bits, double norm_cond, int cluster_w, float h_safe, const AssocOut& out, int want_aux, const int* __restrict__ group_perm, int dbg, int asker_rows,
                     const int block_x) {
    constexpr int NT = NW * 64;
    constexpr int NRUN = 2 * NT;
    static_assert(kTileCap % (2 * PPT) == 0, "the tile must hold whole trips (the padding of the last trip stays inside it)");
    __shared__ float4 s_xy[kTileCap / 2];
    __shared__ float4 s_zg[kTileCap / 2];
    __shared__ int s_ring[kTileCap];
    __shared__ int s_run_j0[NRUN];
    __shared__ int s_run_off[NRUN + 1];
    __shared__ int s_wave_tot[NW];
    __shared__ int s_lo[NT], s_hi[NT], s_plo[NT], s_phi[NT];   // per-row x-interval of this phase / of what phase 1 visited
    __shared__ int s_box[2][6][64];                            // per-query cell box of phase 1 / phase 2 (x0, x1, y0, y1, z0, z1)
    __shared__ int s_phase[9];                                 // Y0, Y1, Z0, Z1, asking-lane mask (lo, hi), total rows asked for, bounds beyond one cell, bounds beyond four cells
    // The transformed queries live HERE, not in registers: lane i of every wave re-reads query i where it needs it (cell boxes, the
    // rare exact-distance path of the sweep, the finish).  Three registers less across the sweep is what keeps the kernel at
    // 96 VGPRs -- five waves per SIMD -- without a single spilled register (a kernel that touches scratch pays ~11 us per launch).
    __shared__ float s_q[3][64];
#define VELO_Q(x, y, z) const float x = s_q[0][lane], y = s_q[1][lane], z = s_q[2][lane]
    // merge scratch aliases the tile: the barrier that closes the last sweep separates the two uses (NW <= 4)
    static_assert(NW * 64 * 16 <= (int)sizeof(float4) * (kTileCap / 2) && NW * 64 * 8 <= (int)sizeof(int) * kTileCap, "merge scratch must fit the tile");
    unsigned long long (*m1)[64] = reinterpret_cast<unsigned long long (*)[64]>(s_xy);
    unsigned long long (*m2)[64] = m1 + NW;
    int (*mr)[64] = reinterpret_cast<int (*)[64]>(s_ring);
    int (*mr2)[64] = mr + NW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (rotating the wave that runs the single-wave sections -- set-up, cell boxes, finish -- over the SIMDs of the CU: measured, no change)
    constexpr int lead = 0;
    // chain mode: the pose scalars come from the device record of the solve that ran just ahead of this launch
    if (chain_fail && *chain_fail) return;
    if (P_dev && !P_dev->ready) { if (block_x == 0 && tid == 0) *chain_fail = 1; return; }
    const PoseScalars& P = P_dev ? P_dev->P : P_in;
    // diagnostic instantiation only (DBG && dbg & 8): per-section cycle totals of wave 0, added to out.dbg[0..7]
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = (DBG && (dbg & 8)) ? (long long)__builtin_readcyclecounter() : 0;
#define VELO_STAMP(k) do { if (DBG && (dbg & 8)) { const long long now__ = (long long)__builtin_readcyclecounter(); tacc[k] += now__ - tlast; tlast = now__; } } while (0)
    // workgroup -> group through the host-built table (XCD-aware wedges, see build_group_perm); placement affects speed only.
    // group_perm == kXcdChunks (a tag, not a pointer): workgroups are dealt round-robin over the 8 XCDs, so block 8 j + k goes to the j-th
    // group of the k-th EIGHTH of the list -- every XCD's L2 then sees one contiguous eighth of the queries (in patch order: one band of
    // rings) and the part of the target they look at, instead of the whole scene.  The grid has 8 * ceil(groups / 8) blocks.
    int group = (int)block_x;
    if (group_perm == kXcdChunks) {
        const int n_groups = (q_end - q_begin + 63) / 64, per = (n_groups + 7) / 8;
        group = (block_x & 7) * per + (block_x >> 3);
        if (group >= n_groups) return;
    } else if (group_perm == kXcdTiles) {
        // kXcdTiles: the list cut into 64 tiles (in patch order: 8 bands of rings x 8 stretches of azimuth); XCD k takes the tiles (band b,
        // stretch (k - b) mod 8) -- one compact piece of every band, an eighth of the scene in all, and every XCD the same mix of cheap and
        // dear bands (kXcdChunks gives an XCD ONE band: local, but the bands cost differently).  The grid has 64 * ceil(groups / 64) blocks.
        const int n_groups = (q_end - q_begin + 63) / 64, per = (n_groups + 63) / 64;
        const int k = block_x & 7, j = block_x >> 3, b = j / per, w = j - b * per;
        group = (b * 8 + ((k + 8 - (b & 7)) & 7)) * per + w;
        if (b >= 8 || group >= n_groups) return;
    } else if (group_perm) group = group_perm[block_x];
    if (out.n_valid_next && block_x == 0 && tid == 0) *out.n_valid_next = 0;   // its last reader ran before this launch (same stream)
    if (ASKER == 2 && block_x == 0 && tid == 0) *out.ask_count_next = 0;
    if (DBG && out.wg_times && tid == 0) out.wg_times[2 * group] = __builtin_amdgcn_s_memrealtime();
    const int qi = q_begin + group * 64 + lane;
    const bool active = qi < q_end;
    const unsigned long long key_inf = ((unsigned long long)gate_bits + 1ull) << 32;
    Top2 t;
    t.b1 = key_inf; t.b2 = key_inf; t.b1ring = -1; t.b2ring = -1; t.b2d = __uint_as_float(gate_bits + 1u);
    const GridDesc g = G.d;
    // Set-up once per GROUP, not once per wave: wave 0 transforms the 64 queries (double precision) and enters the warm-start
    // seeds, the other waves pick the result up from LDS (scratch aliases the tile, which is not in use yet).
#if VELO_ASSOC_SYNC_LEAN
    // Round 5: EVERY wave sets itself up -- the same coalesced loads (three of the four waves hit the L2 lines the first one pulled), the
    // same arithmetic, hence the same state in every wave -- instead of wave 0 computing and handing over through LDS behind two workgroup
    // barriers.  A barrier costs this kernel ~1 us (four waves, each sharing its SIMD with four other workgroups' waves); the redundant
    // ~100 double-precision instructions per wave cost far less.  s_q receives identical values from all four waves, and a lane only
    // reads back the slot it wrote itself.
    {
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (active) {
            const float4 psrc = qpts[qi];
            const bool seeded = out.prev_a && !(DBG && (dbg & 512));
            float4 sa = make_float4(0.f, 0.f, 0.f, __int_as_float(-1)), sb = sa;
            int2 sr = make_int2(-1, -1);
            if (seeded) { sa = out.prev_a[qi]; sb = out.prev_b[qi]; sr = out.prev_r[qi]; }
            transform_query(P, psrc, &qx, &qy, &qz);
            if (__float_as_int(sa.w) >= 0) {
                const float d = dist2_f(sa.x, sa.y, sa.z, qx, qy, qz);
                if (__float_as_uint(d) <= gate_bits) top2_update(t, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(__float_as_int(sa.w) + out.first_point), sr.x);
            }
            if (__float_as_int(sb.w) >= 0) {
                const float d = dist2_f(sb.x, sb.y, sb.z, qx, qy, qz);
                if (__float_as_uint(d) <= gate_bits) top2_update(t, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(__float_as_int(sb.w) + out.first_point), sr.y);
            }
        }
        s_q[0][lane] = qx; s_q[1][lane] = qy; s_q[2][lane] = qz;
        // the per-row interval slots start out empty; from then on the thread that consumes a row's interval empties it again (see the run list)
        s_lo[tid] = 1 << 28; s_hi[tid] = -(1 << 28); s_plo[tid] = 1 << 28; s_phi[tid] = -(1 << 28);
    }
#else
    {
        unsigned long long* sb = reinterpret_cast<unsigned long long*>(s_zg);   // [2][64] best1 / best2 keys
        int* sr = s_ring;                                                        // [2][64] their rings
        if (wid == lead) {
            float qx = 0.f, qy = 0.f, qz = 0.f;
            if (active) {
                // everything the set-up needs is requested at once (coalesced, no load behind another load's result)
                const float4 psrc = qpts[qi];
                const bool seeded = out.prev_a && !(DBG && (dbg & 512));
                float4 sa = make_float4(0.f, 0.f, 0.f, __int_as_float(-1)), sb = sa;
                int2 sr = make_int2(-1, -1);
                if (seeded) { sa = out.prev_a[qi]; sb = out.prev_b[qi]; sr = out.prev_r[qi]; }
                transform_query(P, psrc, &qx, &qy, &qz);
                // Warm start: the two winners of the previous round (same source, same target, slightly different pose) are real
                // candidates of this round, so entering them first changes nothing in the result (top-2 is idempotent) but starts the
                // search with a tight second-best bound: almost every later candidate fails the cheap trip test and the per-query
                // second phase is rarely needed.  Seeds beyond the current gate are dropped like any other candidate.
                if (__float_as_int(sa.w) >= 0) {
                    const float d = dist2_f(sa.x, sa.y, sa.z, qx, qy, qz);
                    if (__float_as_uint(d) <= gate_bits) top2_update(t, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(__float_as_int(sa.w) + out.first_point), sr.x);
                }
                if (__float_as_int(sb.w) >= 0) {
                    const float d = dist2_f(sb.x, sb.y, sb.z, qx, qy, qz);
                    if (__float_as_uint(d) <= gate_bits) top2_update(t, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(__float_as_int(sb.w) + out.first_point), sr.y);
                }
            }
            s_q[0][lane] = qx; s_q[1][lane] = qy; s_q[2][lane] = qz;
            sb[lane] = t.b1; sb[64 + lane] = t.b2; sr[lane] = t.b1ring; sr[64 + lane] = t.b2ring;
        }
        __syncthreads();
        if (wid != lead) {
            t.b1 = sb[lane]; t.b2 = sb[64 + lane]; t.b1ring = sr[lane]; t.b2ring = sr[64 + lane];
            t.b2d = __uint_as_float((unsigned)(t.b2 >> 32));
        }
        __syncthreads();                                                         // the tile may be overwritten from here on
    }
#endif
    VELO_STAMP(0);
    float* s_xy_f = reinterpret_cast<float*>(s_xy);
    float* s_zg_f = reinterpret_cast<float*>(s_zg);
    const int big = 1 << 28;
    bool pending = active;
    bool deferred = false;                                             // this lane's query went on the asker list: assoc_asker_kernel finishes it
    unsigned long long gstat[6] = {0, 0, 0, 0, 0, 0};                   // diagnostics (dbg & 32): clusters, chunks, candidates, askers, asker candidates, asker time
    for (;;) {                                                         // clusters (identical control flow in every wave)
        const unsigned long long pm = __ballot(pending);
        if (pm == 0ull) break;
        const int leader = (int)__ffsll((long long)pm) - 1;
        bool member;
        {
            VELO_Q(qx, qy, qz);
            int cx = 0, cy = 0, cz = 0;
            if (active) { cx = cell_coord(qx, g.ox, g.inv_h, g.nx); cy = cell_coord(qy, g.oy, g.inv_h, g.ny); cz = cell_coord(qz, g.oz, g.inv_h, g.nz); }
            const int scx = __builtin_amdgcn_readlane(cx, leader), scy = __builtin_amdgcn_readlane(cy, leader), scz = __builtin_amdgcn_readlane(cz, leader);
            member = pending && abs(cx - scx) <= cluster_w && abs(cy - scy) <= cluster_w && abs(cz - scz) <= cluster_w;
        }
        if (DBG && (dbg & 16) && tid == 0) atomicAdd(&out.dbg[0], 1ull);
        if (DBG && (dbg & 32)) gstat[0]++;
        // Per-query cell boxes.  Everything closer to the query than r = sqrt(b2d) lies in the cells [cell(q - r), cell(q + r)] per
        // axis (the cell function is monotone; r is padded against rounding).  Phase 1 visits that box clipped to the query's
        // own cell +- 1: with warm-start seeds r is a few centimetres and the box is 1-2 cells per axis instead of 3, without
        // seeds it is the +-1 neighbourhood.  A query is finished when the box of its CURRENT bound lies inside what phase 1
        // visited for it; the others ask phase 2 for the box of their current bound.
        const float r1 = sqrtf(t.b2d) * 1.0001f + 1e-6f;               // bound radius phase 1 works with (padded against rounding)
        bool asker_phase = false;
        // Dense targets only (ASKER instantiations): a group whose phase-1 boxes span more than `dense_rows` grid rows -- queries strung
        // along a wall that thirty scans have sampled -- would stage the union of 64 nearly disjoint little boxes (12,000-17,000 candidates)
        // and make every lane test all of them.  Such a group skips the tile machinery: every member goes query by query (all 64 lanes
        // of a wave on one query's own bound sphere), the path the asking queries of phase 2 take anyway.  Exact: a query searched that
        // way sees its whole bound sphere.  (dense_rows rides in the `dbg` argument, which the product instantiations do not use.)
        // ... and only when at most `dense_far` members have a bound that reaches beyond FOUR cells (a cold round, or one behind a solve that
        // moved the pose by decimetres, has big spheres: there the union tube is the cheaper way).  dbg = dense_rows | dense_far << 20.
        const int dense_rows = (ASKER && !DBG) ? (dbg & 0xfffff) : 0, dense_far = (ASKER && !DBG) ? (dbg >> 20) : 0;
        bool asker_all = false;
        for (int ph = 0; ph < 2; ph++) {
            // No second phase at all when no member's bound reaches beyond one cell: the box of a radius <= h lies inside the query's own
            // cell +- 1, which is what phase 1 visited -- nobody can ask (s_phase[7] was published with phase 1's boxes).
            if (ph == 1 && s_phase[7] == 0) break;
            // Who asks for cells in this phase, with which box, and the (y, z) extent of all boxes: worked out by wave 0 only and
            // published through LDS (s_box, s_phase) -- the four waves hold identical states here, three of them would only
            // repeat ~250 VALU instructions of cell arithmetic and wave reductions per phase.
            if (wid == lead) {
                bool asks0 = member;
                CellBox bb;
                int rows_all = 0;
                VELO_Q(qx, qy, qz);
                const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx), cy = cell_coord(qy, g.oy, g.inv_h, g.ny), cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
                if (ph == 0) bb = query_box(g, qx, qy, qz, cx, cy, cz, r1, true);
                else {
                    const float rq = sqrtf(t.b2d) * 1.0001f + 1e-6f;
                    bb = query_box(g, qx, qy, qz, cx, cy, cz, rq, false);
                    asks0 = member && !(bb.x0 >= s_box[0][0][lane] && bb.x1 <= s_box[0][1][lane] && bb.y0 >= s_box[0][2][lane] &&
                                        bb.y1 <= s_box[0][3][lane] && bb.z0 >= s_box[0][4][lane] && bb.z1 <= s_box[0][5][lane]);
                    rows_all = asks0 ? (bb.y1 - bb.y0 + 1) * (bb.z1 - bb.z0 + 1) : 0;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) rows_all += __shfl_xor(rows_all, off);
                }
                s_box[ph][0][lane] = bb.x0; s_box[ph][1][lane] = bb.x1; s_box[ph][2][lane] = bb.y0;
                s_box[ph][3][lane] = bb.y1; s_box[ph][4][lane] = bb.z0; s_box[ph][5][lane] = bb.z1;
                const unsigned long long am0 = __ballot(asks0);
                const unsigned long long far0 = (ph == 0) ? __ballot(member && r1 > h_safe) : 1ull;
                const unsigned long long far4 = (ASKER && ph == 0) ? __ballot(member && r1 > 4.0f * h_safe) : 0ull;
                const int y0 = wave_min_i(asks0 ? bb.y0 : big), y1 = wave_max_i(asks0 ? bb.y1 : -big);
                const int z0 = wave_min_i(asks0 ? bb.z0 : big), z1 = wave_max_i(asks0 ? bb.z1 : -big);
                if (lane == 0) {
                    s_phase[0] = max(y0, 0); s_phase[1] = min(y1, g.ny - 1); s_phase[2] = max(z0, 0); s_phase[3] = min(z1, g.nz - 1);
                    s_phase[4] = (int)(unsigned)(am0 & 0xffffffffull); s_phase[5] = (int)(unsigned)(am0 >> 32); s_phase[6] = rows_all;
                    if (ph == 0) s_phase[7] = (int)__popcll(far0);            // members whose bound reaches beyond one cell
                    if (ASKER && ph == 0) s_phase[8] = (int)__popcll(far4);   // ... beyond four cells (what the query-by-query search takes in its first stage)
                }
            }
            __syncthreads();
            if (ASKER && ph == 0 && dense_rows > 0) {
                const int ny0 = s_phase[1] - s_phase[0] + 1, nz0 = s_phase[3] - s_phase[2] + 1;
                if (ny0 > 0 && nz0 > 0 && ny0 * nz0 > dense_rows && s_phase[8] <= dense_far) { asker_all = true; asker_phase = true; break; }
            }
            const unsigned long long am_ph = ((unsigned long long)(unsigned)s_phase[5] << 32) | (unsigned long long)(unsigned)s_phase[4];
            const bool asks = ((am_ph >> lane) & 1ull) != 0ull;
            if (ph == 1) {
                if (am_ph == 0ull || (DBG && (dbg & 1024))) break;                   // dbg & 1024: diagnostic, no second phase (wrong results)
                // Two ways through phase 2.  Few askers with small boxes (a 120k-point scan: ~10 per group, <= 81 rows each, a few new
                // points): the row/tile machinery below takes all their rows in one parallel pass.  Many askers with big boxes (2M-point
                // map on its density-shrunk grid: every query, up to 961 rows each): the union tube makes all 64 lanes test thousands
                // of candidates that matter to one query each -- then one query at a time is cheaper.
                if (ASKER && s_phase[6] > asker_rows) { asker_phase = true; break; }
            }
            const int Y0 = s_phase[0], Y1 = s_phase[1], Z0 = s_phase[2], Z1 = s_phase[3];
            const int nyb = Y1 - Y0 + 1, nzb = Z1 - Z0 + 1;
            const int nrows = (nyb > 0 && nzb > 0) ? nyb * nzb : 0;
            const float rcp_nyb = 1.0f / (float)max(nyb, 1);
            for (int rbase = 0; rbase < ((DBG && (dbg & 4)) ? 0 : nrows); rbase += NT) {          // row chunks (one row per thread)
                // z-layers that have rows in this chunk (rows are z-major): a query only walks those layers of its box, so a box
                // that spans several chunks costs its rows once, not once per chunk
                const int zc0 = Z0 + (int)((float)rbase * rcp_nyb) - 1, zc1 = Z0 + (int)((float)(rbase + NT - 1) * rcp_nyb) + 1;
                // A chunk no asking query's box reaches into is skipped before it costs a barrier (every wave holds the same boxes, so
                // the test is workgroup-uniform).
                {
                    bool hit = asks && s_box[ph][4][lane] <= zc1 && s_box[ph][5][lane] >= zc0;
                    if (hit && nyb >= NT / 4) {                        // wide layers: a chunk is a slice of one or two of them, so test the y-range as well
                        hit = false;
                        const int ya = max(s_box[ph][2][lane], Y0) - Y0 - rbase, yb = min(s_box[ph][3][lane], Y1) - Y0 - rbase;
                        for (int z = max(max(s_box[ph][4][lane], zc0), Z0); z <= min(min(s_box[ph][5][lane], zc1), Z1); z++)
                            hit = hit || ((z - Z0) * nyb + yb >= 0 && (z - Z0) * nyb + ya < NT);
                    }
                    if (__ballot(hit) == 0ull) continue;
                }
                // ---- 0. per-row x-intervals ----
#if !VELO_ASSOC_SYNC_LEAN
                s_lo[tid] = big; s_hi[tid] = -big;
                if (ph == 1) { s_plo[tid] = big; s_phi[tid] = -big; }
                __syncthreads();
#endif
                if (asks) {                                            // the rows of this query's box: z-layers dealt over the waves
                    const int bx0 = s_box[ph][0][lane], bx1 = s_box[ph][1][lane], by0 = s_box[ph][2][lane], by1 = s_box[ph][3][lane];
                    const int bz0 = s_box[ph][4][lane], bz1 = s_box[ph][5][lane];
                    for (int z = max(bz0, zc0) + wid; z <= min(bz1, zc1); z += NW) {
                        if (z < Z0 || z > Z1) continue;
                        const int rz = (z - Z0) * nyb - Y0 - rbase;
                        for (int y = max(by0, Y0); y <= min(by1, Y1); y++) {
                            const int r = rz + y;
                            if (r >= 0 && r < NT) { atomicMin(&s_lo[r], bx0); atomicMax(&s_hi[r], bx1); }
                        }
                    }
                }
                if (ph == 1 && member) {                               // what phase 1 staged: the phase-1 boxes of ALL members
                    const int bx0 = s_box[0][0][lane], bx1 = s_box[0][1][lane], by0 = s_box[0][2][lane], by1 = s_box[0][3][lane];
                    const int bz0 = s_box[0][4][lane], bz1 = s_box[0][5][lane];
                    for (int z = max(bz0, zc0) + wid; z <= min(bz1, zc1); z += NW) {
                        if (z < Z0 || z > Z1) continue;
                        const int rz = (z - Z0) * nyb - Y0 - rbase;
                        for (int y = max(by0, Y0); y <= min(by1, Y1); y++) {
                            const int r = rz + y;
                            if (r >= 0 && r < NT) { atomicMin(&s_plo[r], bx0); atomicMax(&s_phi[r], bx1); }
                        }
                    }
                }
                __syncthreads();
                VELO_STAMP(1);
                // ---- 1. run list ----
                int ja0 = 0, la = 0, jb0 = 0, lb = 0;
                const int r = rbase + tid;
#if VELO_ASSOC_SYNC_LEAN
                const int lo_raw = s_lo[tid], hi_raw = s_hi[tid], plo_raw = s_plo[tid], phi_raw = s_phi[tid];
                s_lo[tid] = big; s_hi[tid] = -big; s_plo[tid] = big; s_phi[tid] = -big;   // emptied by their consumer: the next chunk / phase / cluster needs no barrier to start filling
                                                                                          // (its atomics come behind the two barriers of the scan below)
#else
                const int lo_raw = s_lo[tid], hi_raw = s_hi[tid], plo_raw = (ph == 1) ? s_plo[tid] : big, phi_raw = (ph == 1) ? s_phi[tid] : -big;
#endif
                if (r < nrows) {
                    const int lo = max(lo_raw, 0), hi = min(hi_raw, g.nx - 1);
                    if (lo <= hi) {
                        // r / nyb without the integer-division sequence: r < 2^24 (bounded cluster radius), one float multiply + fix-up
                        int zq = (int)((float)r * rcp_nyb), yr = r - zq * nyb;
                        if (yr < 0) { zq--; yr += nyb; } else if (yr >= nyb) { zq++; yr -= nyb; }
                        const int y = Y0 + yr, z = Z0 + zq;
                        const int row = (z * g.ny + y);
                        const int plo = (ph == 1) ? plo_raw : big, phi = (ph == 1) ? phi_raw : -big;
                        if (plo > phi) {                                // nothing of this row visited yet
                            ja0 = grid_start(G, row, lo); la = grid_start(G, row, hi + 1) - ja0;
                        } else {                                        // only the cells left of plo and right of phi are new
                            const int a1 = min(plo - 1, hi), b0 = max(phi + 1, lo);
                            if (lo <= a1) { ja0 = grid_start(G, row, lo); la = grid_start(G, row, a1 + 1) - ja0; }
                            if (b0 <= hi) { jb0 = grid_start(G, row, b0); lb = grid_start(G, row, hi + 1) - jb0; }
                        }
                    }
                }
                // workgroup exclusive scan of (la + lb)
                const int mine = la + lb;
                int inc = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
                if (lane == 63) s_wave_tot[wid] = inc;
                __syncthreads();
                int wbase = 0, total = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) { const int v = s_wave_tot[w]; if (w < wid) wbase += v; total += v; }
                const int ex = wbase + inc - mine;
                s_run_j0[2 * tid] = ja0; s_run_off[2 * tid] = ex;
                s_run_j0[2 * tid + 1] = jb0; s_run_off[2 * tid + 1] = ex + la;
                if (tid == 0) s_run_off[NRUN] = total;
                __syncthreads();
                VELO_STAMP(2);
                if (DBG && (dbg & 2)) total = 0;
                if (DBG && (dbg & 32)) { gstat[1]++; gstat[2] += (unsigned long long)total; }
                if (DBG && (dbg & 16) && tid == 0) { atomicAdd(&out.dbg[1], 1ull); atomicAdd(&out.dbg[2], (unsigned long long)total); if (ph == 1) atomicAdd(&out.dbg[3], (unsigned long long)total); atomicAdd(&out.dbg[4], (unsigned long long)nrows); }
                // ---- 2./3. tiles ----
                for (int tbase = 0; tbase < total; tbase += kTileCap) {
                    const int tn = min(total - tbase, kTileCap);
                    const int tn2 = (tn + 2 * PPT - 1) / (2 * PPT) * (2 * PPT);   // the sweep consumes trips of PPT pairs; +inf sentinels pad
                    for (int i = tid; i < tn2; i += NT) {
                        float4 c = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __int_as_float(0x7fffffff));
                        int cr = 0x7fffffff;
                        if (i < tn) {
                            const int slot = tbase + i;
                            int lo = 0;                                // largest k with s_run_off[k] <= slot (NRUN is a power of two)
#pragma unroll
                            for (int step = NRUN / 2; step > 0; step >>= 1) {
                                if (s_run_off[lo + step] <= slot) lo += step;
                            }
                            const int j = s_run_j0[lo] + (slot - s_run_off[lo]);
                            c = G.sorted[j];
                            cr = G.sring[j];
                        }
                        const int pr = i >> 1, hb = i & 1;
                        s_xy_f[4 * pr + hb] = c.x; s_xy_f[4 * pr + 2 + hb] = c.y;
                        s_zg_f[4 * pr + hb] = c.z; s_zg_f[4 * pr + 2 + hb] = c.w;
                        s_ring[i] = cr;
                    }
                    __syncthreads();
                    VELO_STAMP(3);
                    // Trips of PPT pairs (2 PPT candidates) are dealt round-robin over the waves, so every wave samples the whole
                    // tile instead of one quarter of it -- its bounds tighten as fast as the best rows allow.  Per trip: all LDS
                    // reads first (coordinates AND ring ids), then the packed distance math, then the (rare) updates.
                    const int npairs = tn2 >> 1;
                    if (member && !(DBG && (dbg & 1))) {
                        const int2* s_ring2 = reinterpret_cast<const int2*>(s_ring);
                        for (int pi = wid * PPT; pi < npairs; pi += NW * PPT) {
                            float4 a[PPT], bq[PPT];
                            int2 rg[PPT];
#pragma unroll
                            for (int u = 0; u < PPT; u++) { a[u] = s_xy[pi + u]; bq[u] = s_zg[pi + u]; rg[u] = s_ring2[pi + u]; }
                            float d2x[PPT], d2y[PPT];
                            {
                                VELO_Q(qx, qy, qz);                    // plain f32 operations (a packed-f32 instruction takes two issue slots: measured equal)
#pragma unroll
                                for (int u = 0; u < PPT; u++) {
                                    d2x[u] = dist2_f(qx, qy, qz, a[u].x, a[u].z, bq[u].x);
                                    d2y[u] = dist2_f(qx, qy, qz, a[u].y, a[u].w, bq[u].y);
                                }
                            }
                            float dmin = fminf(d2x[0], d2y[0]);
#pragma unroll
                            for (int u = 1; u < PPT; u++) dmin = fminf(dmin, fminf(d2x[u], d2y[u]));
                            if (dmin <= t.b2d) {                       // some candidate of the trip may matter for this lane
#pragma unroll
                                for (int u = 0; u < PPT; u++) {
                                    if (d2x[u] <= t.b2d) {
                                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2x[u]) << 32) | (unsigned)__float_as_int(bq[u].z);
                                        top2_update(t, key, rg[u].x);
                                    }
                                    if (d2y[u] <= t.b2d) {
                                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2y[u]) << 32) | (unsigned)__float_as_int(bq[u].w);
                                        top2_update(t, key, rg[u].y);
                                    }
                                }
                            }
                        }
                    }
                    VELO_STAMP(4);
                    __syncthreads();
                    VELO_STAMP(5);
                }
            }
            // ---- 4. merge across waves ----
            if (NW > 1) {
                m1[wid][lane] = t.b1; m2[wid][lane] = t.b2; mr[wid][lane] = t.b1ring; mr2[wid][lane] = t.b2ring;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    if (w == wid) continue;
                    const unsigned long long c1 = m1[w][lane], c2 = m2[w][lane];
                    if (c1 < t.b2) top2_update(t, c1, mr[w][lane]);
                    if (c2 < t.b2) top2_update(t, c2, mr2[w][lane]);
                }
                // (lean: the scratch aliases the tile, and every way from here to the next write into the tile -- the staging of a later phase or
                //  cluster -- passes the barrier behind a box computation and the two of a scan; the query-by-query paths write it at once)
                if (!VELO_ASSOC_SYNC_LEAN || ASKER != 0) __syncthreads();
            }
            VELO_STAMP(6);
        }
        if (ASKER == 2 && asker_phase) {     // (instantiation 2: the asking queries are deferred; 1: searched in place; 0: regular grid, no askers)
            // ---- phase 2 handed to assoc_asker_kernel: the asking queries of this cluster go on the global list with their state ----
            VELO_Q(qx, qy, qz);
            const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx), cy = cell_coord(qy, g.oy, g.inv_h, g.ny), cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
            const float rq0 = sqrtf(t.b2d) * 1.0001f + 1e-6f;
            const CellBox b2 = query_box(g, qx, qy, qz, cx, cy, cz, rq0, false);       // against the phase-1 box wave 0 left in s_box[0]
            const bool asks = member && (asker_all || !(b2.x0 >= s_box[0][0][lane] && b2.x1 <= s_box[0][1][lane] && b2.y0 >= s_box[0][2][lane] &&
                                                        b2.y1 <= s_box[0][3][lane] && b2.z0 >= s_box[0][4][lane] && b2.z1 <= s_box[0][5][lane]));
            if (wid == 0) {                                            // every wave holds the same states: one of them writes
                const unsigned long long am = __ballot(asks);
                int base = 0;
                if (lane == 0 && am != 0ull) base = atomicAdd(out.ask_count, (int)__popcll(am));
                base = __builtin_amdgcn_readfirstlane(base);
                if (asks) {
                    const int q_of_lane = q_begin + group * 64 + lane;
                    out.ask_list[base + (int)__popcll(am & ((1ull << lane) - 1ull))] = q_of_lane;
                    out.ask_keys[2 * (size_t)q_of_lane] = t.b1; out.ask_keys[2 * (size_t)q_of_lane + 1] = t.b2;
                    out.ask_rings[q_of_lane] = make_int2(t.b1ring, t.b2ring);
                }
            }
            deferred = deferred || asks;
        } else if (ASKER == 1 && asker_phase) {
            // ---- phase 2, one asking query at a time ("asker-centric") ---------------------------------------------------------------
            // After phase 1 few queries still need cells (those whose second ring is farther than a cell), each a large box of its own
            // that shares little with the others'.  Pushing them through the row/tile machinery costs ~10 barriers and a mostly
            // empty tile per chunk; instead the askers are dealt over the waves and a wave turns ALL 64 LANES on ONE query: the
            // lanes take the rows of its box (cell offsets -> runs -> wave prefix sum), then each lane tests its own share of the
            // candidates against that one query, starting from the query's current bounds (so nearly everything is pruned), and
            // six xor-shuffle merges pool the lanes' top-2 states.  Cells phase 1 already staged are simply tested again
            // (top-2 is idempotent).  No workgroup barrier until the results are handed to the other waves.
            // Two stages: first every asker looks no farther than 4 cells (a query that starts without a second ring -- bound = the
            // gate, 15 cells on the shrunk grid -- usually finds one nearby and shrinks its bound), then only the queries whose bound
            // still reaches beyond those 4 cells search their full sphere.
            const float cap = 4.0f * h_safe;
            bool asks;
            VELO_Q(qx, qy, qz);
            const int cx = cell_coord(qx, g.ox, g.inv_h, g.nx), cy = cell_coord(qy, g.oy, g.inv_h, g.ny), cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
            {
                const float rq0 = sqrtf(t.b2d) * 1.0001f + 1e-6f;
                const CellBox b2 = query_box(g, qx, qy, qz, cx, cy, cz, rq0, false), b1 = query_box(g, qx, qy, qz, cx, cy, cz, r1, true);
                asks = member && (asker_all || !(b2.x0 >= b1.x0 && b2.x1 <= b1.x1 && b2.y0 >= b1.y0 && b2.y1 <= b1.y1 && b2.z0 >= b1.z0 && b2.z1 <= b1.z1));
            }
            for (int stage = 0; stage < 2; stage++) {
            const float rnow = sqrtf(t.b2d) * 1.0001f + 1e-6f;
            const float rq = stage == 0 ? fminf(rnow, cap) : rnow;
            if (stage == 1) asks = asks && rnow > cap;                 // the others have seen their whole bound sphere in stage 0
            unsigned long long am = __ballot(asks);
            if (DBG && (dbg & 1024)) am = 0ull;
            const unsigned long long t_ask0 = (DBG && (dbg & 32)) ? __builtin_amdgcn_s_memrealtime() : 0ull;
            if (DBG && (dbg & 32)) gstat[3] += (unsigned long long)__popcll(am);
            if (am != 0ull) {
                // LPA lanes per asker (64: one asker per wave at a time; 16 -- four at a time -- measured slower on the map: 314 vs 244 us)
                constexpr int LPA = 64;
                const int sg = lane / LPA, sl = lane % LPA;
                int* w_j0 = s_run_j0 + wid * 128 + sg * (LPA + 1);     // this sub-group's run list (16 rows at a time)
                int* w_off = s_run_off + wid * 128 + sg * (LPA + 1);
                if (asks) s_lo[(int)__popcll(am & ((1ull << lane) - 1ull))] = lane;   // rank -> lane (every wave writes the same values)
                __syncthreads();
                const int n_ask = (int)__popcll(am);
                const int nq_w = (n_ask - wid + NW - 1) / NW;          // askers of this wave: ranks wid, wid + NW, ...
                for (int p0 = 0; p0 < nq_w; p0 += 64 / LPA) {
                    const int q = p0 + sg;
                    const bool on = q < nq_w;
                    const int la = on ? s_lo[q * NW + wid] : 0;
                    // the asker's query and current state, broadcast to the lanes of its sub-group
                    const float ax = __shfl(qx, la), ay = __shfl(qy, la), az = __shfl(qz, la), ar = __shfl(rq, la);
                    Top2 tl;
                    tl.b1 = __shfl(t.b1, la); tl.b2 = __shfl(t.b2, la); tl.b1ring = __shfl(t.b1ring, la); tl.b2ring = __shfl(t.b2ring, la);
                    tl.b2d = __uint_as_float((unsigned)(tl.b2 >> 32));
                    const CellBox bb = query_box(g, ax, ay, az, 0, 0, 0, ar, false);
                    const int x0 = max(bb.x0, 0), x1 = min(bb.x1, g.nx - 1);
                    const int y0 = max(bb.y0, 0), y1 = min(bb.y1, g.ny - 1), z0 = max(bb.z0, 0), z1 = min(bb.z1, g.nz - 1);
                    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
                    const int nrows_a = (on && x0 <= x1 && ny > 0 && nz > 0) ? ny * nz : 0;
                    const float rcp_ny = 1.0f / (float)max(ny, 1);
                    for (int r0 = 0; r0 < nrows_a; r0 += LPA) {
                        const int r = r0 + sl;
                        int j0 = 0, len = 0;
                        if (r < nrows_a) {
                            int zq = (int)((float)r * rcp_ny), yr = r - zq * ny;
                            if (yr < 0) { zq--; yr += ny; } else if (yr >= ny) { zq++; yr -= ny; }
                            const int row = ((z0 + zq) * g.ny + (y0 + yr));
                            j0 = grid_start(G, row, x0); len = grid_start(G, row, x1 + 1) - j0;
                        }
                        int inc = len;
#pragma unroll
                        for (int off = 1; off < LPA; off <<= 1) { const int v = __shfl_up(inc, off, LPA); if (sl >= off) inc += v; }
                        const int total = __shfl(inc, LPA - 1, LPA);
                        w_j0[sl] = j0; w_off[sl] = inc - len;
                        if (sl == 0) w_off[LPA] = total;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (DBG && (dbg & 32)) gstat[4] += (unsigned long long)total;
                        if (DBG && (dbg & 16) && sl == 0) { atomicAdd(&out.dbg[2], (unsigned long long)total); atomicAdd(&out.dbg[3], (unsigned long long)total); atomicAdd(&out.dbg[4], (unsigned long long)min(LPA, nrows_a - r0)); }
                        for (int slot = sl; slot < total; slot += LPA) {
                            int lo = 0;                                // largest i with w_off[i] <= slot
#pragma unroll
                            for (int step = LPA / 2; step > 0; step >>= 1) { if (w_off[lo + step] <= slot) lo += step; }
                            const int j = w_j0[lo] + (slot - w_off[lo]);
                            const float4 c = G.sorted[j];
                            const float d = dist2_f(ax, ay, az, c.x, c.y, c.z);
                            if (d <= tl.b2d) top2_update(tl, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(c.w), G.sring[j]);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();               // the run list is rewritten by the next rows
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
#pragma unroll
                    for (int m = 1; m < LPA; m <<= 1) top2_merge_xor(tl, m);     // the sub-group's lanes now hold the asker's result
                    if (on && sl == 0) { m1[0][la] = tl.b1; m2[0][la] = tl.b2; mr[0][la] = tl.b1ring; mr2[0][la] = tl.b2ring; }
                }
                __syncthreads();                                       // results of all waves' askers are in LDS
                if (asks) {
                    t.b1 = m1[0][lane]; t.b2 = m2[0][lane]; t.b1ring = mr[0][lane]; t.b2ring = mr2[0][lane];
                    t.b2d = __uint_as_float((unsigned)(t.b2 >> 32));
                }
                __syncthreads();                                       // the scratch aliases the tile of the next cluster
            }
            if (DBG && (dbg & 32)) gstat[5] += __builtin_amdgcn_s_memrealtime() - t_ask0;
            }
            VELO_STAMP(6);
        }
        pending = pending && !member;
        // lean: a group of several clusters -- wave 0 of the next cluster publishes s_phase / s_box while a slow wave may still be reading this
        // cluster's (the barrier the merge used to end with kept them apart)
        if (VELO_ASSOC_SYNC_LEAN && ASKER == 0 && __ballot(pending) != 0ull) __syncthreads();
    }
    if (NW > 1 && wid != lead) return;
    VELO_Q(qx, qy, qz);
    // the query index is recomputed here (opaque to the compiler) instead of living, sign-extended to 64 bits, across the whole search
    int lane_fin = lane;
    asm volatile("" : "+v"(lane_fin));
    const int qi_fin = q_begin + group * 64 + lane_fin;
    const bool active_fin = qi_fin < q_end;
#define qi qi_fin
#define active active_fin
    if (DBG && (dbg & 128)) {                                          // diagnostic: finish without the gathers (wrong results)
        if (active) { out.p[qi] = make_float4(qx, qy, qz, 0.f); out.n[qi] = make_float4((float)(t.b1 >> 32), (float)(t.b2 >> 32), 0.f, 0.f); out.v0[qi] = make_float4(0.f, 0.f, 0.f, 0.f); }
    } else
    if (active && !deferred) finish_correspondence_pad(qi, qpts, qx, qy, qz, t.b1, t.b2, t.b1ring, t.b2ring, key_inf, tgt_pad, tgt_off, norm_cond, out, want_aux != 0);
    VELO_STAMP(7);
    if (DBG && out.wg_times && tid == 0) {
        out.wg_times[2 * group + 1] = __builtin_amdgcn_s_memrealtime();
        const int n_groups = (q_end - q_begin + 63) / 64;              // the tube kernel's own counters follow the start/end stamps
        for (int k = 0; k < 6; k++) out.wg_times[2 * (size_t)n_groups + 14 * (size_t)group + k] = gstat[k];
        for (int k = 0; k < 8; k++) out.wg_times[2 * (size_t)n_groups + 14 * (size_t)group + 6 + k] = (unsigned long long)tacc[k];
    }
    if (DBG && (dbg & 8) && tid == 0) { for (int k = 0; k < 8; k++) atomicAdd((unsigned long long*)&out.dbg[k], (unsigned long long)tacc[k]); }
#undef VELO_STAMP
#undef VELO_Q
#undef qi
#undef active
}


// one launch = one context's round
template <int NW, int MINW, bool DBG, int PPT, int ASKER>
__global__ void __launch_bounds__(NW * 64, MINW)
assoc_search_v5_kernel(PoseScalars P, const PoseRecord* __restrict__ P_dev, int* __restrict__ chain_fail, GridView G, const float4* __restrict__ qpts, int q_begin, int q_end,
                       const float4* __restrict__ tgt_pad, const int* __restrict__ tgt_off,
                       unsigned gate_bits, double norm_cond, int cluster_w, float h_safe, AssocOut out, int want_aux, const int* __restrict__ group_perm, int dbg, int asker_rows) {
    assoc_search_v5_body<NW, MINW, DBG, PPT, ASKER>(P, P_dev, chain_fail, G, qpts, q_begin, q_end, tgt_pad, tgt_off, gate_bits, norm_cond, cluster_w, h_safe, out,
                                                    want_aux, group_perm, dbg, asker_rows, (int)blockIdx.x);
}

// one launch = the same round of SEVERAL contexts (velo_frame_to_frame_batch / velo_register_batch): blockIdx.y = context.  A round of
// one 120k-query scan is 1,875 workgroups on 1,280 resident slots -- one and a half waves of workgroups, the second half empty;
// four contexts in one grid fill the chip evenly (measured: 41 -> 28 us per 120k queries).  Arguments by value in the kernarg
// segment, read with scalar loads.
constexpr int kAssocBatchMax = 4;
struct AssocArgs {
    PoseScalars P; const PoseRecord* P_dev; int* chain_fail; GridView G; const float4* qpts; int q_begin, q_end;
    const float4* tgt_pad; const int* tgt_off; unsigned gate_bits; double norm_cond; int cluster_w; float h_safe;
    AssocOut out; int want_aux; const int* group_perm; int dbg; int asker_rows;
};
struct AssocBatch { AssocArgs item[kAssocBatchMax]; };
template <int NW, int MINW, bool DBG, int PPT, int ASKER>
__global__ void __launch_bounds__(NW * 64, MINW)
assoc_search_v5_batch_kernel(AssocBatch B) {
    const AssocArgs& a = B.item[blockIdx.y];
    if (a.group_perm != kXcdChunks && a.group_perm != kXcdTiles && (int)blockIdx.x * 64 >= a.q_end - a.q_begin) return;
    assoc_search_v5_body<NW, MINW, DBG, PPT, ASKER>(a.P, a.P_dev, a.chain_fail, a.G, a.qpts, a.q_begin, a.q_end, a.tgt_pad, a.tgt_off, a.gate_bits, a.norm_cond,
                                                    a.cluster_w, a.h_safe, a.out, a.want_aux, a.group_perm, a.dbg, a.asker_rows, (int)blockIdx.x);
}

// ---- the asking queries of a tube launch, searched by the whole chip ----------------------------------------------------------
// On the density-shrunk grid of a 2M-point map the queries that still need cells after phase 1 own most of a cold round's work
// (14 of 16 M candidate tests), and the heavy ones -- ~50 us of dependent gathers each -- come in clumps: groups of 64 consecutive
// queries that look into space where only one ring is near.  Searched inside their group's workgroup (16 per wave, one after the
// other) they make the launch tail-bound: workgroup duration mean 88 us, p99 466 us, max 704 us = the launch, where 1,875
// workgroups on 1,280 slots would need ~130 us if they were equal.  Here the tube launch only lists them; this kernel gives every
// wave kAskChunk of them TAKEN WITH A STRIDE (neighbours in the list go to different waves), searches one at a time with all 64
// lanes exactly as the in-place code does (rows over lanes -> wave prefix sum -> candidates over lanes -> xor-shuffle merge; first
// no farther than 4 cells, then -- if the bound still reaches beyond -- the whole sphere) and finishes its correspondences.
constexpr int kAskChunk = 8;
__device__ __forceinline__ void asker_search(const GridView& G, const GridDesc& g, float ax, float ay, float az, float rq, Top2& tl, int lane,
                                             int* __restrict__ w_j0, int* __restrict__ w_off) {
    const CellBox bb = query_box(g, ax, ay, az, 0, 0, 0, rq, false);
    const int x0 = max(bb.x0, 0), x1 = min(bb.x1, g.nx - 1);
    const int y0 = max(bb.y0, 0), y1 = min(bb.y1, g.ny - 1), z0 = max(bb.z0, 0), z1 = min(bb.z1, g.nz - 1);
    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    const int nrows_a = (x0 <= x1 && ny > 0 && nz > 0) ? ny * nz : 0;
    const float rcp_ny = 1.0f / (float)max(ny, 1);
    for (int r0 = 0; r0 < nrows_a; r0 += 64) {
        const int r = r0 + lane;
        int j0 = 0, len = 0;
        if (r < nrows_a) {
            int zq = (int)((float)r * rcp_ny), yr = r - zq * ny;
            if (yr < 0) { zq--; yr += ny; } else if (yr >= ny) { zq++; yr -= ny; }
            const int row = ((z0 + zq) * g.ny + (y0 + yr));
            j0 = grid_start(G, row, x0); len = grid_start(G, row, x1 + 1) - j0;
        }
        int inc = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
        const int total = __shfl(inc, 63);
        w_j0[lane] = j0; w_off[lane] = inc - len;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int slot = lane; slot < total; slot += 128) {             // two candidates per trip: two independent search + gather chains
            const int slot2 = slot + 64;                               // (lane-contiguous slots: a trip's gathers are coalesced; per-lane segments measured slower)
            const bool two = slot2 < total;
            int lo = 0, lo2 = 0;                                       // largest i with w_off[i] <= slot
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                if (w_off[lo + step] <= slot) lo += step;
                if (w_off[lo2 + step] <= slot2) lo2 += step;
            }
            const int j = w_j0[lo] + (slot - w_off[lo]);
            const int j2 = two ? w_j0[lo2] + (slot2 - w_off[lo2]) : j;
            const float4 c = G.sorted[j], c2 = G.sorted[j2];
            const float d = dist2_f(ax, ay, az, c.x, c.y, c.z), d2 = dist2_f(ax, ay, az, c2.x, c2.y, c2.z);
            if (d <= tl.b2d) top2_update(tl, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(c.w), G.sring[j]);
            if (two && d2 <= tl.b2d) top2_update(tl, ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(c2.w), G.sring[j2]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                               // the run list is rewritten by the next rows
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) top2_merge_xor(tl, m);            // every lane now holds the query's state
}

__device__ __forceinline__ void
assoc_asker_body(const PoseScalars& P_in, const PoseRecord* __restrict__ P_dev, const int* __restrict__ chain_fail, const GridView& G, const float4* __restrict__ qpts,
                 const float4* __restrict__ tgt_pad, const int* __restrict__ tgt_off, unsigned gate_bits, double norm_cond, float h_safe, const AssocOut& out, int want_aux,
                 const int chunk, const int n_waves) {
    __shared__ int s_j0[64], s_off[66];
    const int lane = threadIdx.x & 63;
    if (chain_fail && *chain_fail) return;                             // the tube launch ahead of this one has raised it if the record was not ready
    const int count = *out.ask_count;
    const int n_chunks = (count + kAskChunk - 1) / kAskChunk;
    if (out.ask_map == 0 && chunk >= n_chunks) return;
    const PoseScalars& P = P_dev ? P_dev->P : P_in;
    const unsigned long long key_inf = ((unsigned long long)gate_bits + 1ull) << 32;
    const GridDesc g = G.d;
    // Which kAskChunk entries this wave takes.  0: with a stride -- the list's neighbours (a clump of heavy queries) go to different waves,
    // which evens the waves out but makes 8 different CUs on 8 different XCDs fetch the same cells of a clump.  1: a contiguous block, and
    // blocks dealt so that workgroups 8 j + k (XCD k) take the k-th EIGHTH of the list: neighbours share a wave's L1 and an XCD's L2.
    int my_idx = chunk + lane * n_chunks;
    if (out.ask_map == 1) {
        const int per = (n_chunks + 7) >> 3, blk = (chunk & 7) * per + (chunk >> 3);
        if ((chunk >> 3) >= per || blk >= n_chunks) return;
        my_idx = blk * kAskChunk + lane;
    }
    const bool mine = lane < kAskChunk && my_idx < count;
    int qi = 0;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    Top2 t;
    t.b1 = key_inf; t.b2 = key_inf; t.b1ring = -1; t.b2ring = -1; t.b2d = __uint_as_float(gate_bits + 1u);
    if (mine) {
        qi = out.ask_list[my_idx];
        const float4 psrc = qpts[qi];
        t.b1 = out.ask_keys[2 * (size_t)qi]; t.b2 = out.ask_keys[2 * (size_t)qi + 1];
        const int2 rr = out.ask_rings[qi];
        t.b1ring = rr.x; t.b2ring = rr.y;
        t.b2d = __uint_as_float((unsigned)(t.b2 >> 32));
        transform_query(P, psrc, &qx, &qy, &qz);
    }
    const float cap = 4.0f * h_safe;
    const unsigned long long mm = __ballot(mine);
    for (int a = 0; a < kAskChunk; a++) {
        if (!((mm >> a) & 1ull)) continue;                             // wave-uniform
        const float ax = __shfl(qx, a), ay = __shfl(qy, a), az = __shfl(qz, a);
        Top2 tl;
        tl.b1 = __shfl(t.b1, a); tl.b2 = __shfl(t.b2, a); tl.b1ring = __shfl(t.b1ring, a); tl.b2ring = __shfl(t.b2ring, a);
        tl.b2d = __uint_as_float((unsigned)(tl.b2 >> 32));
        const float r0 = sqrtf(tl.b2d) * 1.0001f + 1e-6f;
        asker_search(G, g, ax, ay, az, fminf(r0, cap), tl, lane, s_j0, s_off);
        const float r1 = sqrtf(tl.b2d) * 1.0001f + 1e-6f;
        if (r1 > cap) asker_search(G, g, ax, ay, az, r1, tl, lane, s_j0, s_off);      // wave-uniform: tl is merged
        if (lane == a) { t.b1 = tl.b1; t.b2 = tl.b2; t.b1ring = tl.b1ring; t.b2ring = tl.b2ring; t.b2d = tl.b2d; }
    }
    if (mine) finish_correspondence_pad(qi, qpts, qx, qy, qz, t.b1, t.b2, t.b1ring, t.b2ring, key_inf, tgt_pad, tgt_off, norm_cond, out, want_aux != 0);
}
__global__ void __launch_bounds__(64)
assoc_asker_kernel(PoseScalars P, const PoseRecord* __restrict__ P_dev, const int* __restrict__ chain_fail, GridView G, const float4* __restrict__ qpts,
                   const float4* __restrict__ tgt_pad, const int* __restrict__ tgt_off, unsigned gate_bits, double norm_cond, float h_safe, AssocOut out, int want_aux)
#if VELO_DEF_ASSOC
{
    assoc_asker_body(P, P_dev, chain_fail, G, qpts, tgt_pad, tgt_off, gate_bits, norm_cond, h_safe, out, want_aux, (int)blockIdx.x, (int)gridDim.x);
}
#else
;
#endif
__global__ void __launch_bounds__(64)
assoc_asker_batch_kernel(AssocBatch B)
#if VELO_DEF_ASSOC
{
    const AssocArgs& a = B.item[blockIdx.y];
    const int n_waves = (a.q_end - a.q_begin + kAskChunk - 1) / kAskChunk;
    if (!a.out.ask_list || (int)blockIdx.x >= n_waves + 8) return;      // (+ 8: the XCD-chunked map rounds the list up to eight equal parts)
    assoc_asker_body(a.P, a.P_dev, a.chain_fail, a.G, a.qpts, a.tgt_pad, a.tgt_off, a.gate_bits, a.norm_cond, a.h_safe, a.out, a.want_aux, (int)blockIdx.x, n_waves);
}
#else
;
#endif

// ---- seeds from the target's direction image -----------------------------------------------------------------------------------
// A round's search starts from two candidates per query ("seeds", AssocOut::prev_*): any real target points will do -- the tube kernel
// recomputes their distances and the result is exact whatever they are -- but the tighter the second-best bound they give, the fewer
// rows, candidates and asking queries the round has.  The previous round's winners are good seeds while the pose moves by millimetres
// (rounds 4-6 of a call); the FIRST round has none, and rounds 2-3 follow solves that moved the pose by centimetres to a metre.
// For those rounds the seeds come from the target itself: a scan (and a map in its newest pose's frame) is a range image around the
// origin, so the target point seen in (nearly) the direction of the transformed query lies on (nearly) the same surface.  The
// direction image is a kDimgW x kDimgH table over (azimuth, elevation) holding, per bucket, the NEAREST point that falls into it
// (64-bit key: range^2 bits << 32 | local index; built with one atomicMin per target point).  seed_kernel looks the query's bucket and
// its four neighbours up, adds the previous winners when there are any, and leaves the best two of different rings as the seeds.
// Only a hash: for clouds that are no range images the seeds are merely poor.  Results never depend on it.
#ifndef VELO_DIMG_W
#define VELO_DIMG_W 1024
#define VELO_DIMG_H 256
#endif
constexpr int kDimgW = VELO_DIMG_W, kDimgH = VELO_DIMG_H;       // 1024 x 256: 0.35 deg x 0.35 deg over +-45 deg of elevation, 2 MB (W: a power of two)
__device__ __forceinline__ void dimg_bucket(float x, float y, float z, int* a, int* e) {
    // camera frame (x right, y down, z forward; kitti.h:100-107): azimuth about the y axis, elevation up positive
    const float az = atan2f(x, z), el = atan2f(-y, sqrtf(x * x + z * z));
    int ia = (int)((az + 3.14159265f) * ((float)kDimgW / 6.2831853f)), ie = (int)((el + 0.78539816f) * ((float)kDimgH / 1.5707963f));
    *a = min(max(ia, 0), kDimgW - 1);
    *e = min(max(ie, 0), kDimgH - 1);
}
__global__ void __launch_bounds__(256)
dimg_build_kernel(const float4* __restrict__ tgt, int n, unsigned long long* __restrict__ dimg)
#if VELO_DEF_ASSOC
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = tgt[i];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) return;
    int a, e;
    dimg_bucket(p.x, p.y, p.z, &a, &e);
    const float r2 = p.x * p.x + p.y * p.y + p.z * p.z;
    atomicMin(&dimg[e * kDimgW + a], ((unsigned long long)__float_as_uint(r2) << 32) | (unsigned)i);
}
#else
;
#endif
struct SeedArgs {
    PoseScalars P; const PoseRecord* P_dev; const int* chain_fail;
    const float4* qpts; int q_begin, q_end;
    const unsigned long long* dimg; const float4* tgt; const int* ring_of;    // direction image, ring-major cloud, global ring of each local point
    int first_point;
    float4* prev_a; float4* prev_b; int2* prev_r;
    int has_prev;                                                              // the arrays hold the previous round's winners (else: uninitialised)
};
struct SeedBatch { SeedArgs item[kAssocBatchMax]; };
struct SeedState { unsigned long long b1, b2; int r1, r2; float x1, y1, z1, x2, y2, z2; int i1, i2; };
__device__ __forceinline__ void seed_enter(SeedState& t, float qx, float qy, float qz, float x, float y, float z, int idx, int ring, int first_point) {
    const float d = dist2_f(x, y, z, qx, qy, qz);
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(idx + first_point);
    if (!(key < t.b2)) return;
    if (key < t.b1) {
        if (ring != t.r1) { t.b2 = t.b1; t.r2 = t.r1; t.x2 = t.x1; t.y2 = t.y1; t.z2 = t.z1; t.i2 = t.i1; }
        t.b1 = key; t.r1 = ring; t.x1 = x; t.y1 = y; t.z1 = z; t.i1 = idx;
    } else if (ring != t.r1) {
        t.b2 = key; t.r2 = ring; t.x2 = x; t.y2 = y; t.z2 = z; t.i2 = idx;
    }
}
__device__ __forceinline__ void seed_body(const SeedArgs& A, const int i) {
    if (i >= A.q_end) return;
    if (A.chain_fail && *A.chain_fail) return;
    if (A.P_dev && !A.P_dev->ready) return;                                    // the association launch behind this one raises the chain's flag
    const PoseScalars& P = A.P_dev ? A.P_dev->P : A.P;
    float qx, qy, qz;
    transform_query(P, A.qpts[i], &qx, &qy, &qz);
    SeedState t;
    t.b1 = ~0ull; t.b2 = ~0ull; t.r1 = -1; t.r2 = -1; t.x1 = t.y1 = t.z1 = t.x2 = t.y2 = t.z2 = 0.f; t.i1 = -1; t.i2 = -1;
    if (A.has_prev) {
        const float4 sa = A.prev_a[i], sb = A.prev_b[i];
        const int2 sr = A.prev_r[i];
        if (__float_as_int(sa.w) >= 0) seed_enter(t, qx, qy, qz, sa.x, sa.y, sa.z, __float_as_int(sa.w), sr.x, A.first_point);
        if (__float_as_int(sb.w) >= 0) seed_enter(t, qx, qy, qz, sb.x, sb.y, sb.z, __float_as_int(sb.w), sr.y, A.first_point);
    }
    int a, e;
    dimg_bucket(qx, qy, qz, &a, &e);
    // the query's bucket and its four neighbours: all look-ups first, then all gathers, then the (rare) updates -- scalars, no arrays
    const unsigned long long k0 = A.dimg[e * kDimgW + a], k1 = A.dimg[max(e - 1, 0) * kDimgW + a], k2 = A.dimg[min(e + 1, kDimgH - 1) * kDimgW + a];
    const unsigned long long k3 = A.dimg[e * kDimgW + ((a + 1) & (kDimgW - 1))], k4 = A.dimg[e * kDimgW + ((a + kDimgW - 1) & (kDimgW - 1))];
    const int j0 = (k0 == ~0ull) ? 0 : (int)(unsigned)(k0 & 0xffffffffull), j1 = (k1 == ~0ull) ? 0 : (int)(unsigned)(k1 & 0xffffffffull);
    const int j2 = (k2 == ~0ull) ? 0 : (int)(unsigned)(k2 & 0xffffffffull), j3 = (k3 == ~0ull) ? 0 : (int)(unsigned)(k3 & 0xffffffffull);
    const int j4 = (k4 == ~0ull) ? 0 : (int)(unsigned)(k4 & 0xffffffffull);
    const float4 p0 = A.tgt[j0], p1 = A.tgt[j1], p2 = A.tgt[j2], p3 = A.tgt[j3], p4 = A.tgt[j4];
    const int g0 = A.ring_of[j0], g1 = A.ring_of[j1], g2 = A.ring_of[j2], g3 = A.ring_of[j3], g4 = A.ring_of[j4];
    if (k0 != ~0ull) seed_enter(t, qx, qy, qz, p0.x, p0.y, p0.z, j0, g0, A.first_point);
    if (k1 != ~0ull) seed_enter(t, qx, qy, qz, p1.x, p1.y, p1.z, j1, g1, A.first_point);
    if (k2 != ~0ull) seed_enter(t, qx, qy, qz, p2.x, p2.y, p2.z, j2, g2, A.first_point);
    if (k3 != ~0ull) seed_enter(t, qx, qy, qz, p3.x, p3.y, p3.z, j3, g3, A.first_point);
    if (k4 != ~0ull) seed_enter(t, qx, qy, qz, p4.x, p4.y, p4.z, j4, g4, A.first_point);
    A.prev_a[i] = make_float4(t.x1, t.y1, t.z1, __int_as_float(t.i1));
    A.prev_b[i] = make_float4(t.x2, t.y2, t.z2, __int_as_float(t.i2));
    A.prev_r[i] = make_int2(t.r1, t.r2);
}
__global__ void __launch_bounds__(256)
seed_kernel(SeedArgs A)
#if VELO_DEF_ASSOC
{ seed_body(A, A.q_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x)); }
#else
;
#endif
__global__ void __launch_bounds__(256)
seed_batch_kernel(SeedBatch B)
#if VELO_DEF_ASSOC
{ const SeedArgs& A = B.item[blockIdx.y]; seed_body(A, A.q_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x)); }
#else
;
#endif

// ---- sparse queries: one wave per query -------------------------------------------------------------------------------------
// With the reference's own constants (icp_skip = 200, kitti.h:8) a round has 640 queries, 6 m apart along their rings: the 64
// queries of a tube group share nothing, its tube is 64 separate boxes, and ten workgroups walk them one row chunk after the other
// (55-150 us per round -- more than the six solves of the call together).  Here every query gets a wave of its own: the 64 lanes
// search its bound sphere exactly as assoc_asker_kernel does for a listed query (seeds first, then no farther than 4 cells,
// then the whole sphere if the bound still reaches beyond), lane 0 finishes the correspondence.  640 waves, all resident at once.
__device__ __forceinline__ void
assoc_direct_body(const PoseScalars& P_in, const PoseRecord* __restrict__ P_dev, int* __restrict__ chain_fail, const GridView& G, const float4* __restrict__ qpts,
                  int q_begin, int q_end, const float4* __restrict__ tgt_pad, const int* __restrict__ tgt_off, unsigned gate_bits, double norm_cond, float h_safe,
                  const AssocOut& out, int want_aux, const int block_x) {
    __shared__ int s_j0[64], s_off[66];
    const int lane = threadIdx.x & 63;
    if (chain_fail && *chain_fail) return;
    if (P_dev && !P_dev->ready) { if (block_x == 0 && lane == 0) *chain_fail = 1; return; }
    const PoseScalars& P = P_dev ? P_dev->P : P_in;
    if (out.n_valid_next && block_x == 0 && lane == 0) *out.n_valid_next = 0;
    const int qi = q_begin + block_x;
    if (qi >= q_end) return;
    const unsigned long long key_inf = ((unsigned long long)gate_bits + 1ull) << 32;
    const GridDesc g = G.d;
    Top2 tl;
    tl.b1 = key_inf; tl.b2 = key_inf; tl.b1ring = -1; tl.b2ring = -1; tl.b2d = __uint_as_float(gate_bits + 1u);
    float qx, qy, qz;
    {   // every lane holds the same query (uniform loads, the transform costs what it costs on one lane)
        const float4 psrc = qpts[qi];
        transform_query(P, psrc, &qx, &qy, &qz);
        if (out.prev_a) {
            const float4 sa = out.prev_a[qi], sb = out.prev_b[qi];
            const int2 sr = out.prev_r[qi];
            if (__float_as_int(sa.w) >= 0) {
                const float d = dist2_f(sa.x, sa.y, sa.z, qx, qy, qz);
                if (__float_as_uint(d) <= gate_bits) top2_update(tl, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(__float_as_int(sa.w) + out.first_point), sr.x);
            }
            if (__float_as_int(sb.w) >= 0) {
                const float d = dist2_f(sb.x, sb.y, sb.z, qx, qy, qz);
                if (__float_as_uint(d) <= gate_bits) top2_update(tl, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(__float_as_int(sb.w) + out.first_point), sr.y);
            }
        }
    }
    const float cap = 4.0f * h_safe;
    const float r0 = sqrtf(tl.b2d) * 1.0001f + 1e-6f;
    asker_search(G, g, qx, qy, qz, fminf(r0, cap), tl, lane, s_j0, s_off);
    const float r1 = sqrtf(tl.b2d) * 1.0001f + 1e-6f;
    if (r1 > cap) asker_search(G, g, qx, qy, qz, r1, tl, lane, s_j0, s_off);
    if (lane == 0) finish_correspondence_pad(qi, qpts, qx, qy, qz, tl.b1, tl.b2, tl.b1ring, tl.b2ring, key_inf, tgt_pad, tgt_off, norm_cond, out, want_aux != 0);
}
__global__ void __launch_bounds__(64)
assoc_direct_kernel(PoseScalars P, const PoseRecord* __restrict__ P_dev, int* __restrict__ chain_fail, GridView G, const float4* __restrict__ qpts, int q_begin, int q_end,
                    const float4* __restrict__ tgt_pad, const int* __restrict__ tgt_off, unsigned gate_bits, double norm_cond, float h_safe, AssocOut out, int want_aux)
#if VELO_DEF_ASSOC
{
    assoc_direct_body(P, P_dev, chain_fail, G, qpts, q_begin, q_end, tgt_pad, tgt_off, gate_bits, norm_cond, h_safe, out, want_aux, (int)blockIdx.x);
}
#else
;
#endif
__global__ void __launch_bounds__(64)
assoc_direct_batch_kernel(AssocBatch B)
#if VELO_DEF_ASSOC
{
    const AssocArgs& a = B.item[blockIdx.y];
    if ((int)blockIdx.x >= a.q_end - a.q_begin) return;
    assoc_direct_body(a.P, a.P_dev, a.chain_fail, a.G, a.qpts, a.q_begin, a.q_end, a.tgt_pad, a.tgt_off, a.gate_bits, a.norm_cond, a.h_safe, a.out, a.want_aux, (int)blockIdx.x);
}
#else
;
#endif

// ---- association search, lane variant: the rounds that start from seeds ------------------------------------------------------
// The tube kernel shares every staged candidate among the 64 queries of a group: 350 candidates x 64 lanes per warm round, of
// which a query needs the ~15 of its own bound sphere -- 9,400 lane-instructions per query.  A round that starts from the
// previous round's winners knows a tight bound for (nearly) every query BEFORE it looks at a single cell, so sharing buys
// nothing: here ONE LANE OWNS ONE QUERY and walks the cells of its own bound sphere straight from the cell-sorted copy in L2
// (neighbouring lanes walk neighbouring cells, so the gathers share lines).  No tile, no run list of the group, no workgroup
// barrier -- a workgroup is four independent waves.
//   * easy query (bound box <= 3 x 3 rows, the rule whenever the bound is below one cell): the lane fetches the (start, end)
//     of its <= 9 row runs at once, keeps the non-empty ones in its LDS column and then consumes them four candidates per trip
//     (all four loads in flight together; entries behind the end of a run are real points of the next cells or the +inf
//     sentinels behind the last point -- harmless extra candidates);
//   * hard query (a query whose second ring is farther than a cell: no second seed, first iteration's gate = 4 cells): the
//     wave turns all 64 lanes on that one query, as the tube kernel's asker phase does (rows over lanes -> wave prefix sum ->
//     candidates over lanes -> xor-shuffle merge).
// Candidate set = every point of every cell the bound sphere touches, keys and tie rules as everywhere else: the tables equal the
// tube kernel's bit for bit (tests: warm rounds against cold rounds, against the oracle, lane against tube).
// MEASURED (C2, one pair in flight): 88-90 us per second-iteration round against the tube kernel's 46-50, 440 us per seeded
// first-iteration round (10-25 % hard queries, each a 9 x 9-row box) against 80.  The instruction count is what was hoped for, but
// every lane-private 16-byte gather pulls a 128-byte line from L2 into a 16 KB L1 that 64 lanes x 8 gathers per trip thrash: the
// kernel waits on L2->L1 line traffic (~0.8 MB per wave) that the tube kernel's coalesced staging never creates.  Kept as an
// independent second implementation for the parity tests and as an A/B (VELO_ASSOC_LANE=1), NOT the default.
template <bool DBG>
__device__ __forceinline__ void
assoc_lane_body(const PoseScalars& P_in, const PoseRecord* __restrict__ P_dev, int* __restrict__ chain_fail, const GridView& G, const float4* __restrict__ qpts, int q_begin, int q_end,
                const float4* __restrict__ tgt_pad, const int* __restrict__ tgt_off, unsigned gate_bits, double norm_cond, const AssocOut& out, int want_aux, const int block_x) {
    constexpr int kRows = 9;
    __shared__ int s_j0[4][kRows][64];
    __shared__ int s_j1[4][kRows][64];
    __shared__ int s_cj0[4][66];
    __shared__ int s_coff[4][66];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (chain_fail && *chain_fail) return;
    if (P_dev && !P_dev->ready) { if (block_x == 0 && tid == 0) *chain_fail = 1; return; }
    const PoseScalars& P = P_dev ? P_dev->P : P_in;
    if (out.n_valid_next && block_x == 0 && tid == 0) *out.n_valid_next = 0;
    const int group = block_x * 4 + wid;
    if (group * 64 >= q_end - q_begin) return;                         // wave-uniform
    const int qi = q_begin + group * 64 + lane;
    const bool active = qi < q_end;
    const unsigned long long key_inf = ((unsigned long long)gate_bits + 1ull) << 32;
    Top2 t;
    t.b1 = key_inf; t.b2 = key_inf; t.b1ring = -1; t.b2ring = -1; t.b2d = __uint_as_float(gate_bits + 1u);
    const GridDesc g = G.d;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (active) {
        const float4 psrc = qpts[qi];
        const float4 sa = out.prev_a[qi], sb = out.prev_b[qi];
        const int2 sr = out.prev_r[qi];
        transform_query(P, psrc, &qx, &qy, &qz);
        if (__float_as_int(sa.w) >= 0) {
            const float d = dist2_f(sa.x, sa.y, sa.z, qx, qy, qz);
            if (__float_as_uint(d) <= gate_bits) top2_update(t, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(__float_as_int(sa.w) + out.first_point), sr.x);
        }
        if (__float_as_int(sb.w) >= 0) {
            const float d = dist2_f(sb.x, sb.y, sb.z, qx, qy, qz);
            if (__float_as_uint(d) <= gate_bits) top2_update(t, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(__float_as_int(sb.w) + out.first_point), sr.y);
        }
    }
    // the cell box of the bound sphere (padded against rounding, like the tube kernel's)
    const float r0 = sqrtf(t.b2d) * 1.0001f + 1e-6f;
    int x0, x1, y0, y1, z0, z1;
    {
        const CellBox bb = query_box(g, qx, qy, qz, 0, 0, 0, r0, false);
        x0 = max(bb.x0, 0); x1 = min(bb.x1, g.nx - 1); y0 = max(bb.y0, 0); y1 = min(bb.y1, g.ny - 1); z0 = max(bb.z0, 0); z1 = min(bb.z1, g.nz - 1);
    }
    const int nyb = y1 - y0 + 1, nzb = z1 - z0 + 1;
    const bool some = active && x0 <= x1 && nyb > 0 && nzb > 0;        // else: the sphere lies outside the grid, nothing to look at
    const bool easy = some && nyb <= 3 && nzb <= 3;
    const bool hard = some && !easy;
    // ---- easy lanes: own row runs -> LDS column -> four candidates per trip ----
    int nrun = 0;
    if (easy) {
        int a[kRows], b[kRows];
#pragma unroll
        for (int zz = 0; zz < 3; zz++) {
#pragma unroll
            for (int yy = 0; yy < 3; yy++) {
                const bool on = zz < nzb && yy < nyb;
                const int row = on ? ((z0 + zz) * g.ny + (y0 + yy)) : 0;
                a[zz * 3 + yy] = on ? grid_start(G, row, x0) : 0;
                b[zz * 3 + yy] = on ? grid_start(G, row, x1 + 1) : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < kRows; k++) {
            if (b[k] > a[k]) { s_j0[wid][nrun][lane] = a[k]; s_j1[wid][nrun][lane] = b[k]; nrun++; }
        }
    }
    {
        int k = 0, j = 0, jend = 0;
        for (;;) {
            if (j >= jend && k < nrun) { j = s_j0[wid][k][lane]; jend = s_j1[wid][k][lane]; k++; }
            const bool have = j < jend;
            if (__ballot(have) == 0ull) break;
            const int jj = have ? j : 0;
            const float4 c0 = G.sorted[jj], c1 = G.sorted[jj + 1], c2 = G.sorted[jj + 2], c3 = G.sorted[jj + 3];
            const int g0 = G.sring[jj], g1 = G.sring[jj + 1], g2 = G.sring[jj + 2], g3 = G.sring[jj + 3];
            const float d0 = dist2_f(qx, qy, qz, c0.x, c0.y, c0.z), d1 = dist2_f(qx, qy, qz, c1.x, c1.y, c1.z);
            const float d2 = dist2_f(qx, qy, qz, c2.x, c2.y, c2.z), d3 = dist2_f(qx, qy, qz, c3.x, c3.y, c3.z);
            if (have && fminf(fminf(d0, d1), fminf(d2, d3)) <= t.b2d) {
                if (d0 <= t.b2d) top2_update(t, ((unsigned long long)__float_as_uint(d0) << 32) | (unsigned)__float_as_int(c0.w), g0);
                if (d1 <= t.b2d) top2_update(t, ((unsigned long long)__float_as_uint(d1) << 32) | (unsigned)__float_as_int(c1.w), g1);
                if (d2 <= t.b2d) top2_update(t, ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(c2.w), g2);
                if (d3 <= t.b2d) top2_update(t, ((unsigned long long)__float_as_uint(d3) << 32) | (unsigned)__float_as_int(c3.w), g3);
            }
            j += 4;
        }
    }
    // ---- hard lanes: the whole wave on one query at a time ----
    unsigned long long hm = __ballot(hard);
    while (hm != 0ull) {
        const int la = (int)__ffsll((long long)hm) - 1;
        hm &= hm - 1ull;
        const float ax = __shfl(qx, la), ay = __shfl(qy, la), az = __shfl(qz, la);
        Top2 tl;
        tl.b1 = __shfl(t.b1, la); tl.b2 = __shfl(t.b2, la); tl.b1ring = __shfl(t.b1ring, la); tl.b2ring = __shfl(t.b2ring, la);
        tl.b2d = __uint_as_float((unsigned)(tl.b2 >> 32));
        const int bx0 = __shfl(x0, la), bx1 = __shfl(x1, la), by0 = __shfl(y0, la), bz0 = __shfl(z0, la);
        const int ny = __shfl(nyb, la), nz = __shfl(nzb, la);
        const int nrows_a = ny * nz;
        const float rcp_ny = 1.0f / (float)ny;
        for (int rb = 0; rb < nrows_a; rb += 64) {
            const int r = rb + lane;
            int j0 = 0, len = 0;
            if (r < nrows_a) {
                int zq = (int)((float)r * rcp_ny), yr = r - zq * ny;
                if (yr < 0) { zq--; yr += ny; } else if (yr >= ny) { zq++; yr -= ny; }
                const int row = ((bz0 + zq) * g.ny + (by0 + yr));
                j0 = grid_start(G, row, bx0); len = grid_start(G, row, bx1 + 1) - j0;
            }
            int inc = len;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
            const int total = __shfl(inc, 63);
            s_cj0[wid][lane] = j0; s_coff[wid][lane] = inc - len;
            if (lane == 0) s_coff[wid][64] = total;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int slot = lane; slot < total; slot += 64) {
                int lo = 0;                                            // largest i with s_coff[i] <= slot
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) { if (s_coff[wid][lo + step] <= slot) lo += step; }
                const int j = s_cj0[wid][lo] + (slot - s_coff[wid][lo]);
                const float4 c = G.sorted[j];
                const float d = dist2_f(ax, ay, az, c.x, c.y, c.z);
                if (d <= tl.b2d) top2_update(tl, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(c.w), G.sring[j]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                           // the run list is rewritten by the next rows
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) top2_merge_xor(tl, m);
        if (lane == la) { t.b1 = tl.b1; t.b2 = tl.b2; t.b1ring = tl.b1ring; t.b2ring = tl.b2ring; t.b2d = tl.b2d; }
    }
    if (active) finish_correspondence_pad(qi, qpts, qx, qy, qz, t.b1, t.b2, t.b1ring, t.b2ring, key_inf, tgt_pad, tgt_off, norm_cond, out, want_aux != 0);
}

__global__ void __launch_bounds__(256)
assoc_lane_kernel(PoseScalars P, const PoseRecord* __restrict__ P_dev, int* __restrict__ chain_fail, GridView G, const float4* __restrict__ qpts, int q_begin, int q_end,
                  const float4* __restrict__ tgt_pad, const int* __restrict__ tgt_off, unsigned gate_bits, double norm_cond, AssocOut out, int want_aux)
#if VELO_DEF_ASSOC
{
    assoc_lane_body<false>(P, P_dev, chain_fail, G, qpts, q_begin, q_end, tgt_pad, tgt_off, gate_bits, norm_cond, out, want_aux, (int)blockIdx.x);
}
#else
;
#endif
// the same round of several contexts in one launch (blockIdx.y = context)
__global__ void __launch_bounds__(256)
assoc_lane_batch_kernel(AssocBatch B)
#if VELO_DEF_ASSOC
{
    const AssocArgs& a = B.item[blockIdx.y];
    if ((int)blockIdx.x * 256 >= a.q_end - a.q_begin) return;
    assoc_lane_body<false>(a.P, a.P_dev, a.chain_fail, a.G, a.qpts, a.q_begin, a.q_end, a.tgt_pad, a.tgt_off, a.gate_bits, a.norm_cond, a.out, a.want_aux, (int)blockIdx.x);
}
#else
;
#endif

// ---- association as a balanced pipeline: prepare (clusters -> work items) + persistent per-cluster search -----------------
// The monolithic kernel above walks a group's clusters one after the other, so a 64-query group with several clusters and
// two phases each is a long latency chain while most workgroups have already left.  Here a cheap prepare kernel (one
// wave per group) transforms the queries, forms the clusters and appends ONE WORK ITEM PER CLUSTER; a persistent kernel
// then keeps every workgroup slot busy: workgroups pull items from a global queue, run that cluster's phases (same LDS
// staged box walk) and finish the correspondences of the cluster's member lanes.
struct AssocItem {            // 32 bytes
    int group;                // 64-query group
    unsigned mask_lo, mask_hi;
    int bx, by, bz;           // cell bounding box of the members: lo | hi << 16
    int pad0, pad1;
};
// The queue is split into kQShards sub-queues (one returning atomic on ONE word tops out near 90 per microsecond on
// this chip): group g appends to shard g % kQShards with a single reservation, a workgroup pulls from shard
// blockIdx % kQShards first and steals from the others when its own runs dry.  Counters sit on separate 128-byte lines.
constexpr int kQShards = 8;
constexpr int kQStride = 32;         // ints between counters
struct AssocQueue {
    AssocItem* __restrict__ items;   // kQShards regions of shard_cap items
    int shard_cap;
    int* __restrict__ counters;      // [s * kQStride] = items in shard s, [(kQShards + s) * kQStride] = dequeue head of shard s
    float4* __restrict__ qpos;       // transformed query coordinates (float, as the reference rounds them) per query
};

__global__ void __launch_bounds__(64)
assoc_prepare_kernel(PoseScalars P, GridDesc g, const float4* __restrict__ src, const int* __restrict__ q_src, int q_begin, int q_end,
                     int cluster_w, AssocQueue Q)
#if VELO_DEF_ASSOC
{
    const int lane = threadIdx.x;
    const int group = blockIdx.x;
    const int qi = q_begin + group * 64 + lane;
    const bool active = qi < q_end;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    int cx = 0, cy = 0, cz = 0;
    if (active) {
        const float4 psrc = src[q_src[qi]];
        transform_query(P, psrc, &qx, &qy, &qz);
        cx = cell_coord(qx, g.ox, g.inv_h, g.nx); cy = cell_coord(qy, g.oy, g.inv_h, g.ny); cz = cell_coord(qz, g.oz, g.inv_h, g.nz);
        Q.qpos[qi] = make_float4(qx, qy, qz, 0.f);
    }
    // pass 1: count this group's clusters; one reservation per group in its shard
    int n_clusters = 0;
    {
        bool pending = active;
        for (;;) {
            const unsigned long long pm = __ballot(pending);
            if (pm == 0ull) break;
            const int leader = (int)__ffsll((long long)pm) - 1;
            const int scx = __builtin_amdgcn_readlane(cx, leader), scy = __builtin_amdgcn_readlane(cy, leader), scz = __builtin_amdgcn_readlane(cz, leader);
            const bool member = pending && abs(cx - scx) <= cluster_w && abs(cy - scy) <= cluster_w && abs(cz - scz) <= cluster_w;
            n_clusters++;
            pending = pending && !member;
        }
    }
    const int shard = group % kQShards;
    int base = 0;
    if (lane == 0 && n_clusters > 0) base = atomicAdd(&Q.counters[shard * kQStride], n_clusters);
    base = __builtin_amdgcn_readfirstlane(base);
    // pass 2: write the items
    bool pending = active;
    int k = 0;
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (pm == 0ull) break;
        const int leader = (int)__ffsll((long long)pm) - 1;
        const int scx = __builtin_amdgcn_readlane(cx, leader), scy = __builtin_amdgcn_readlane(cy, leader), scz = __builtin_amdgcn_readlane(cz, leader);
        const bool member = pending && abs(cx - scx) <= cluster_w && abs(cy - scy) <= cluster_w && abs(cz - scz) <= cluster_w;
        const unsigned long long mm = __ballot(member);
        const int big = 1 << 28;
        const int bx0 = wave_min_i(member ? cx : big), bx1 = wave_max_i(member ? cx : -big);
        const int by0 = wave_min_i(member ? cy : big), by1 = wave_max_i(member ? cy : -big);
        const int bz0 = wave_min_i(member ? cz : big), bz1 = wave_max_i(member ? cz : -big);
        if (lane == 0) {
            AssocItem it;
            it.group = group; it.mask_lo = (unsigned)(mm & 0xffffffffull); it.mask_hi = (unsigned)(mm >> 32);
            // cell coordinates live in [-2, n+1] with n <= 8192: bias by 2 and pack two 16-bit fields
            it.bx = (bx0 + 2) | ((bx1 + 2) << 16); it.by = (by0 + 2) | ((by1 + 2) << 16); it.bz = (bz0 + 2) | ((bz1 + 2) << 16);
            it.pad0 = 0; it.pad1 = 0;
            Q.items[(size_t)shard * Q.shard_cap + base + k] = it;
        }
        k++;
        pending = pending && !member;
    }
}
#else
;
#endif

template <int NW, int MINW>
__global__ void __launch_bounds__(NW * 64, MINW)
assoc_cluster_kernel(GridView G, AssocQueue Q, const float4* __restrict__ src, const int* __restrict__ q_src, int q_begin, int q_end,
                     const float4* __restrict__ tgt, const int* __restrict__ tgt_off, const int* __restrict__ ring_of,
                     unsigned gate_bits, double norm_cond, float h_safe, AssocOut out, int want_aux) {
    constexpr int NT = NW * 64;
    constexpr int NRUN = 2 * NT;
    __shared__ float4 s_xy[kTileCap / 2];
    __shared__ float4 s_zg[kTileCap / 2];
    __shared__ int s_ring[kTileCap];
    __shared__ int s_run_j0[NRUN];
    __shared__ int s_run_off[NRUN + 1];
    __shared__ int s_wave_tot[NW];
    __shared__ unsigned long long m1[NW][64], m2[NW][64];
    __shared__ int mr[NW][64];
    __shared__ int s_item;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long key_inf = ((unsigned long long)gate_bits + 1ull) << 32;
    const GridDesc g = G.d;
    float* s_xy_f = reinterpret_cast<float*>(s_xy);
    float* s_zg_f = reinterpret_cast<float*>(s_zg);
    int shard = blockIdx.x % kQShards, tries = 0;

    for (;;) {                                                         // persistent: pull the next cluster
        if (tid == 0) {
            int got = -1;
            while (tries < kQShards) {
                const int n = Q.counters[shard * kQStride];
                int* head = &Q.counters[(kQShards + shard) * kQStride];
                // plain peek first: an exhausted shard costs no atomic
                if (__hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
                    const int h = atomicAdd(head, 1);
                    if (h < n) { got = shard * Q.shard_cap + h; break; }
                }
                shard = (shard + 1) % kQShards; tries++;
            }
            s_item = got;
        }
        __syncthreads();
        const int item = s_item;
        __syncthreads();
        if (item < 0) break;
        const AssocItem it = Q.items[item];
        const unsigned long long mm = ((unsigned long long)it.mask_hi << 32) | it.mask_lo;
        const bool member = (mm >> lane) & 1ull;
        const int qi = q_begin + it.group * 64 + lane;
        const int bx0 = (it.bx & 0xffff) - 2, bx1 = (it.bx >> 16) - 2;
        const int by0 = (it.by & 0xffff) - 2, by1 = (it.by >> 16) - 2;
        const int bz0 = (it.bz & 0xffff) - 2, bz1 = (it.bz >> 16) - 2;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (member) { const float4 qp = Q.qpos[qi]; qx = qp.x; qy = qp.y; qz = qp.z; }
        const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        Top2 t;
        t.b1 = key_inf; t.b2 = key_inf; t.b1ring = -1; t.b2ring = -1; t.b2d = __uint_as_float(gate_bits + 1u);
        int e_prev = -1, e = 1;
        for (;;) {                                                     // phases (see assoc_search_v3_kernel)
            const int X0 = max(bx0 - e, 0), X1 = min(bx1 + e, g.nx - 1);
            const int Y0 = max(by0 - e, 0), Y1 = min(by1 + e, g.ny - 1);
            const int Z0 = max(bz0 - e, 0), Z1 = min(bz1 + e, g.nz - 1);
            const int px0 = bx0 - e_prev, px1 = bx1 + e_prev, py0 = by0 - e_prev, py1 = by1 + e_prev, pz0 = bz0 - e_prev, pz1 = bz1 + e_prev;
            const int nyb = Y1 - Y0 + 1, nzb = Z1 - Z0 + 1;
            const int nrows = (X0 <= X1 && nyb > 0 && nzb > 0) ? nyb * nzb : 0;
            const float inv_nyb = 1.0f / (float)max(nyb, 1);
            for (int rbase = 0; rbase < nrows; rbase += NT) {
                int ja0 = 0, la = 0, jb0 = 0, lb = 0;
                const int r = rbase + tid;
                if (r < nrows) {
                    int zq = (int)(((float)r + 0.5f) * inv_nyb);      // r / nyb for small non-negative ints, fixed up below
                    zq = (zq * nyb > r) ? zq - 1 : ((zq + 1) * nyb <= r ? zq + 1 : zq);
                    const int y = Y0 + (r - zq * nyb), z = Z0 + zq;
                    const int row = (z * g.ny + y);
                    const bool fresh = e_prev < 0 || y < py0 || y > py1 || z < pz0 || z > pz1;
                    if (fresh) {
                        ja0 = grid_start(G, row, X0); la = grid_start(G, row, X1 + 1) - ja0;
                    } else {
                        const int a1 = min(px0 - 1, X1), b0 = max(px1 + 1, X0);
                        if (X0 <= a1) { ja0 = grid_start(G, row, X0); la = grid_start(G, row, a1 + 1) - ja0; }
                        if (b0 <= X1) { jb0 = grid_start(G, row, b0); lb = grid_start(G, row, X1 + 1) - jb0; }
                    }
                }
                const int mine = la + lb;
                int inc = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
                if (lane == 63) s_wave_tot[wid] = inc;
                __syncthreads();
                int wbase = 0, total = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) { const int v = s_wave_tot[w]; if (w < wid) wbase += v; total += v; }
                const int ex = wbase + inc - mine;
                s_run_j0[2 * tid] = ja0; s_run_off[2 * tid] = ex;
                s_run_j0[2 * tid + 1] = jb0; s_run_off[2 * tid + 1] = ex + la;
                if (tid == 0) s_run_off[NRUN] = total;
                __syncthreads();
                for (int tbase = 0; tbase < total; tbase += kTileCap) {
                    const int tn = min(total - tbase, kTileCap);
                    const int tn8 = (tn + 7) & ~7;                     // the sweep consumes 4 pairs per trip
                    for (int i = tid; i < tn8; i += NT) {
                        float4 c = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __int_as_float(0x7fffffff));
                        int cr = 0x7fffffff;
                        if (i < tn) {
                            const int slot = tbase + i;
                            int lo = 0;
#pragma unroll
                            for (int step = NRUN / 2; step > 0; step >>= 1) {
                                if (s_run_off[lo + step] <= slot) lo += step;
                            }
                            const int j = s_run_j0[lo] + (slot - s_run_off[lo]);
                            c = G.sorted[j];
                            cr = G.sring[j];
                        }
                        const int pr = i >> 1, hb = i & 1;
                        s_xy_f[4 * pr + hb] = c.x; s_xy_f[4 * pr + 2 + hb] = c.y;
                        s_zg_f[4 * pr + hb] = c.z; s_zg_f[4 * pr + 2 + hb] = c.w;
                        s_ring[i] = cr;
                    }
                    __syncthreads();
                    const int nquads = tn8 >> 3;                       // groups of 4 pairs
                    const int per = (nquads + NW - 1) / NW;
                    const int g0 = wid * per, g1 = min(g0 + per, nquads);
                    if (member) {
                        for (int gq = g0; gq < g1; gq++) {
                            const int pi = gq * 4;
                            float4 a[4], bq[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) { a[u] = s_xy[pi + u]; bq[u] = s_zg[pi + u]; }
                            f32x2 d2[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const f32x2 cxp = {a[u].x, a[u].y}, cyp = {a[u].z, a[u].w}, czp = {bq[u].x, bq[u].y};
                                const f32x2 dx = qx2 - cxp, dy = qy2 - cyp, dz = qz2 - czp;
                                f32x2 d = dx * dx;                     // x -> y -> z accumulation, no FMA (-ffp-contract=off)
                                d = d + dy * dy;
                                d = d + dz * dz;
                                d2[u] = d;
                            }
                            const float dmin = fminf(fminf(fminf(d2[0].x, d2[0].y), fminf(d2[1].x, d2[1].y)), fminf(fminf(d2[2].x, d2[2].y), fminf(d2[3].x, d2[3].y)));
                            if (dmin <= t.b2d) {
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    if (d2[u].x <= t.b2d) {
                                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2[u].x) << 32) | (unsigned)__float_as_int(bq[u].z);
                                        top2_update(t, key, s_ring[2 * (pi + u)]);
                                    }
                                    if (d2[u].y <= t.b2d) {
                                        const unsigned long long key = ((unsigned long long)__float_as_uint(d2[u].y) << 32) | (unsigned)__float_as_int(bq[u].w);
                                        top2_update(t, key, s_ring[2 * (pi + u) + 1]);
                                    }
                                }
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            if (NW > 1) {
                m1[wid][lane] = t.b1; m2[wid][lane] = t.b2; mr[wid][lane] = t.b1ring;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    if (w == wid) continue;
                    const unsigned long long c1 = m1[w][lane], c2 = m2[w][lane];
                    if (c1 < t.b2) top2_update(t, c1, mr[w][lane]);
                    if (c2 < t.b2) top2_update(t, c2, ring_of[(int)(unsigned)(c2 & 0xffffffffull) - out.first_point]);
                }
                __syncthreads();
            }
            const float rw = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(wave_max_i(member ? (int)__float_as_uint(t.b2d) : 0)));
            const float reach = (float)e * h_safe;
            if (reach * reach > rw) break;
            e_prev = e;
            e = max(e + 1, (int)ceilf(sqrtf(rw) / h_safe));
            if ((float)e * h_safe * ((float)e * h_safe) <= rw) e++;
        }
        // rows A3-A6 for the member lanes of this cluster (wave 0 holds the merged state like every other wave)
        if (wid == 0 && member && qi < q_end) {
            const float4 psrc = src[q_src[qi]];
            finish_correspondence(qi, psrc, qx, qy, qz, t.b1, t.b2, key_inf, tgt, tgt_off, ring_of, norm_cond, out, want_aux != 0);
        }
    }
}

#endif  // VELO_UNIT_LM_ONLY
// ---- visual blocks (rows G1, R2-R5) ------------------------------------------------------------------------------------
// One record per match, three block slots: slot 0 = 3D3D or 2D2D, slot 1 = 3D2D, slot 2 = 2D3D (velo.h order).
struct VisualMatch {      // device copy of velo_match, floats kept as floats and widened at use (costfunctions.h ctors)
    float p3_1[3], p3_2[3], p2_1[2], p2_2[2], t_cam[3];
    int cam, point1, point2;
    unsigned char d1, d2, pad[2];
};
struct VisualParams {
    double w_3d2d, w_2d2d;
    double th_3d2d, th_2d2d, th_3d3d;
    double outlier_reject;
    int enable_2d2d, enable_3d2d;
};

// pose at which a sweep evaluates
struct PoseEval {
    double t[3];
    PoseRot fwd;   // R(omega)
    PoseRot inv;   // R(-omega), cost2D3D only (costfunctions.h:154-160)
};
__device__ __forceinline__ void pose_eval_init(const double x[6], PoseEval* P, bool need_inv) {
    P->t[0] = x[3]; P->t[1] = x[4]; P->t[2] = x[5];
    pose_rot_init(x, &P->fwd);
    if (need_inv) { const double m[3] = {-x[0], -x[1], -x[2]}; pose_rot_init(m, &P->inv); }
}

// residuals r[<=3] and row-major Jacobian J[<=3][6] of one visual block; returns the residual dimension
// (WITH_2D2D = false: the caller evaluates the epipolar block itself -- visual_sweep_one, in three narrow passes -- and this function
//  must not carry the 6-wide code for it)
template <bool WITH_2D2D = true>
__device__ __forceinline__ int visual_block_eval3(const PoseEval& P, const VisualMatch& m, int slot, double& r0, double& r1, double& r2, double J[18]) {
    // one exit, residuals carried as scalars: written as an array on several branches they end up on the stack
    r0 = 0.0; r1 = 0.0; r2 = 0.0;
    int d;
    if (slot == 0) {
        if (!WITH_2D2D || (m.d1 && m.d2)) {
            const double a[3] = {m.p3_1[0], m.p3_1[1], m.p3_1[2]}, s[3] = {m.p3_2[0], m.p3_2[1], m.p3_2[2]};
            double rl[3];
            res_3d3d(P.fwd, P.t, a, s, rl, J);
            r0 = rl[0]; r1 = rl[1]; r2 = rl[2]; d = 3;
        } else {
            const double a[2] = {m.p2_1[0], m.p2_1[1]}, s[2] = {m.p2_2[0], m.p2_2[1]}, tc[3] = {m.t_cam[0], m.t_cam[1], m.t_cam[2]};
            double rl[1];
            res_2d2d(P.fwd, P.t, a, s, tc, rl, J);
            r0 = rl[0]; d = 1;
        }
    } else {
        const double tc[3] = {m.t_cam[0], m.t_cam[1], m.t_cam[2]};
        double rl[2];
        if (slot == 1) {
            const double a[3] = {m.p3_1[0], m.p3_1[1], m.p3_1[2]}, s[2] = {m.p2_2[0], m.p2_2[1]};
            res_3d2d(P.fwd, P.t, a, s, tc, rl, J);
        } else {
            const double a[3] = {m.p3_2[0], m.p3_2[1], m.p3_2[2]}, s[2] = {m.p2_1[0], m.p2_1[1]};
            res_2d3d(P.inv, P.t, a, s, tc, rl, J);
        }
        r0 = rl[0]; r1 = rl[1]; d = 2;
    }
    return d;
}
__device__ __forceinline__ int visual_block_eval(const PoseEval& P, const VisualMatch& m, int slot, double r[3], double J[18]) {
    double r0, r1, r2;
    const int d = visual_block_eval3(P, m, slot, r0, r1, r2, J);
    r[0] = r0; r[1] = r1; r[2] = r2;
    return d;
}

// flags[3*m + slot]: 0 = no block, 1 + residual_type = block present
// counts (chain mode, may be null): [0] += blocks, [1] += residuals selected here -- the host reads them once, at the end of the call
__device__ __forceinline__ void visual_gate_body(const double* __restrict__ xdev, const VisualParams& V, const VisualMatch* __restrict__ matches, int n, int iter,
                                                 unsigned char* __restrict__ flags, int* __restrict__ counts, const int i) {
    if (i >= n) return;
    double x[6];
#pragma unroll
    for (int k = 0; k < 6; k++) x[k] = xdev[k];
    PoseEval P;
    pose_eval_init(x, &P, true);
    const VisualMatch m = matches[i];
    unsigned char f0 = 0, f1 = 0, f2 = 0;
    bool stop = false;          // a failed gate `continue`s past the remaining residual types of the match
    double r[3], J[18];
    const double it = (double)iter;
    if (m.d1 && m.d2) {                                                      // velo.h:662-693
        visual_block_eval(P, m, 0, r, J);
        const double th2 = V.th_3d3d * V.outlier_reject / it * V.th_3d3d * V.outlier_reject / it;      // velo.h:674-681, left to right as written
        if (iter > 1 && r[0] * r[0] + r[1] * r[1] + r[2] * r[2] > th2) stop = true; else f0 = 1 + 0;
    }
    if (!stop && !m.d1 && !m.d2 && V.enable_2d2d) {                          // velo.h:694-722
        visual_block_eval(P, m, 0, r, J);
        if (iter > 1 && fabs(r[0]) > V.th_2d2d * V.outlier_reject / it) stop = true; else f0 = 1 + 3;
    }
    if (V.enable_3d2d) {
        const double th2 = V.th_3d2d * V.outlier_reject / it * V.th_3d2d * V.outlier_reject / it;      // velo.h:739-742,772-775
        if (!stop && m.d1) {                                                 // velo.h:724-756
            visual_block_eval(P, m, 1, r, J);
            if (iter > 1 && r[0] * r[0] + r[1] * r[1] > th2) stop = true; else f1 = 1 + 1;
        }
        if (!stop && m.d2) {                                                 // velo.h:757-789
            visual_block_eval(P, m, 2, r, J);
            if (iter > 1 && r[0] * r[0] + r[1] * r[1] > th2) stop = true; else f2 = 1 + 2;
        }
    }
    flags[3 * i + 0] = f0; flags[3 * i + 1] = f1; flags[3 * i + 2] = f2;
    if (counts) {
        const int nb = (f0 ? 1 : 0) + (f1 ? 1 : 0) + (f2 ? 1 : 0);
        const int nr = (f0 ? (f0 - 1 == VELO_RESIDUAL_3D3D ? 3 : 1) : 0) + (f1 ? 2 : 0) + (f2 ? 2 : 0);     // f0: 3D3D (3 rows) or 2D2D (1 row)
        if (nb) { atomicAdd(&counts[0], nb); atomicAdd(&counts[1], nr); }
    }
}
__global__ void visual_gate_kernel(const double* __restrict__ xdev, VisualParams V, const VisualMatch* __restrict__ matches, int n, int iter,
                                   unsigned char* __restrict__ flags, int* __restrict__ counts)
#if VELO_DEF_LMA
{
    visual_gate_body(xdev, V, matches, n, iter, flags, counts, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}
#else
;
#endif
// the gates of all contexts of a lock-step group in one launch (a queue operation per context and f2f iteration less)
constexpr int kGateJobs = 8;
struct GateBatch {
    const double* x[kGateJobs]; const VisualMatch* m[kGateJobs]; unsigned char* flags[kGateJobs]; int* counts[kGateJobs]; int n[kGateJobs];
    VisualParams V; int iter;
};
__global__ void __launch_bounds__(128) visual_gate_batch_kernel(GateBatch G)
#if VELO_DEF_LMA
{
    const int j = blockIdx.y;
    visual_gate_body(G.x[j], G.V, G.m[j], G.n[j], G.iter, G.flags[j], G.counts[j], (int)(blockIdx.x * 128 + threadIdx.x));
}
#else
;
#endif

// ---- LM state (row S1) ------------------------------------------------------------------------------------------------------
enum { PHASE_INIT = 0, PHASE_CAND = 1 };
struct LMParams {
    int max_num_iterations, max_invalid;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius, min_relative_decrease, min_diag, max_diag;
};
struct LMState {
    double x[6];        // last accepted iterate (the reference's `transform`)
    double xc[6];       // candidate x + delta
    double H[21];       // upper triangle of J^T J at x (robustified, unscaled)
    double g[6];        // J^T r at x
    double cost;
    double scale[6];    // Jacobi scaling 1/(1+||J_j||), fixed per solve
    double diag[6];
    double radius, decrease, x_norm, model_change, step_norm;
    double initial_cost;
    int reuse_diag, invalid, iter, evals, done, termination, phase, n_valid;   // n_valid: copy of the association's counter
};
static_assert(sizeof(LMState) % 8 == 0 && sizeof(LMState) / 8 <= 64, "the state is copied by one wave, one 8-byte word per lane");

__device__ __forceinline__ int tri(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }   // i <= j

// ---- evaluation sweeps ---------------------------------------------------------------------------------------------------------
// The point a sweep evaluates at, as ONE block the LM step (or lm_begin) leaves behind for the sweep workgroups: the pose, the
// rotation R(omega) and its three partials dR/d omega_j as plain 3x3 matrices (R p and its derivative are linear in the point p, so a
// residual row is four matrix-vector products instead of a dual-number pass through Rodrigues' formula per row), and the done
// flag.  A sweep workgroup fetches it with one cooperative load -- no chain of dependent reads through the LM state, no
// sqrt / sin / cos per workgroup.  The matrices are built by pushing the unit vectors through the SAME dual-number rotation
// (eval_point_column), so they are the autodiff values, not a re-derivation.
struct LMEvalPoint {
    double M[4][9];     // row-major 3x3: R, dR/dw0, dR/dw1, dR/dw2
    double t[3];
    double x[6];
    int done;           // LMState::done at the time the block was written
    int pad;
};
static_assert(sizeof(LMEvalPoint) % 8 == 0 && sizeof(LMEvalPoint) / 8 <= 64, "the eval point is copied by one wave, one 8-byte word per lane");

// lane `col` (0..2) fills column `col` of the four matrices; lane 3 the rest
__device__ __forceinline__ void eval_point_column(const double x[6], int done, int col, LMEvalPoint* P) {
    if (col < 3) {
        PoseRot R;
        pose_rot_init(x, &R);
        const double e[3] = {col == 0 ? 1.0 : 0.0, col == 1 ? 1.0 : 0.0, col == 2 ? 1.0 : 0.0};
        D3 m[3];
        rotate_point(R, e, m);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            P->M[0][3 * k + col] = m[k].a;
#pragma unroll
            for (int j = 0; j < 3; j++) P->M[1 + j][3 * k + col] = m[k].v[j];
        }
    } else if (col == 3) {
#pragma unroll
        for (int k = 0; k < 3; k++) P->t[k] = x[3 + k];
#pragma unroll
        for (int k = 0; k < 6; k++) P->x[k] = x[k];
        P->done = done; P->pad = 0;
    }
}

// R1 cost3DPD (costfunctions.h:39-54) through the matrices of the eval point: r = N . (R p + t - v0), dr/dw_j = N . (dR/dw_j p)
__device__ __forceinline__ void res_3dpd_mat(const double (*M)[9], const double t[3], const double p[3], const double n[3], const double v0[3],
                                             double* r, double J[6]) {
    double m[3];
#pragma unroll
    for (int k = 0; k < 3; k++) m[k] = (M[0][3 * k] * p[0] + M[0][3 * k + 1] * p[1] + M[0][3 * k + 2] * p[2]) + (t[k] - v0[k]);
    *r = m[0] * n[0] + m[1] * n[1] + m[2] * n[2];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double d[3];
#pragma unroll
        for (int k = 0; k < 3; k++) d[k] = M[1 + j][3 * k] * p[0] + M[1 + j][3 * k + 1] * p[1] + M[1 + j][3 * k + 2] * p[2];
        J[j] = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];
    }
    J[3] = n[0]; J[4] = n[1]; J[5] = n[2];
}

struct EvalArgs {
    const LMState* __restrict__ state;      // the LM state (trace index only; the sweeps read the eval point)
    const LMEvalPoint* __restrict__ pt;     // where to evaluate (written by lm_begin / the LM step)
    const double* __restrict__ x_override;  // when non-null: evaluate at this x instead (velo_evaluate; LDS x of the one-launch solve)
    const float4* __restrict__ cp;          // correspondences
    const float4* __restrict__ cn;
    const float4* __restrict__ cv0;
    int q_begin, q_end;
    const VisualMatch* __restrict__ vm;
    const unsigned char* __restrict__ vflags;
    int n_matches;                          // 0 on ranks that do not own the visual blocks
    double loss_a_3dpd, w_3dpd;
    VisualParams V;
    double* __restrict__ partials;          // ICP sweep: rows [0, gridDim.x); visual sweep: rows [vis_row0, vis_row0 + gridDim.x)
    int vis_row0;
    double* __restrict__ rows_r;            // optional row output
    double* __restrict__ rows_J;
    const int* __restrict__ row_offset_vis; // [3*n_matches] row index of each visual slot (exclusive scan of dims)
    const int* __restrict__ row_offset_icp; // [nq] row index of each query's block (or -1)
    unsigned long long* __restrict__ trace; // diagnostics build only (VELO_LM_TRACE): [evaluation][16 stages][first, last] s_memrealtime stamps
    int trace_eval;                         // index of this evaluation within the solve (from the host: the stamps add no loads)
};

// ---- residualStats (velo.h:921-1025): per residual type the median, mean and count of the block norms, loss NOT applied ---------
// The reference evaluates the problem without loss functions, turns every block into one number (3D3D: sqrt(r0^2 + r1^2 + r2^2);
// 3D2D / 2D3D: sqrt(r0^2 + r1^2); 2D2D and 3DPD: |r|), sorts each type's numbers and prints sorted[size / 2], sum / size, size.
// Here: one thread per block slot writes its number and type; the median is a 4-pass 16-bit radix SELECT on the bit patterns (the
// numbers are >= 0, so patterns order like values) -- no sort; sums are fixed-order block reductions (the mean can differ from the
// reference's left-to-right sum in the last bits, the median and the counts cannot).
constexpr int kStatTypes = 5;                 // VELO_RESIDUAL_3D3D, _3D2D, _2D3D, _2D2D, VELO_FUNCTOR_3DPD
constexpr int kStatBins = 65536;
struct StatWork {                             // device scratch of one statistics call
    unsigned long long prefix[kStatTypes];    // bit pattern of the median found so far (high digits)
    long long k[kStatTypes];                  // rank still to go inside the current prefix
    long long count[kStatTypes];
    double sum[kStatTypes];
    double cost;                              // 1/2 sum r^2 over all residuals (ceres::Problem::Evaluate without loss)
};
__global__ void __launch_bounds__(256)
residual_norms_kernel(const double* __restrict__ xdev, EvalArgs A, double* __restrict__ vals, signed char* __restrict__ types,
                      double* __restrict__ part /* [gridDim.x][kStatTypes + 1] */)
#if VELO_DEF_LMA
{
    __shared__ PoseEval s_P;
    __shared__ double s_red[256];
    if (threadIdx.x == 0) {
        double x[6];
        for (int k = 0; k < 6; k++) x[k] = xdev[k];
        pose_eval_init(x, &s_P, true);
    }
    __syncthreads();
    const int n_vis = 3 * A.n_matches, nq = A.q_end - A.q_begin;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0, sq = 0.0;
    int type = -1;
    if (i < n_vis) {
        const unsigned char f = A.vflags[i];
        if (f) {
            const VisualMatch m = A.vm[i / 3];
            double r0, r1, r2, J[18];
            const int d = visual_block_eval3(s_P, m, i % 3, r0, r1, r2, J);
            type = f - 1;
            if (d == 3) { sq = r0 * r0 + r1 * r1 + r2 * r2; v = sqrt(sq); }
            else if (d == 2) { sq = r0 * r0 + r1 * r1; v = sqrt(sq); }
            else { sq = r0 * r0; v = fabs(r0); }
        }
    } else if (i < n_vis + nq) {
        const int qi = A.q_begin + (i - n_vis);
        const float4 cp = A.cp[qi];
        if (__float_as_int(cp.w)) {
            const float4 cn = A.cn[qi], c0 = A.cv0[qi];
            const double p[3] = {cp.x, cp.y, cp.z}, n[3] = {cn.x, cn.y, cn.z}, v0[3] = {c0.x, c0.y, c0.z};
            double r, J[6];
            res_3dpd(s_P.fwd, s_P.t, p, n, v0, &r, J);
            type = VELO_FUNCTOR_3DPD; sq = r * r; v = fabs(r);
        }
    }
    if (i < n_vis + nq) { vals[i] = v; types[i] = (signed char)type; }
    // fixed-order block sums: per type, then the cost
    for (int t = 0; t <= kStatTypes; t++) {
        s_red[threadIdx.x] = (t < kStatTypes) ? (type == t ? v : 0.0) : sq;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off]; __syncthreads(); }
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * (kStatTypes + 1) + t] = s_red[0];
        __syncthreads();
    }
}
#else
;
#endif
// pass p (0..3): histogram of digit p (16 bits, from the top) over the elements whose higher digits equal the prefix found so far
__global__ void __launch_bounds__(256)
stats_hist_kernel(const double* __restrict__ vals, const signed char* __restrict__ types, int n, int pass, const StatWork* __restrict__ W, int* __restrict__ hist)
#if VELO_DEF_LMA
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = types[i];
    if (t < 0) return;
    const unsigned long long b = (unsigned long long)__double_as_longlong(vals[i]);
    if (pass > 0 && (b >> (64 - 16 * pass)) != W->prefix[t]) return;
    atomicAdd(&hist[(size_t)t * kStatBins + (int)((b >> (48 - 16 * pass)) & 0xffffull)], 1);
}
#else
;
#endif
// one workgroup per type: the bin that holds rank k; pass 0 also derives count and k = count / 2.  Clears the bins behind itself.
__global__ void __launch_bounds__(256)
stats_pick_kernel(int pass, StatWork* __restrict__ W, int* __restrict__ hist)
#if VELO_DEF_LMA
{
    __shared__ long long s_tot[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    int* h = hist + (size_t)t * kStatBins;
    constexpr int per = kStatBins / 256;
    long long mine = 0;
    for (int u = 0; u < per; u++) mine += h[tid * per + u];
    s_tot[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        long long total = 0;
        for (int u = 0; u < 256; u++) total += s_tot[u];
        if (pass == 0) { W->count[t] = total; W->k[t] = total / 2; W->prefix[t] = 0ull; }
        long long k = W->k[t];
        if (total > 0) {
            int seg = 0;
            while (seg < 255 && k >= s_tot[seg]) { k -= s_tot[seg]; seg++; }
            int bin = seg * per;
            while (bin < seg * per + per - 1 && k >= h[bin]) { k -= h[bin]; bin++; }
            W->prefix[t] = (W->prefix[t] << 16) | (unsigned long long)bin;
            W->k[t] = k;
        }
    }
    __syncthreads();
    for (int u = 0; u < per; u++) h[tid * per + u] = 0;
}
#else
;
#endif
__global__ void stats_final_kernel(const double* __restrict__ part, int n_blocks, StatWork* __restrict__ W, velo_residual_stats* __restrict__ out)
#if VELO_DEF_LMA
{
    const int t = threadIdx.x;
    if (t > kStatTypes) return;
    double s = 0.0;
    for (int b = 0; b < n_blocks; b++) s += part[(size_t)b * (kStatTypes + 1) + t];
    if (t == kStatTypes) { out->cost = 0.5 * s; return; }
    const long long n = W->count[t];
    out->type[t].count = n;
    out->type[t].mean = n > 0 ? s / (double)n : 0.0;
    out->type[t].median = n > 0 ? __longlong_as_double((long long)W->prefix[t]) : 0.0;
}
#else
;
#endif

// Time line of the LM chain (tools/lm_trace.py): every workgroup's thread 0 stamps the stages it passes with the 100 MHz real-time
// counter; per evaluation and stage the buffer keeps the first and the last stamp.  Compiled only into the tools' build.
#ifdef VELO_DIAGNOSTICS
constexpr int kTraceMaxEvals = 64, kTraceStages = 16, kTraceWgs = 512;    // a slot per (evaluation, stage, workgroup): plain stores, no atomics
__device__ __forceinline__ void lm_trace(unsigned long long* trace, int eval, int stage) {
    if (trace && threadIdx.x == 0 && eval < kTraceMaxEvals && (int)blockIdx.x < kTraceWgs)
        trace[((size_t)eval * kTraceStages + stage) * kTraceWgs + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}
#define VELO_LM_TRACE(trace, eval, stage) lm_trace(trace, eval, stage)
#else
#define VELO_LM_TRACE(trace, eval, stage) do { } while (0)
#endif

// acc += (sr J)^T (sr J), (sr J)^T (sr r): one robustified residual row
__device__ __forceinline__ void accumulate_row(double acc[kNumAcc], double r, const double Jr[6], double sr) {
    const double rk = r * sr;
    double J[6];
#pragma unroll
    for (int i = 0; i < 6; i++) J[i] = Jr[i] * sr;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int j = i; j < 6; j++) acc[k++] += J[i] * J[j];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) acc[21 + i] += J[i] * rk;
}

// Workgroup reduction of the 28 accumulators, laid out for latency (the LM chain waits for it): exchanges with the lanes 32 and 16
// away (2 x 28 independent shuffles), the 16 x 4 partial sums of every accumulator through LDS ([accumulator][64], conflict-free
// 8-byte stores), eight threads per accumulator add 8 values each, one thread per accumulator adds those eight and stores.
// Two barriers; 14 KB of LDS, so that a sweep workgroup fits next to four association workgroups.  Fixed order -> deterministic.
constexpr int kScratchDoubles = 128 * kNumAcc;                    // 28,672 bytes: the sweep's reduction and the step's chunk of partial rows share it
#ifndef VELO_REDUCE_LDS
#define VELO_REDUCE_LDS 1
#endif
#if !VELO_REDUCE_LDS
// A/B: the halving butterfly (1 KB of LDS)
__device__ __forceinline__ void block_reduce_store(double acc[kNumAcc], double* __restrict__ dst /* [28] */, double* __restrict__) {
    __shared__ double red[kEvalThreads / kWave][32];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double v[32];
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] = (k < kNumAcc) ? acc[k] : 0.0;
#pragma unroll
    for (int n = 16, mask = 32; n >= 1; n >>= 1, mask >>= 1) {
        const bool hi = (lane & mask) != 0;
#pragma unroll
        for (int i = 0; i < n; i++) {
            const double mine = hi ? v[i + n] : v[i];
            const double send = hi ? v[i] : v[i + n];
            v[i] = mine + __shfl_xor(send, mask);
        }
    }
    v[0] += __shfl_xor(v[0], 1);
    const int idx = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    if ((lane & 1) == 0) red[wid][idx] = v[0];
    __syncthreads();
    if (threadIdx.x < kNumAcc) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kEvalThreads / kWave; w++) t += red[w][threadIdx.x];
        dst[threadIdx.x] = t;
    }
}
#else
// Agent-scope accesses for data one workgroup hands to another INSIDE a launch (the partial rows of the fused sweep + step): the
// store is written through to where every XCD sees it, the load does not stop at an L2 line another XCD's store has made stale.
// No release/acquire fence anywhere on that path -- on this chip an agent-scope fence writes back / invalidates a whole L2.
__device__ __forceinline__ void store_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double load_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// tid_in (here and below): the thread's index as the CALLER holds it -- the all-gather solve hands in a copy that is opaque per loop
// iteration, so that the index arithmetic derived from it is recomputed inside its loop instead of being hoisted out of it (hoisted,
// ~100 thread-derived values stay live across the whole loop: 260 registers instead of 126); -1 = threadIdx.x
template <bool COHERENT = false>
__device__ __forceinline__ void block_reduce_store(double acc[kNumAcc], double* __restrict__ dst /* [28] */, double* __restrict__ scratch, const int tid_in = -1) {
    constexpr int kCols = kEvalThreads / 4;                       // 64 partial sums per accumulator
    static_assert(kNumAcc * kCols <= kScratchDoubles / 2, "scratch size");
    double (*red)[kCols] = reinterpret_cast<double (*)[kCols]>(scratch);
    __shared__ double red2[kNumAcc][8];
    const int t = tid_in < 0 ? (int)threadIdx.x : tid_in, lane = t & 63, wid = t >> 6;
#pragma unroll
    for (int k = 0; k < kNumAcc; k++) acc[k] += __shfl_xor(acc[k], 32);
#pragma unroll
    for (int k = 0; k < kNumAcc; k++) acc[k] += __shfl_xor(acc[k], 16);
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < kNumAcc; k++) red[k][wid * 16 + lane] = acc[k];
    }
    __syncthreads();
    if (t < kNumAcc * 8) {
        const int k = t >> 3, part = t & 7;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < kCols / 8; i++) v += red[k][part * (kCols / 8) + i];
        red2[k][part] = v;
    }
    __syncthreads();
    if (t < kNumAcc) {
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 8; i++) v += red2[t][i];
        if (COHERENT) store_agent(dst + t, v); else dst[t] = v;
    }
}
#endif

// The eval point into LDS: one cooperative copy of the block the LM step left (or, for an explicit x, built here by four
// lanes).  Returns false when the solve is already done (every thread of the workgroup gets the same answer).
__device__ __forceinline__ bool eval_point_load(const EvalArgs& A, LMEvalPoint* s_pt) {
    if (A.x_override) {
        if (threadIdx.x < 4) {
            double x[6];
#pragma unroll
            for (int k = 0; k < 6; k++) x[k] = A.x_override[k];
            eval_point_column(x, 0, threadIdx.x, s_pt);
        }
    } else if (threadIdx.x < sizeof(LMEvalPoint) / 8) {
        reinterpret_cast<unsigned long long*>(s_pt)[threadIdx.x] = reinterpret_cast<const unsigned long long*>(A.pt)[threadIdx.x];
    }
    __syncthreads();
    return A.x_override || !s_pt->done;
}

// point-to-plane blocks (row R1 + Scaled(Cauchy(loss_thresh_3DPD), weight_3DPD), velo.h:875-892)
// The correspondences do not depend on the pose: the first kPre strided rows of a thread are requested at the very start of the
// kernel, BEFORE the eval point (or, in the one-launch iteration, the whole LM step) is dealt with, so the latencies overlap.
constexpr int kPre = 4;
template <int PRE> struct RowPrefetchT { float4 p[PRE], n[PRE], v[PRE]; };
using RowPrefetch = RowPrefetchT<kPre>;
template <int PRE = kPre>
__device__ __forceinline__ RowPrefetchT<PRE> prefetch_rows(const EvalArgs& A, const int bx, const int nbx, const int tid_in = -1) {
    RowPrefetchT<PRE> f;
    const int tid = bx * blockDim.x + (tid_in < 0 ? (int)threadIdx.x : tid_in), nthreads = nbx * blockDim.x;
#pragma unroll
    for (int k = 0; k < PRE; k++) {
        const int i = min(A.q_begin + tid + k * nthreads, A.q_end - 1);
        f.p[k] = A.cp[i]; f.n[k] = A.cn[i]; f.v[k] = A.cv0[i];
    }
    return f;
}
// one point-to-plane row into the accumulators
template <bool M_FROM_LDS>
__device__ __forceinline__ void sweep_row(const EvalArgs& A, const LMEvalPoint& s_pt, const double (*Mreg)[9], const double t[3], const float4& p, const float4& n,
                                          const float4& v, const int i, double acc[kNumAcc]) {
    if (__float_as_int(p.w) == 0) return;
    // M_FROM_LDS (the sweep kernels that must run two waves per SIMD next to other pairs' kernels): the four matrices are read
    // from LDS for every row (uniform addresses: broadcast reads) instead of living in 72 registers across the loop -- with them,
    // the 56 accumulator registers and the prefetched rows the kernel would need scratch memory, and a kernel that touches scratch
    // at all costs ~11 us more per launch on this system.  The pointer is made opaque so that the loads are not hoisted out of the
    // loop again.  The one-launch iteration (one wave per SIMD, registers to spare) keeps them in registers: 3.2 vs 4.7 us of rows.
    const double (*M)[9] = M_FROM_LDS ? s_pt.M : Mreg;
    if (M_FROM_LDS) asm volatile("" : "+v"(M));
    const double pd[3] = {p.x, p.y, p.z}, nd[3] = {n.x, n.y, n.z}, vd[3] = {v.x, v.y, v.z};
    double r, J[6];
    res_3dpd_mat(M, t, pd, nd, vd, &r, J);
    double rho0, rho1;
    loss_cauchy(A.loss_a_3dpd, A.w_3dpd, r * r, &rho0, &rho1);
    acc[27] += 0.5 * rho0;
    const double sr = sqrt(rho1);
    accumulate_row(acc, r, J, sr);
    if (A.rows_r) {
        const int row = A.row_offset_icp[i];
        A.rows_r[row] = r * sr;
#pragma unroll
        for (int k = 0; k < 6; k++) A.rows_J[(size_t)row * 6 + k] = J[k] * sr;
    }
}
// this workgroup's rows at the eval point in LDS -> the 28 accumulators of every thread (the prefetched rows first, with
// compile-time indices, then whatever is left)
template <bool M_FROM_LDS, int PRE = kPre>
__device__ __forceinline__ void sweep_rows(const EvalArgs& A, const RowPrefetchT<PRE>& f, const LMEvalPoint& s_pt, const int bx, const int nbx, double acc[kNumAcc], const int tid_in = -1) {
    const int tid = bx * blockDim.x + (tid_in < 0 ? (int)threadIdx.x : tid_in), nthreads = nbx * blockDim.x;
    double Mreg[4][9];
    if (!M_FROM_LDS) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int k = 0; k < 9; k++) Mreg[a][k] = s_pt.M[a][k];
        }
    }
    double t[3];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = s_pt.t[k];
#pragma unroll
    for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < PRE; k++) {
        const int i = A.q_begin + tid + k * nthreads;
        if (i < A.q_end) sweep_row<M_FROM_LDS>(A, s_pt, Mreg, t, f.p[k], f.n[k], f.v[k], i, acc);
    }
    for (int i = A.q_begin + tid + PRE * nthreads; i < A.q_end; i += nthreads) sweep_row<M_FROM_LDS>(A, s_pt, Mreg, t, A.cp[i], A.cn[i], A.cv0[i], i, acc);
}

__device__ __forceinline__ void eval_icp_body(const EvalArgs& A, const int bx, const int nbx) {
    const RowPrefetch f = prefetch_rows(A, bx, nbx);
    VELO_LM_TRACE(A.trace, A.trace_eval, 0);
    __shared__ LMEvalPoint s_pt;
    if (!eval_point_load(A, &s_pt)) return;
    VELO_LM_TRACE(A.trace, A.trace_eval, 1);
    double acc[kNumAcc];
    sweep_rows<true>(A, f, s_pt, bx, nbx, acc);
    VELO_LM_TRACE(A.trace, A.trace_eval, 2);
#if VELO_REDUCE_LDS
    __shared__ double s_scratch[kScratchDoubles / 2];            // 64 columns x 28: 14 KB (the sweep must fit next to association workgroups)
#else
    double* s_scratch = nullptr;
#endif
    block_reduce_store(acc, A.partials + (size_t)bx * kNumAcc, s_scratch);
    VELO_LM_TRACE(A.trace, A.trace_eval, 3);
}

__global__ void __launch_bounds__(kEvalThreads)
eval_icp_kernel(EvalArgs A)
#if VELO_DEF_LMA
{ eval_icp_body(A, blockIdx.x, gridDim.x); }
#else
;
#endif

// ---- several contexts in one launch (velo_frame_to_frame_batch): blockIdx.y = context -------------------------------------------
// Each context keeps its own state, partial sums and correspondence table; the batch only shares LAUNCHES, so a sweep over 8
// contexts is one kernel of ~950 workgroups instead of 8 kernels of 118 that each leave half the chip idle.
struct LMBatchItem {
    EvalArgs A;            // A.state / A.partials are this context's
    LMState* S;
    const double* xd;      // x to start the solve from (device)
    const int* n_valid;    // association counter of this round
    int nb_icp;            // workgroups of this context's point-to-plane sweep
    int nb_vis;            // workgroups of its visual sweep (A.vis_row0 = nb_icp)
    int n_rows;            // rows of partial sums the LM step reduces
    PoseRecord* pose_out;  // chain mode (or null): where the step that finishes the solve leaves the next round's pose scalars
    SolveLog* log;         //   "          and the summary of the solve
};
__global__ void __launch_bounds__(kEvalThreads)
eval_icp_batch_kernel(const LMBatchItem* __restrict__ items)
#if VELO_DEF_LMA
{
    const LMBatchItem& it = items[blockIdx.y];
    if ((int)blockIdx.x >= it.nb_icp) return;
    eval_icp_body(it.A, blockIdx.x, it.nb_icp);
}
#else
;
#endif
// visual blocks (rows R2-R5; losses velo.h:688,714-717,748-751,781-784)
// the visual blocks of workgroup bx of nbx at the eval point pt (in LDS), summed into acc[28]
__device__ __forceinline__ void visual_sweep_acc(const EvalArgs& A, const LMEvalPoint& pt, const int bx, const int nbx, double acc[kNumAcc]) {
    // the rotation constants of R(omega) and R(-omega) once per workgroup, read from LDS where a block needs them (in registers
    // they, the 56 accumulator registers and the 6-wide duals of the epipolar block push the kernel into scratch memory)
    __shared__ PoseEval s_P;
    if (threadIdx.x == 0) {
        double x[6];
#pragma unroll
        for (int k = 0; k < 6; k++) x[k] = pt.x[k];
        pose_eval_init(x, &s_P, true);
    }
    __syncthreads();
    const PoseEval& P = s_P;
#pragma unroll
    for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
    const int tid = bx * blockDim.x + threadIdx.x, nthreads = nbx * blockDim.x;
    for (int s = tid; s < 3 * A.n_matches; s += nthreads) {
        const unsigned char f = A.vflags[s];
        if (!f) continue;
        const int mi = s / 3, slot = s - 3 * mi;
        const VisualMatch m = A.vm[mi];
        // residuals as three scalars and every use with compile-time indices: an array written by the four block types on
        // different branches ends up on the stack, and a kernel with any stack at all pays ~11 us per launch
        double r0, r1, r2, J[18];
        const int d = visual_block_eval3(P, m, slot, r0, r1, r2, J);
        const double sq = r0 * r0 + (d > 1 ? r1 * r1 : 0.0) + (d > 2 ? r2 * r2 : 0.0);
        double rho0, rho1;
        const int type = f - 1;
        if (type == 0) loss_arctan(A.V.th_3d3d, 1.0, sq, &rho0, &rho1);
        else if (type == 3) loss_arctan(A.V.th_2d2d, A.V.w_2d2d, sq, &rho0, &rho1);
        else loss_arctan(A.V.th_3d2d, A.V.w_3d2d, sq, &rho0, &rho1);
        acc[27] += 0.5 * rho0;
        const double sr = sqrt(rho1);
        accumulate_row(acc, r0, J, sr);
        if (d > 1) accumulate_row(acc, r1, J + 6, sr);
        if (d > 2) accumulate_row(acc, r2, J + 12, sr);
        if (A.rows_r) {
            const int row = A.row_offset_vis[s];
            A.rows_r[row] = r0 * sr;
            if (d > 1) A.rows_r[row + 1] = r1 * sr;
            if (d > 2) A.rows_r[row + 2] = r2 * sr;
#pragma unroll
            for (int c = 0; c < 18; c++) if (c < 6 * d) A.rows_J[(size_t)row * 6 + c] = J[c] * sr;
        }
    }
}
// The same sweep for launches in which every thread holds AT MOST ONE block slot (nbx x 256 >= 3 n_matches: up to 5,461 matches with
// kMaxVisBlocks workgroups) and registers are short -- the visual workgroups of the lean sweep + step kernel.  The block is evaluated
// BEFORE the 28 accumulators exist (56 registers that visual_sweep_acc carries through its loop), the epipolar block in three passes
// of two pose parameters (res_2d2d_win: a third of the 6-wide duals' registers); then the accumulators are zeroed and take the
// block's rows in visual_sweep_acc's order -- the same additions on the same values: the partial row is bit-identical.
__device__ __forceinline__ void visual_sweep_one(const EvalArgs& A, const LMEvalPoint& pt, const int bx, const int nbx, double acc[kNumAcc]) {
    __shared__ PoseEval s_P;
    if (threadIdx.x == 0) {
        double x[6];
#pragma unroll
        for (int k = 0; k < 6; k++) x[k] = pt.x[k];
        pose_eval_init(x, &s_P, true);
    }
    __syncthreads();
    const PoseEval& P = s_P;
    const int s = bx * blockDim.x + threadIdx.x;
    const unsigned char f = s < 3 * A.n_matches ? A.vflags[s] : (unsigned char)0;
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, J[18], sr = 0.0, rho0 = 0.0;
    int d = 0;
#pragma unroll
    for (int c = 0; c < 18; c++) J[c] = 0.0;
    if (f) {
        const int mi = s / 3, slot = s - 3 * mi;
        const VisualMatch m = A.vm[mi];
        if (slot == 0 && !(m.d1 && m.d2)) {                            // the epipolar block, two columns of the Jacobian at a time
            const double a[2] = {m.p2_1[0], m.p2_1[1]}, q[2] = {m.p2_2[0], m.p2_2[1]}, tc[3] = {m.t_cam[0], m.t_cam[1], m.t_cam[2]};
            double rr;
            res_2d2d_win<2, 0>(P.fwd, P.t, a, q, tc, &r0, J);
            res_2d2d_win<2, 2>(P.fwd, P.t, a, q, tc, &rr, J + 2);
            res_2d2d_win<2, 4>(P.fwd, P.t, a, q, tc, &rr, J + 4);
            d = 1;
        } else d = visual_block_eval3<false>(P, m, slot, r0, r1, r2, J);
        const double sq = r0 * r0 + (d > 1 ? r1 * r1 : 0.0) + (d > 2 ? r2 * r2 : 0.0);
        double rho1;
        const int type = f - 1;
        if (type == 0) loss_arctan(A.V.th_3d3d, 1.0, sq, &rho0, &rho1);
        else if (type == 3) loss_arctan(A.V.th_2d2d, A.V.w_2d2d, sq, &rho0, &rho1);
        else loss_arctan(A.V.th_3d2d, A.V.w_3d2d, sq, &rho0, &rho1);
        sr = sqrt(rho1);
    }
#pragma unroll
    for (int k = 0; k < kNumAcc; k++) acc[k] = 0.0;
    if (f) {
        acc[27] += 0.5 * rho0;
        accumulate_row(acc, r0, J, sr);
        if (d > 1) accumulate_row(acc, r1, J + 6, sr);
        if (d > 2) accumulate_row(acc, r2, J + 12, sr);
        if (A.rows_r) {
            const int row = A.row_offset_vis[s];
            A.rows_r[row] = r0 * sr;
            if (d > 1) A.rows_r[row + 1] = r1 * sr;
            if (d > 2) A.rows_r[row + 2] = r2 * sr;
#pragma unroll
            for (int c = 0; c < 18; c++) if (c < 6 * d) A.rows_J[(size_t)row * 6 + c] = J[c] * sr;
        }
    }
}
__device__ __forceinline__ void eval_visual_body(const EvalArgs& A, const int bx, const int nbx) {
    __shared__ LMEvalPoint s_pt;
    if (!eval_point_load(A, &s_pt)) return;
    double acc[kNumAcc];
    visual_sweep_acc(A, s_pt, bx, nbx, acc);
#if VELO_REDUCE_LDS
    __shared__ double s_scratch[kScratchDoubles / 2];
#else
    double* s_scratch = nullptr;
#endif
    block_reduce_store(acc, A.partials + (size_t)(A.vis_row0 + bx) * kNumAcc, s_scratch);
}

__global__ void __launch_bounds__(kEvalThreads)
eval_visual_kernel(EvalArgs A)
#if VELO_DEF_LMA
{ eval_visual_body(A, blockIdx.x, gridDim.x); }
#else
;
#endif
__global__ void __launch_bounds__(kEvalThreads)
eval_visual_batch_kernel(const LMBatchItem* __restrict__ items)
#if VELO_DEF_LMA
{
    const LMBatchItem& it = items[blockIdx.y];
    if ((int)blockIdx.x >= it.nb_vis) return;
    eval_visual_body(it.A, blockIdx.x, it.nb_vis);
}
#else
;
#endif

// ---- one-shot all-reduce of the 28-double block through peer-mapped slabs (SURVEY.md sections 5 and 8(e)) -------------------------
// Query-sharded registration all-reduces 224 bytes per LM evaluation: pure latency.  Every rank owns a small slab in fine-grained
// device memory that all its peers have mapped (hipIpc handles, exchanged once by the host program); an all-reduce is: write my
// 28 doubles into MY slot of EVERY rank's slab (one store per value and peer, over xGMI for the remote ones), then my sequence
// number into the flag of that slot, wait until the flags of all slots of my own slab carry the sequence number, and add the
// slots in rank order -- every rank adds the same values in the same order, so all ranks hold bit-identical sums and take
// identical LM decisions.  No collective library call, no extra kernel: it runs inside the LM step.  The slab is double-buffered
// by the parity of the sequence number (a rank can be at most one all-reduce ahead of a peer that is still reading).
// All slab traffic is system-scope (sc0 sc1: bypasses the caches of the issuing device); waits are bounded.
constexpr int kMaxPeers = 8;
struct PeerSlab {
    double data[2][kMaxPeers][32];             // [parity][writer rank][value]
    unsigned long long flag[2][kMaxPeers];     // sequence number the writer's block belongs to
    unsigned long long xflag[2][kMaxPeers];    // the same for the record exchange of the target-sharded mode
    int kval[2][kMaxPeers][64];                // launch-count agreement of a chained call (peer_agree_kernel): [parity][writer rank][solve]
    unsigned long long kflag[2][kMaxPeers];    // its sequence numbers
};
struct PeerComm {
    PeerSlab* slab[kMaxPeers];                 // slab[r]: rank r's slab as mapped into this process (slab[rank]: my own)
    unsigned long long* seq;                   // my all-reduce counter (device memory, starts at 0)
    unsigned long long* kseq;                  // my agreement counter (device memory, starts at 0)
    int* error;                                // set to 1 when a wait ran into its time limit
    int rank, world;
};
__device__ __forceinline__ void sys_store(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sys_load(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
// in/out: 28 doubles in LDS; called by all 256 threads of a workgroup (uniformly); returns with the workgroup synchronised
__device__ __forceinline__ void peer_allreduce28(const PeerComm& C, double* __restrict__ v28) {
    __shared__ double s_peer[kMaxPeers][32];
    const int t = threadIdx.x, p = t >> 5, k = t & 31;
    const unsigned long long seq = *C.seq + 1ull;
    const int par = (int)(seq & 1ull);
    if (p < C.world && k < kNumAcc) sys_store(&C.slab[p]->data[par][C.rank][k], v28[k]);
    __threadfence_system();                                  // every storing thread: its values are out before the barrier is passed
    __syncthreads();
    if (t < C.world) {
        __hip_atomic_store(&C.slab[t]->flag[par][C.rank], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&C.slab[C.rank]->flag[par][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { *C.error = 1; break; }   // 5 s at 100 MHz: a peer is gone
        }
    }
    __syncthreads();
    __threadfence_system();
    if (p < C.world && k < kNumAcc) s_peer[p][k] = sys_load(&C.slab[C.rank]->data[par][p][k]);
    __syncthreads();
    if (t < kNumAcc) {
        double v = 0.0;
        for (int r = 0; r < C.world; r++) v += s_peer[r][t];   // rank order: identical on every rank
        v28[t] = v;
    }
    if (t == 0) *C.seq = seq;
    __syncthreads();
}

// A chained call over peers enqueues a PREDICTED number of LM launches per solve, and the all-reduce inside the step kernel only
// completes when every rank has enqueued the same number.  The ranks therefore agree on the counts before anything is enqueued:
// every rank pushes its own prediction (from its own call history) into every peer's slab and takes the maximum over ranks --
// uniform by construction, whatever the ranks' histories are.  One wave; `out` is host-pinned memory the host reads after a sync.
struct AgreeCounts { int v[64]; };
__global__ void __launch_bounds__(64) peer_agree_kernel(PeerComm C, AgreeCounts mine, int n, int* __restrict__ out)
#if VELO_DEF_LMA
{
    const int t = threadIdx.x;
    const unsigned long long seq = *C.kseq + 1ull;
    const int par = (int)(seq & 1ull);
    if (t < C.world) {
        for (int k = 0; k < n; k++) __hip_atomic_store(&C.slab[t]->kval[par][C.rank][k], mine.v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(&C.slab[t]->kflag[par][C.rank], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&C.slab[C.rank]->kflag[par][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { *C.error = 1; break; }   // 5 s at 100 MHz: a peer is gone
        }
        __threadfence_system();
    }
    __builtin_amdgcn_wave_barrier();
    if (t < n) {
        int m = 0;
        for (int r = 0; r < C.world; r++) m = max(m, __hip_atomic_load(&C.slab[C.rank]->kval[par][r][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        __hip_atomic_store(&out[t], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (t == 0) *C.kseq = seq;
}
#else
;
#endif

#ifndef VELO_UNIT_LM_ONLY        // (the record exchange of the target-sharded mode takes the association unit's PartialRec)
// ---- target-sharded mode over the same peers (BASELINE config 5): the per-round exchange of the per-query top-2 records -------------
// Every rank has searched ITS rings for ALL queries (PartialRec per query).  Rank r must end up with everybody's records of ITS
// query share: each rank stores the slices straight into the owners' receive areas (peer-mapped, fine-grained memory; area of
// rank p: [parity][writer rank][max_share] records), then a one-workgroup kernel publishes "my slices of exchange `seq` are out"
// in every peer's slab and waits for the same word from every peer.  The receive area is double-buffered by the parity of seq:
// a fast rank may scatter round n + 1 while a slow one still merges round n.
struct PeerRecs {
    PartialRec* area[kMaxPeers];               // receive area of rank p as mapped into this process
    int rank, world, max_share;                // max_share = largest query share: slot stride inside an area
    size_t parity_stride;                      // records per parity half (= world * max_share, rounded up by the host)
};
__global__ void __launch_bounds__(256)
peer_scatter_records_kernel(const PartialRec* __restrict__ mine, int n_q, PeerRecs R, int parity)
#if VELO_DEF_LMA
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= n_q) return;
    int owner = 0, qb = 0;
    for (int p = 0; p < R.world; p++) {                      // same rule as q_range(): share p = [n_q p / W, n_q (p + 1) / W)
        const int b = (int)((long long)n_q * p / R.world), e = (int)((long long)n_q * (p + 1) / R.world);
        if (qi >= b && qi < e) { owner = p; qb = b; }
    }
    const uint4* src = reinterpret_cast<const uint4*>(mine + qi);
    uint4* dst = reinterpret_cast<uint4*>(R.area[owner] + (size_t)parity * R.parity_stride + (size_t)R.rank * R.max_share + (qi - qb));
    static_assert(sizeof(PartialRec) == 80, "five 16-byte words per record");
#pragma unroll
    for (int k = 0; k < 5; k++) dst[k] = src[k];
    __threadfence_system();
}
#else
;
#endif
__global__ void __launch_bounds__(64)
peer_exchange_sync_kernel(PeerComm C, unsigned long long seq)
#if VELO_DEF_LMA
{
    const int t = threadIdx.x, par = (int)(seq & 1ull);
    if (t < C.world) {
        // the scatter kernel ahead of this one on the stream has completed: its stores are out
        __hip_atomic_store(&C.slab[t]->xflag[par][C.rank], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(&C.slab[C.rank]->xflag[par][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) { *C.error = 1; break; }
        }
    }
    __threadfence_system();
}
#else
;
#endif

// velo_evaluate behind a peer communicator: fixed-order sum of my partial rows, all-reduce, out[28]
__global__ void __launch_bounds__(256)
peer_reduce_kernel(const double* __restrict__ partials, int n_blocks, PeerComm C, double* __restrict__ out)
#if VELO_DEF_LMA
{
    __shared__ double E[32];
    if (threadIdx.x < kNumAcc) {
        double v = 0.0;
        for (int b = 0; b < n_blocks; b++) v += partials[(size_t)b * kNumAcc + threadIdx.x];
        E[threadIdx.x] = v;
    }
    __syncthreads();
    peer_allreduce28(C, E);
    if (threadIdx.x < kNumAcc) out[threadIdx.x] = E[threadIdx.x];
}
#else
;
#endif

#endif  // VELO_UNIT_LM_ONLY
// sums the per-workgroup partials in a fixed order into out[28] (used before the RCCL all-reduce and by velo_evaluate)
__global__ void reduce_partials_kernel(const LMState* __restrict__ state, const double* __restrict__ partials, int n_blocks, double* __restrict__ out)
#if VELO_DEF_LMA
{
    if (state && state->done) return;
    if (threadIdx.x < kNumAcc) {
        double v = 0.0;
        for (int b = 0; b < n_blocks; b++) v += partials[(size_t)b * kNumAcc + threadIdx.x];
        out[threadIdx.x] = v;
    }
}
#else
;
#endif

// ---- one trust-region LM state transition --------------------------------------------------------------------------
// 6x6 SPD solve by Cholesky; one reciprocal per pivot, everything else multiplies (the serial critical path of the
// LM step is divide/sqrt latency)
__device__ inline bool chol_solve6(const double A[36], const double b[6], double y[6]) {
    double L[36], inv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k];
        if (!(d > 0.0) || !isfinite(d)) return false;
        const double ljj = sqrt(d);
        L[j * 6 + j] = ljj;
        inv[j] = 1.0 / ljj;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = s * inv[j];
        }
    }
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[i * 6 + k] * z[k];
        z[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double s = z[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * y[k];
        y[i] = s * inv[i];
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; i++) ok = ok && isfinite(y[i]);
    return ok;
}

// Computes trust-region steps until one is valid (invalid ones cost an iteration but no evaluation).
__device__ inline void lm_compute_step(const LMParams& Q, LMState* S) {
    for (;;) {
        if (S->iter + 1 > Q.max_num_iterations) { S->done = 1; S->termination = 1; return; }
        if (S->radius < Q.min_radius) { S->done = 1; S->termination = 0; return; }
        S->iter++;
        double Hs[36], gs[6];
        for (int i = 0; i < 6; i++) {
            gs[i] = S->g[i] * S->scale[i];
            for (int j = 0; j < 6; j++) {
                const double h = (i <= j) ? S->H[tri(i, j)] : S->H[tri(j, i)];
                Hs[i * 6 + j] = h * S->scale[i] * S->scale[j];
            }
        }
        if (!S->reuse_diag) for (int j = 0; j < 6; j++) S->diag[j] = fmin(fmax(Hs[j * 6 + j], Q.min_diag), Q.max_diag);
        double A[36];
        for (int i = 0; i < 36; i++) A[i] = Hs[i];
        const double inv_radius = 1.0 / S->radius;
        for (int j = 0; j < 6; j++) A[j * 6 + j] += S->diag[j] * inv_radius;      // D^2 = diag / radius
        double y[6], step[6];
        bool ok = chol_solve6(A, gs, y);
        S->reuse_diag = 1;
        double mc = 0.0;
        if (ok) {
            double gd = 0.0, dHd = 0.0;
            for (int i = 0; i < 6; i++) step[i] = -y[i];
            for (int i = 0; i < 6; i++) { gd += gs[i] * step[i]; for (int j = 0; j < 6; j++) dHd += step[i] * Hs[i * 6 + j] * step[j]; }
            mc = -(gd + 0.5 * dHd);
            if (!(mc > 0.0)) ok = false;
        }
        if (!ok) {
            if (++S->invalid >= Q.max_invalid) { S->done = 1; S->termination = 2; return; }
            S->radius = S->radius / S->decrease; S->decrease *= 2.0; S->reuse_diag = 1;
            continue;
        }
        S->invalid = 0;
        double dn = 0.0;
        for (int i = 0; i < 6; i++) { const double d = step[i] * S->scale[i]; S->xc[i] = S->x[i] + d; dn += d * d; }
        S->step_norm = sqrt(dn);
        S->model_change = mc;
        S->phase = PHASE_CAND;
        return;
    }
}

// Start of a solve: the state's bookkeeping and the eval point of the first sweep (four lanes build it).
__device__ __forceinline__ void lm_begin_body(LMState* S, LMEvalPoint* pt, const double* __restrict__ x_in, const int* __restrict__ n_valid) {
    const int t = threadIdx.x;
    double x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = x_in ? x_in[i] : S->x[i];
    if (t == 0) {
        if (x_in) for (int i = 0; i < 6; i++) S->x[i] = x[i];
        S->n_valid = n_valid ? *n_valid : 0;
        S->phase = PHASE_INIT; S->done = 0; S->termination = 1; S->iter = 0; S->evals = 0; S->invalid = 0; S->reuse_diag = 0;
    }
    if (t < 4) eval_point_column(x, 0, t, pt);
}
__global__ void lm_begin_kernel(LMState* S, LMEvalPoint* pt, const double* __restrict__ x_in, const int* __restrict__ n_valid, PoseRecord* __restrict__ pose_out)
#if VELO_DEF_LMA
{
    lm_begin_body(S, pt, x_in, n_valid);
    if (threadIdx.x == 0 && pose_out) pose_out->ready = 0;              // chain mode: the next round's association waits for this solve
}
#else
;
#endif

// The LM step of one solve, by one 256-thread workgroup: fixed-order sum of the per-workgroup partial rows [n_blocks][28] (or of
// the already reduced [1][28] block behind an all-reduce), ONE trust-region state transition, and the eval point of the next
// sweep -- all into LDS (sL, s_pt); the caller stores what it needs.  Laid out for latency: the partial rows and the 496-byte
// state are requested together at the very start (a thread's share of a 128-row chunk is 14 independent coalesced loads, all
// issued before the first is used), the transition runs on a register copy, four lanes build the eval point.
// first != 0: start of a solve (what lm_begin_kernel does) -- no partial rows yet, the state's bookkeeping is reset and the eval
// point is x_in (or the state's x).  Returns with the workgroup synchronised.
__device__ __forceinline__ void lm_transition_local(const LMParams& Q, LMState* S, const double* E, bool* need_step = nullptr);
__device__ __forceinline__ void lm_transition_wave(const LMParams& Q, LMState* sL, const double* E, int lane);
constexpr int kStepChunk = 128;
static_assert(kStepChunk * kNumAcc == kScratchDoubles && kStepChunk * kNumAcc % 256 == 0 && kStepChunk % 8 == 0, "chunk geometry");
// first = 2 (instantiations with BEGIN_STEP only): the start of a solve AND the transition over the first sweep's rows in one go -- the
// launch that carried the solve's first sweep computed its eval point from x itself (eval_step_batch_body), so no begin launch ran
template <bool COHERENT = false, int CHUNK = kStepChunk, bool STATE_COHERENT = false, bool BEGIN_STEP = false>
__device__ __forceinline__ void lm_advance(const LMParams& Q, const LMState* __restrict__ Sin, const double* __restrict__ partials, int n_blocks, int first,
                                           const double* __restrict__ x_in, const int* __restrict__ n_valid,
                                           double* __restrict__ s_rows, LMState* sL, LMEvalPoint* s_pt, unsigned long long* trace, int trace_eval,
                                           const PeerComm* comm = nullptr, PoseRecord* pose_out = nullptr, SolveLog* log = nullptr, bool writer = true,
                                           bool load_state = true, const int tid_in = -1) {
    __shared__ double part[8][kNumAcc];
    __shared__ double E[kNumAcc];
    // CHUNK rows of partial sums at a time through s_rows (CHUNK x 28 doubles).  Thread (k, p) adds the rows b = p (mod 8) of its
    // accumulator k in increasing order whatever the chunk size (a multiple of 64), so every CHUNK gives the same bits.
    static_assert(CHUNK % 64 == 0 && CHUNK <= kStepChunk, "chunk geometry");
    constexpr int kPerThread = CHUNK * kNumAcc / 256;
    const int t = tid_in < 0 ? (int)threadIdx.x : tid_in;
    // STATE_COHERENT (the one-launch solve): the state was written by ANOTHER workgroup of this launch -- agent-scope loads
    // load_state = false (the all-gather solve): the state has lived in this workgroup's LDS since the previous iteration of the same launch
    if (load_state && t < (int)(sizeof(LMState) / 8))
        reinterpret_cast<unsigned long long*>(sL)[t] = STATE_COHERENT
            ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(Sin) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
            : reinterpret_cast<const unsigned long long*>(Sin)[t];
    double xin[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int nv = 0;
    if (first && t == 0) {
        if (x_in) {
#pragma unroll
            for (int i = 0; i < 6; i++) xin[i] = x_in[i];
        }
        nv = n_valid ? *n_valid : 0;
    }
    const int k_acc = t % kNumAcc, p_acc = t / kNumAcc;
    double v_acc = 0.0;
    const bool step = BEGIN_STEP ? first != 1 : !first;
    if (step && CHUNK < kStepChunk && n_blocks > CHUNK && n_blocks <= 2 * CHUNK) {
        // two chunks (the lean instantiations: 64-row chunks, 118 rows): BOTH chunks' loads are issued before the first is consumed -- one
        // memory round trip instead of two on the path every other workgroup's next launch waits for.  Same rows, same order of sums.
        const int n0 = CHUNK * kNumAcc, n1 = (n_blocks - CHUNK) * kNumAcc;
        const double* __restrict__ src1 = partials + (size_t)CHUNK * kNumAcc;
        double a[kPerThread], b[kPerThread];
#pragma unroll
        for (int u = 0; u < kPerThread; u++) { const int i = t + 256 * u; a[u] = COHERENT ? load_agent(partials + i) : partials[i]; }
#pragma unroll
        for (int u = 0; u < kPerThread; u++) { const int i = t + 256 * u; b[u] = (i < n1) ? (COHERENT ? load_agent(src1 + i) : src1[i]) : 0.0; }
#pragma unroll
        for (int u = 0; u < kPerThread; u++) s_rows[t + 256 * u] = a[u];
        __syncthreads();
        if (t < 8 * kNumAcc) for (int r = p_acc; r < CHUNK; r += 8) v_acc += s_rows[r * kNumAcc + k_acc];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPerThread; u++) { const int i = t + 256 * u; if (i < n1) s_rows[i] = b[u]; }
        __syncthreads();
        if (t < 8 * kNumAcc) { for (int r = p_acc; r < n_blocks - CHUNK; r += 8) v_acc += s_rows[r * kNumAcc + k_acc]; part[p_acc][k_acc] = v_acc; }
        (void)n0;
    } else if (step) {
        for (int c0 = 0; c0 < n_blocks; c0 += CHUNK) {
            const int nrows = min(CHUNK, n_blocks - c0), total = nrows * kNumAcc;
            const double* __restrict__ src = partials + (size_t)c0 * kNumAcc;
            double a[kPerThread];
#pragma unroll
            for (int u = 0; u < kPerThread; u++) { const int i = t + 256 * u; a[u] = (i < total) ? (COHERENT ? load_agent(src + i) : src[i]) : 0.0; }
            if (c0 > 0) __syncthreads();                        // the previous chunk has been summed
#pragma unroll
            for (int u = 0; u < kPerThread; u++) { const int i = t + 256 * u; if (i < total) s_rows[i] = a[u]; }
            __syncthreads();
            if (t < 8 * kNumAcc) for (int b = p_acc; b < nrows; b += 8) v_acc += s_rows[b * kNumAcc + k_acc];
        }
        if (t < 8 * kNumAcc) part[p_acc][k_acc] = v_acc;
    }
    __syncthreads();
    VELO_LM_TRACE(trace, trace_eval, 5);
    if (first) {
        if (t == 0) {                                        // lm_begin
            if (x_in) for (int i = 0; i < 6; i++) sL->x[i] = xin[i];
            sL->n_valid = nv;
            sL->phase = PHASE_INIT; sL->done = 0; sL->termination = 1; sL->iter = 0; sL->evals = 0; sL->invalid = 0; sL->reuse_diag = 0;
            if (pose_out && writer) pose_out->ready = 0;     // chain mode: the next round's association must wait for THIS solve
        }
    }
    if (BEGIN_STEP && first == 2) __syncthreads();           // the reset state is what the transition starts from
    if (step && !sL->done) {                                 // uniform: a step behind a finished solve changes nothing
        if (t < kNumAcc) { double v = 0.0; for (int p = 0; p < 8; p++) v += part[p][t]; E[t] = v; }
        __syncthreads();
        if (comm) peer_allreduce28(*comm, E);                // query-sharded: every rank continues with the same 28 sums
        VELO_LM_TRACE(trace, trace_eval, 6);
#ifndef VELO_X1
        if (t < 64) lm_transition_wave(Q, sL, E, t);
#endif
        if (t == 0 && sL->done && writer) {                  // the solve has just finished: what the host and the next round need
            if (log) {
                for (int i = 0; i < 6; i++) log->x[i] = sL->x[i];
                log->initial_cost = sL->initial_cost; log->final_cost = sL->cost;
                log->termination = sL->termination; log->iter = sL->iter; log->evals = sL->evals; log->n_valid = sL->n_valid;
            }
            if (pose_out) {
                double xf[6];
                for (int i = 0; i < 6; i++) xf[i] = sL->x[i];
                PoseScalars S;
#ifndef VELO_X2
                pose_scalars_compute(xf, &S);
#endif
                pose_out->P = S;
                __threadfence();
                pose_out->ready = 1;
            }
        }
    }
    __syncthreads();
    VELO_LM_TRACE(trace, trace_eval, 7);
    if (t < 4) {
        double x[6];
        const bool cand = sL->phase == PHASE_CAND;
#pragma unroll
        for (int k = 0; k < 6; k++) x[k] = cand ? sL->xc[k] : sL->x[k];
#ifndef VELO_X3
        eval_point_column(x, sL->done, t, s_pt);
#endif
    }
    __syncthreads();
    VELO_LM_TRACE(trace, trace_eval, 8);
}

// two launches per iteration (sweep kernels, then this): used behind an all-reduce, with visual blocks, and by the lock-step batch driver
__device__ __forceinline__ void lm_transition(const LMParams& Q, LMState* Sg, LMEvalPoint* pt, const double* __restrict__ partials, int n_blocks,
                                              unsigned long long* trace = nullptr, int trace_eval = 0, const PeerComm* comm = nullptr,
                                              PoseRecord* pose_out = nullptr, SolveLog* log = nullptr) {
    __shared__ LMState sL;
    __shared__ LMEvalPoint s_pt;
    __shared__ double s_rows[kScratchDoubles];
    const int t = threadIdx.x;
    VELO_LM_TRACE(trace, trace_eval, 4);
    lm_advance(Q, Sg, partials, n_blocks, 0, nullptr, nullptr, s_rows, &sL, &s_pt, trace, trace_eval, comm, pose_out, log);
    if (t < (int)(sizeof(LMState) / 8)) reinterpret_cast<unsigned long long*>(Sg)[t] = reinterpret_cast<const unsigned long long*>(&sL)[t];
    else if (t >= 64 && t < 64 + (int)(sizeof(LMEvalPoint) / 8)) reinterpret_cast<unsigned long long*>(pt)[t - 64] = reinterpret_cast<const unsigned long long*>(&s_pt)[t - 64];
    VELO_LM_TRACE(trace, trace_eval, 9);
}

// ONE launch per LM iteration (single GPU, point-to-plane rows only): every sweep workgroup first consumes the PREVIOUS sweep's
// partial rows itself -- the same lm_advance, redundantly in all of them: identical inputs, identical arithmetic, identical
// state and eval point in every workgroup's LDS, no grid barrier and no hand-off between workgroups -- and then sweeps its rows
// at the new point.  State and partial rows are double-buffered (launch k reads buffer k & 1 and writes the other one; workgroup 0
// stores the state), because a fast workgroup must not overwrite what a slow one has yet to read.  Per iteration this is one
// kernel boundary and one cold-load latency (partial rows, state and this workgroup's correspondences are all requested at the
// start) instead of two each; launch 0 also does what lm_begin_kernel did.  Bit-identical to the two-launch path.
// VIS: the launch also carries the visual blocks -- workgroups [nbx, nbx + nb_vis) take them (same transition, then the visual sweep of
// eval_visual_kernel at the new point; their partial rows follow the point-to-plane ones, as in the two-launch path).  A separate
// instantiation: the visual sweep needs 256 VGPRs + AGPRs, the LiDAR-only kernel stays as it is.
template <bool VIS>
__device__ __forceinline__ void lm_iter_body(const EvalArgs& A, const LMParams& Q, const LMState* __restrict__ Sin, LMState* __restrict__ Sout,
                                             const double* __restrict__ pin, int n_in, double* __restrict__ pout, int first,
                                             const double* __restrict__ x_in, const int* __restrict__ n_valid, const int bx, const int nbx,
                                             PoseRecord* pose_out = nullptr, SolveLog* log = nullptr, const int nb_vis = 0) {
    __shared__ LMState sL;
    __shared__ LMEvalPoint s_pt;
    __shared__ double s_scratch[kScratchDoubles];
    const bool icp_wg = !VIS || bx < nbx;                              // wave-uniform: this workgroup sweeps point-to-plane rows
    const RowPrefetch f = prefetch_rows(A, icp_wg ? bx : 0, nbx);
    VELO_LM_TRACE(A.trace, A.trace_eval, 0);
    lm_advance(Q, Sin, pin, n_in, first, x_in, n_valid, s_scratch, &sL, &s_pt, A.trace, A.trace_eval, nullptr, pose_out, log, bx == 0);
    const int t = threadIdx.x;
    if (bx == 0 && t < (int)(sizeof(LMState) / 8)) reinterpret_cast<unsigned long long*>(Sout)[t] = reinterpret_cast<const unsigned long long*>(&sL)[t];
    if (sL.done) return;
    double acc[kNumAcc];
    if (icp_wg) sweep_rows<false>(A, f, s_pt, bx, nbx, acc);
    else if (VIS) visual_sweep_acc(A, s_pt, bx - nbx, nb_vis, acc);
    VELO_LM_TRACE(A.trace, A.trace_eval, 2);
    block_reduce_store(acc, pout + (size_t)bx * kNumAcc, s_scratch);
    VELO_LM_TRACE(A.trace, A.trace_eval, 3);
}
__global__ void __launch_bounds__(kEvalThreads)
lm_iter_kernel(EvalArgs A, LMParams Q, const LMState* __restrict__ Sin, LMState* __restrict__ Sout, const double* __restrict__ pin, int n_in,
               double* __restrict__ pout, int first, const double* __restrict__ x_in, const int* __restrict__ n_valid,
               PoseRecord* pose_out, SolveLog* log)
#if VELO_DEF_LMA
{
    lm_iter_body<false>(A, Q, Sin, Sout, pin, n_in, pout, first, x_in, n_valid, blockIdx.x, gridDim.x, pose_out, log);
}
#else
;
#endif
// grid = nb_icp + nb_vis workgroups
__global__ void __launch_bounds__(kEvalThreads)
lm_iter_vis_kernel(EvalArgs A, LMParams Q, const LMState* __restrict__ Sin, LMState* __restrict__ Sout, const double* __restrict__ pin, int n_in,
                   double* __restrict__ pout, int first, const double* __restrict__ x_in, const int* __restrict__ n_valid,
                   PoseRecord* pose_out, SolveLog* log, int nb_icp, int nb_vis)
#if VELO_DEF_LMA
{
    lm_iter_body<true>(A, Q, Sin, Sout, pin, n_in, pout, first, x_in, n_valid, blockIdx.x, nb_icp, pose_out, log, nb_vis);
}
#else
;
#endif
// the same for the contexts of a lock-step group: blockIdx.y = context, k = index of the launch within the solve (its parity
// selects the halves of every context's state / partial-row double buffer; `half` = doubles per half)
__global__ void __launch_bounds__(kEvalThreads)
lm_iter_batch_kernel(LMParams Q, const LMBatchItem* __restrict__ items, int k, size_t half)
#if VELO_DEF_LMA
{
    const LMBatchItem& it = items[blockIdx.y];
    if ((int)blockIdx.x >= it.nb_icp) return;
    lm_iter_body<false>(it.A, Q, it.S + (k & 1), it.S + ((k + 1) & 1), it.A.partials + (size_t)(k & 1) * half, it.nb_icp,
                 it.A.partials + (size_t)((k + 1) & 1) * half, k == 0 ? 1 : 0, it.xd, it.n_valid, blockIdx.x, it.nb_icp, it.pose_out, it.log);
}
#else
;
#endif

// need_step != null: the trust-region step is left to the caller (lm_compute_step_wave), *need_step says whether one is due
__device__ __forceinline__ void lm_transition_local(const LMParams& Q, LMState* S, const double* E, bool* need_step) {
    S->evals++;
    const double ecost = E[27];
    if (S->phase == PHASE_INIT) {
        for (int k = 0; k < 21; k++) S->H[k] = E[k];
        for (int k = 0; k < 6; k++) S->g[k] = E[21 + k];
        S->cost = ecost; S->initial_cost = ecost;
        double xn = 0.0, gm = 0.0;
        for (int i = 0; i < 6; i++) { xn += S->x[i] * S->x[i]; gm = fmax(gm, fabs(S->g[i])); }
        S->x_norm = sqrt(xn);
        if (gm <= Q.gradient_tolerance) { S->done = 1; S->termination = 0; return; }
        for (int j = 0; j < 6; j++) S->scale[j] = 1.0 / (1.0 + sqrt(S->H[tri(j, j)]));
        S->radius = Q.initial_radius; S->decrease = 2.0; S->reuse_diag = 0; S->invalid = 0;
        if (need_step) *need_step = true; else lm_compute_step(Q, S);
        return;
    }
    // PHASE_CAND: E is the evaluation at xc
    if (S->step_norm <= Q.parameter_tolerance * (S->x_norm + Q.parameter_tolerance)) { S->done = 1; S->termination = 0; return; }
    const double cost_change = S->cost - ecost;
    if (fabs(cost_change) <= Q.function_tolerance * S->cost) { S->done = 1; S->termination = 0; return; }
    const double q = cost_change / S->model_change;
    if (q > Q.min_relative_decrease) {
        double xn = 0.0, gm = 0.0;
        for (int i = 0; i < 6; i++) { S->x[i] = S->xc[i]; xn += S->x[i] * S->x[i]; }
        for (int k = 0; k < 21; k++) S->H[k] = E[k];
        for (int k = 0; k < 6; k++) { S->g[k] = E[21 + k]; gm = fmax(gm, fabs(S->g[k])); }
        S->cost = ecost; S->x_norm = sqrt(xn);
        if (gm <= Q.gradient_tolerance) { S->done = 1; S->termination = 0; return; }
        const double tt = 2.0 * q - 1.0;
        S->radius = fmin(Q.max_radius, S->radius / fmax(1.0 / 3.0, 1.0 - tt * tt * tt));
        S->decrease = 2.0; S->reuse_diag = 0;
    } else {
        S->radius = S->radius / S->decrease; S->decrease *= 2.0; S->reuse_diag = 1;
    }
    if (need_step) *need_step = true; else lm_compute_step(Q, S);
}

// lm_compute_step by the 64 lanes of one wave.  The serial version is a chain of ~600 dependent double-precision operations on
// one lane (3.8 us of a 14.5 us LM iteration); here the independent ones run side by side -- the 36 scaled entries, the rows of a
// Cholesky column, the forward substitution's updates, the 36 products of d^T H d -- while every SUM keeps the serial order
// (columns left to right, rows top to bottom), so each result is the same sequence of IEEE operations: bit-identical to
// lm_compute_step, which the other paths (two-launch step, single-launch small solve) still run; tests compare them.
// Lane r < 6 owns row r of the matrix; values every lane needs are computed redundantly (same cost as on one lane) or read
// with v_readlane from the lane that owns them.
__device__ __forceinline__ double lane_get(double v, int src_lane) {       // src_lane: compile-time constant after unrolling
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ void lm_compute_step_wave(const LMParams& Q, LMState* S, int lane) {
    const int r = lane < 6 ? lane : 5;                                 // row of this lane (lanes >= 6 shadow row 5; their values are never read)
    double sc[6], gs[6], hrow[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { sc[i] = S->scale[i]; gs[i] = S->g[i] * sc[i]; }
    const double scr = S->scale[r];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const double h = S->H[r <= c ? tri(r, c) : tri(c, r)];
        hrow[c] = h * scr * sc[c];                                     // Hs[r][c] = h * scale[r] * scale[c]
    }
    const int ei = lane < 36 ? lane / 6 : 5, ej = lane < 36 ? lane % 6 : 5;   // element view for d^T Hs d: lane l < 36 owns Hs[l / 6][l % 6]
    const double hs_e = S->H[ei <= ej ? tri(ei, ej) : tri(ej, ei)] * S->scale[ei] * S->scale[ej];
    int iter = S->iter, invalid = S->invalid, reuse = S->reuse_diag;
    double radius = S->radius, decrease = S->decrease;
    double diag_r = S->diag[r];
    for (;;) {
        if (iter + 1 > Q.max_num_iterations) { if (lane == 0) { S->done = 1; S->termination = 1; } break; }
        if (radius < Q.min_radius) { if (lane == 0) { S->done = 1; S->termination = 0; } break; }
        iter++;
        if (!reuse) {
            double hdd = hrow[0];
#pragma unroll
            for (int c = 1; c < 6; c++) hdd = (r == c) ? hrow[c] : hdd;
            diag_r = fmin(fmax(hdd, Q.min_diag), Q.max_diag);
        }
        const double inv_radius = 1.0 / radius;
        const double dterm = diag_r * inv_radius;                      // D^2 = diag / radius
        double L[6], inv[6];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double sj = (r == j) ? hrow[j] + dterm : hrow[j];          // A[r][j]
#pragma unroll
            for (int k = 0; k < j; k++) sj -= L[k] * lane_get(L[k], j);        // L[r][k] * L[j][k]; on lane j itself: L[j][k]^2
            const double d = lane_get(sj, j);
            if (!(d > 0.0) || !isfinite(d)) { ok = false; break; }
            const double ljj = sqrt(d);
            inv[j] = 1.0 / ljj;
            L[j] = (r == j) ? ljj : sj * inv[j];
        }
        double y[6];
        if (ok) {
            double acc = gs[0];
#pragma unroll
            for (int c = 1; c < 6; c++) acc = (r == c) ? gs[c] : acc;  // b[r]
            double z[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                z[k] = lane_get(acc, k) * inv[k];                      // row k has received all its k updates
                acc -= L[k] * z[k];                                    // rows below: s -= L[i][k] * z[k], k ascending as in the serial loop
            }
#pragma unroll
            for (int i = 5; i >= 0; i--) {
                double sv = z[i];
#pragma unroll
                for (int k = i + 1; k < 6; k++) sv -= lane_get(L[i], k) * y[k];       // L[k][i] lives on lane k
                y[i] = sv * inv[i];
            }
#pragma unroll
            for (int i = 0; i < 6; i++) ok = ok && isfinite(y[i]);
        }
        reuse = 1;
        double mc = 0.0, step[6];
        if (ok) {
            double gd = 0.0, dHd = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) step[i] = -y[i];
            double si = step[0], sj = step[0];
#pragma unroll
            for (int c = 1; c < 6; c++) { si = (ei == c) ? step[c] : si; sj = (ej == c) ? step[c] : sj; }
            const double prod = si * hs_e * sj;                        // step[i] * Hs[i][j] * step[j]
#pragma unroll
            for (int i = 0; i < 6; i++) gd += gs[i] * step[i];
#pragma unroll
            for (int l = 0; l < 36; l++) dHd += lane_get(prod, l);     // i-major, j inner: the serial order
            mc = -(gd + 0.5 * dHd);
            if (!(mc > 0.0)) ok = false;
        }
        if (!ok) {
            if (++invalid >= Q.max_invalid) { if (lane == 0) { S->done = 1; S->termination = 2; } break; }
            radius = radius / decrease; decrease *= 2.0; reuse = 1;
            continue;
        }
        invalid = 0;
        if (lane == 0) {
            double dn = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++) { const double d = step[i] * sc[i]; S->xc[i] = S->x[i] + d; dn += d * d; }
            S->step_norm = sqrt(dn);
            S->model_change = mc;
            S->phase = PHASE_CAND;
        }
        break;
    }
    if (lane < 6) S->diag[lane] = diag_r;
    if (lane == 0) { S->iter = iter; S->invalid = invalid; S->reuse_diag = reuse; S->radius = radius; S->decrease = decrease; }
}
// the transition by the first wave of the step's workgroup: bookkeeping on lane 0, the trust-region step by all 64 lanes
__device__ __forceinline__ void lm_transition_wave(const LMParams& Q, LMState* sL, const double* E, int lane) {
    int need = 0;
    if (lane == 0) {
        LMState L = *sL;
        bool ns = false;
        lm_transition_local(Q, &L, E, &ns);
        *sL = L;
        need = ns ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    need = __builtin_amdgcn_readfirstlane(need);
    if (need) lm_compute_step_wave(Q, sL, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(256)
lm_step_kernel(LMParams Q, LMState* S, LMEvalPoint* pt, const double* __restrict__ partials, int n_blocks, unsigned long long* trace, int trace_eval,
               PoseRecord* __restrict__ pose_out, SolveLog* __restrict__ log)
#if VELO_DEF_LMA
{
    lm_transition(Q, S, pt, partials, n_blocks, trace, trace_eval, nullptr, pose_out, log);
}
#else
;
#endif
// the LM step of a query-sharded solve: my partial rows, the peer all-reduce, the transition -- one launch
__global__ void __launch_bounds__(256)
lm_step_peer_kernel(LMParams Q, LMState* S, LMEvalPoint* pt, const double* __restrict__ partials, int n_blocks, PeerComm C,
                    PoseRecord* __restrict__ pose_out, SolveLog* __restrict__ log)
#if VELO_DEF_LMA
{
    lm_transition(Q, S, pt, partials, n_blocks, nullptr, 0, &C, pose_out, log);
}
#else
;
#endif
__global__ void lm_begin_batch_kernel(const LMBatchItem* __restrict__ items)
#if VELO_DEF_LMA
{
    const LMBatchItem& it = items[blockIdx.x];
    lm_begin_body(it.S, const_cast<LMEvalPoint*>(it.A.pt), it.xd, it.n_valid);
    if (threadIdx.x == 0 && it.pose_out) it.pose_out->ready = 0;       // chain mode: the next round's association waits for this solve
}
#else
;
#endif
__global__ void __launch_bounds__(256)
lm_step_batch_kernel(LMParams Q, const LMBatchItem* __restrict__ items)
#if VELO_DEF_LMA
{
    const LMBatchItem& it = items[blockIdx.x];
    lm_transition(Q, it.S, const_cast<LMEvalPoint*>(it.A.pt), it.A.partials, it.n_rows, nullptr, 0, nullptr, it.pose_out, it.log);
}
#else
;
#endif
// Sweep AND step of a lock-step group in ONE launch per LM iteration (point-to-plane rows only): every workgroup sweeps its rows
// and publishes its partial row with agent-scope stores; the workgroup of a context that draws the last ticket then does what
// lm_step_batch_kernel does -- the same fixed-order sum over the same rows, the same transition, so the results are bit-identical
// to the two-launch path -- and leaves state and eval point for the next launch.  One kernel boundary per iteration instead of two
// and ONE transition per context (the one-launch iteration of the single-pair path repeats it in every workgroup, which costs a
// group of 2-4 contexts more residency than the boundary it saves: measured, see the batch driver).
// Ordering: the 28 stores of a row are agent-scope write-through stores; their thread waits for them (s_waitcnt vmcnt(0)) ahead of
// the workgroup barrier behind which thread 0 draws the ticket, so whoever sees ticket n - 1 finds every row where its agent-scope
// loads look.  tickets[context] is 0 at every launch boundary (the last workgroup resets it).
// first != 0: this launch carries the FIRST sweep of a solve: every workgroup builds the start's eval point from x itself (what
// lm_begin_batch_kernel used to leave in memory one launch earlier), and the stepping workgroup starts the solve before it steps.
// VIS: the contexts' visual blocks ride in the same launch -- workgroups nb_icp .. nb_icp + nb_vis - 1 of a context run the visual sweep
// (visual_sweep_acc, the arithmetic of eval_visual_kernel) into the partial rows behind the point-to-plane ones and draw tickets like
// the others, instead of a launch of their own ahead of every iteration (21.7 us each, 48 per call).
template <bool M_LDS, int PRE, int CHUNK, bool VIS = false>
__device__ __forceinline__ void eval_step_batch_body(const LMParams& Q, const LMBatchItem& it, int* __restrict__ tickets, const int first) {
    const int bx = blockIdx.x, nbx = it.nb_icp, nb_all = VIS ? it.nb_icp + it.nb_vis : it.nb_icp;
    if (bx >= nb_all) return;
#ifdef VELO_LM_SETPRIO
    __builtin_amdgcn_s_setprio(VELO_LM_SETPRIO);                      // A/B: the LM chain's waves ahead of other groups' association waves in a SIMD's issue arbitration
#endif
    const EvalArgs& A = it.A;
    const RowPrefetchT<PRE> f = prefetch_rows<PRE>(A, (VIS && bx >= nbx) ? 0 : bx, nbx);   // (a visual workgroup: block 0's rows, unused)
    __shared__ LMEvalPoint s_pt;
    __shared__ LMState sL;
    __shared__ double s_scratch[CHUNK * kNumAcc];
    __shared__ int s_last;
    static_assert(CHUNK * kNumAcc >= kScratchDoubles / 2, "the sweep's reduction needs 64 columns x 28");
    if (first) {
        if (threadIdx.x < 4) {
            double x[6];
#pragma unroll
            for (int k = 0; k < 6; k++) x[k] = it.xd ? it.xd[k] : it.S->x[k];
            eval_point_column(x, 0, threadIdx.x, &s_pt);
        }
        __syncthreads();
    } else
    if (!eval_point_load(A, &s_pt)) return;                           // a launch behind the end of the solve: nothing to do, the ticket stays 0
    double acc[kNumAcc];
    if (VIS && bx >= nbx) {
        if (M_LDS) visual_sweep_one(A, s_pt, bx - nbx, it.nb_vis, acc);      // (the host sends a group here only when every thread holds at most one slot)
        else visual_sweep_acc(A, s_pt, bx - nbx, it.nb_vis, acc);
        block_reduce_store<true>(acc, A.partials + (size_t)(A.vis_row0 + bx - nbx) * kNumAcc, s_scratch);
    } else {
        sweep_rows<M_LDS, PRE>(A, f, s_pt, bx, nbx, acc);
        block_reduce_store<true>(acc, A.partials + (size_t)bx * kNumAcc, s_scratch);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this thread's row entries have been written through
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(tickets + blockIdx.y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb_all - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(tickets + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lm_advance<true, CHUNK, false, true>(Q, it.S, A.partials, it.n_rows, first ? 2 : 0, it.xd, it.n_valid, s_scratch, &sL, &s_pt, nullptr, 0, nullptr, it.pose_out, it.log);
    const int t = threadIdx.x;
    LMEvalPoint* pt = const_cast<LMEvalPoint*>(A.pt);
    if (t < (int)(sizeof(LMState) / 8)) reinterpret_cast<unsigned long long*>(it.S)[t] = reinterpret_cast<const unsigned long long*>(&sL)[t];
    else if (t >= 64 && t < 64 + (int)(sizeof(LMEvalPoint) / 8)) reinterpret_cast<unsigned long long*>(pt)[t - 64] = reinterpret_cast<const unsigned long long*>(&s_pt)[t - 64];
}
// alone on the chip: matrices in registers, four prefetched rows (240 VGPRs, 33 KB of LDS: two waves per SIMD)
__global__ void __launch_bounds__(kEvalThreads)
eval_step_batch_kernel(LMParams Q, const LMBatchItem* __restrict__ items, int* __restrict__ tickets, int first)
#if VELO_DEF_LMB
{
    eval_step_batch_body<false, kPre, kStepChunk>(Q, items[blockIdx.y], tickets, first);
}
#else
;
#endif
// The LEAN instantiation, for launches that share the chip with other lock-step groups' association kernels: matrices read from LDS,
// VELO_LEAN_PRE prefetched rows, the step's partial rows 64 at a time -- few enough registers and LDS (<= 152 VGPRs, < 25 KB) that a
// workgroup fits on a CU beside FIVE association workgroups (5 x 72 VGPRs of 512 per SIMD, 5 x 27 KB of 160 KB with the pad the batch
// driver gives them), so an LM launch never waits for association workgroups to drain.  Same arithmetic, same order: bit-identical.
#ifndef VELO_LEAN_PRE
#define VELO_LEAN_PRE 1
#endif
__global__ void __launch_bounds__(kEvalThreads) __attribute__((amdgpu_num_vgpr(152)))
eval_step_batch_lean_kernel(LMParams Q, const LMBatchItem* __restrict__ items, int* __restrict__ tickets, int first)
#if VELO_DEF_LMB
{
    eval_step_batch_body<true, VELO_LEAN_PRE, 64>(Q, items[blockIdx.y], tickets, first);
}
#else
;
#endif
__global__ void __launch_bounds__(kEvalThreads) __attribute__((amdgpu_num_vgpr(152)))
eval_step_batch_lean_vis_kernel(LMParams Q, const LMBatchItem* __restrict__ items, int* __restrict__ tickets, int first)
#if VELO_DEF_LMB
{
    eval_step_batch_body<true, VELO_LEAN_PRE, 64, true>(Q, items[blockIdx.y], tickets, first);
}
#else
;
#endif
__global__ void __launch_bounds__(kEvalThreads)
eval_step_batch_vis_kernel(LMParams Q, const LMBatchItem* __restrict__ items, int* __restrict__ tickets, int first)
#if VELO_DEF_LMB
{
    eval_step_batch_body<false, kPre, kStepChunk, true>(Q, items[blockIdx.y], tickets, first);
}
#else
;
#endif
// The same two with the group's items BY VALUE in the kernel arguments (groups of up to four contexts): no copy of the items into
// device memory ahead of every round of a chained call -- six copy operations, and the queue hand-overs around them, per call.
struct LMBatchPackV { LMBatchItem item[4]; };
__global__ void __launch_bounds__(kEvalThreads)
eval_step_batch_v_kernel(LMParams Q, LMBatchPackV P, int* __restrict__ tickets, int first)
#if VELO_DEF_LMB
{
    eval_step_batch_body<false, kPre, kStepChunk>(Q, P.item[blockIdx.y], tickets, first);
}
#else
;
#endif
__global__ void __launch_bounds__(kEvalThreads) __attribute__((amdgpu_num_vgpr(152)))
eval_step_batch_lean_v_kernel(LMParams Q, LMBatchPackV P, int* __restrict__ tickets, int first)
#if VELO_DEF_LMB
{
    eval_step_batch_body<true, VELO_LEAN_PRE, 64>(Q, P.item[blockIdx.y], tickets, first);
}
#else
;
#endif
__global__ void __launch_bounds__(kEvalThreads) __attribute__((amdgpu_num_vgpr(152)))
eval_step_batch_lean_vis_v_kernel(LMParams Q, LMBatchPackV P, int* __restrict__ tickets, int first)
#if VELO_DEF_LMB
{
    eval_step_batch_body<true, VELO_LEAN_PRE, 64, true>(Q, P.item[blockIdx.y], tickets, first);
}
#else
;
#endif
// The one-launch iteration of the single-pair path (every workgroup runs the transition itself, then sweeps: no last-workgroup hand-over
// inside the launch) for the contexts of a lock-step group, in the LEAN shape of eval_step_batch_lean_kernel (matrices from LDS, one
// prefetched row, 64-row chunks), items by value.  parity selects the halves of every context's state / partial-row double buffer (a
// running launch count across the call's rounds).  Same virtual blocks, same order of sums: bit-identical to every other path.
// VELO_LM_ITER=1 (diagnostics build).  Measured (round 3, 8 different pairs in flight): 24.4 us per launch against 26.6 us for the
// fused sweep + step, but a solve needs one launch more and every workgroup re-reads all 118 partial rows: 3,204 vs 3,418 pairs/s.
template <bool M_LDS, int PRE, int CHUNK>
__device__ __forceinline__ void lm_iter_lean_body(const EvalArgs& A, const LMParams& Q, const LMState* __restrict__ Sin, LMState* __restrict__ Sout,
                                                  const double* __restrict__ pin, int n_in, double* __restrict__ pout, int first,
                                                  const double* __restrict__ x_in, const int* __restrict__ n_valid, const int bx, const int nbx,
                                                  PoseRecord* pose_out, SolveLog* log) {
    __shared__ LMState sL;
    __shared__ LMEvalPoint s_pt;
    __shared__ double s_scratch[CHUNK * kNumAcc];
    const RowPrefetchT<PRE> f = prefetch_rows<PRE>(A, bx, nbx);
    lm_advance<false, CHUNK>(Q, Sin, pin, n_in, first, x_in, n_valid, s_scratch, &sL, &s_pt, nullptr, 0, nullptr, pose_out, log, bx == 0);
    const int t = threadIdx.x;
    if (bx == 0 && t < (int)(sizeof(LMState) / 8)) reinterpret_cast<unsigned long long*>(Sout)[t] = reinterpret_cast<const unsigned long long*>(&sL)[t];
    if (sL.done) return;
    double acc[kNumAcc];
    sweep_rows<M_LDS, PRE>(A, f, s_pt, bx, nbx, acc);
    block_reduce_store(acc, pout + (size_t)bx * kNumAcc, s_scratch);
}
__global__ void __launch_bounds__(kEvalThreads) __attribute__((amdgpu_num_vgpr(152)))
lm_iter_batch_lean_kernel(LMParams Q, LMBatchPackV P, int parity, int first, size_t half)
#if VELO_DEF_LMB
{
    const LMBatchItem& it = P.item[blockIdx.y];
    if ((int)blockIdx.x >= it.nb_icp) return;
    lm_iter_lean_body<true, VELO_LEAN_PRE, 64>(it.A, Q, it.S + parity, it.S + (parity ^ 1), it.A.partials + (size_t)parity * half, it.nb_icp,
                                               it.A.partials + (size_t)(parity ^ 1) * half, first, it.xd, it.n_valid, blockIdx.x, it.nb_icp, it.pose_out, it.log);
}
#else
;
#endif
// (the all-gather solve, lm_solve_ag_batch_kernel, lives in velo_lm_ag_kernels.h / velo_lm_ag.hip: a translation unit of its own)
__device__ __forceinline__ int ctl_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ctl_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifndef VELO_TEST_ATTR
#define VELO_TEST_ATTR __attribute__((amdgpu_num_vgpr(152)))
#endif
struct AgCtl { int epoch, abort, pad[30]; int flag[2][kMaxEvalBlocks]; };

// ---- a whole solve of a lock-step group in ONE launch ---------------------------------------------------------------------------------
// A solve used to be one launch per LM iteration: ~48 launches per call and group, each paying the queue's hand-over (4-10 us between
// two kernels of a queue when four queues are busy), the dispatch, the cold loads of eval point and rows, and -- arriving while other
// groups' association kernels fill the chip -- the wait for CU slots.  Here the workgroups of a context stay for the whole solve.
// Work is handed out by TICKET: ticket t = (iteration t / nb, virtual block t % nb); a workgroup takes the next ticket, prefetches
// that block's rows (they do not depend on the pose), waits until the eval point of its iteration is published, sweeps, publishes the
// block's partial row and counts itself in; the workgroup that completes an iteration does the transition (lm_advance, same order of
// sums) and publishes the next eval point -- then everybody takes the next ticket.  A workgroup only ever waits for work whose
// tickets were drawn BEFORE its own, i.e. for workgroups that are running: no deadlock however few workgroups are resident, and no
// prediction of the iteration count -- the launch ends when the solve does (kmax bounds it).  Virtual blocks, per-thread rows and
// the reduction are those of the launch-per-iteration kernels, so every partial row and every sum is bit-identical to them.
// Everything one workgroup hands to another inside the launch goes through agent-scope accesses (partial rows, state, eval point).
// One per context; all zero between launches (the last workgroup out resets it).  Every word on a 128-byte line of its own: the
// generation word is polled by every waiting workgroup, and a ticket or arrival counter on the same line would queue behind the polls.
// gen = eval points published so far, | kSolveDone once the solve has ended.
struct SolveCtl { int ticket, pad0[31]; int arrived, pad1[31]; int gen, pad2[31]; int exited, pad3[31]; };
constexpr int kSolveDone = 1 << 30;
// one ticket's work: block bx of iteration k.  -> 0 = swept, 1 = swept and this workgroup completed the iteration, -1 = the solve is over
template <int PRE>
__device__ __forceinline__ int persist_sweep(const EvalArgs& A, SolveCtl* __restrict__ ctl, const int k, const int bx, const int nb,
                                             LMEvalPoint* s_pt, double* s_scratch, int* s_flag) {
    const int t = threadIdx.x;
    const LMEvalPoint* pt = A.pt;
    const RowPrefetchT<PRE> f = prefetch_rows<PRE>(A, bx, nb);
    // wait for the eval point of iteration k (generation k + 1), or for the end of the solve
    if (t == 0) {
        int fl = 0;
        for (;;) {
            const int g = ctl_load(&ctl->gen);
            if (g & kSolveDone) { fl = 1; break; }
            if (g > k) break;
            __builtin_amdgcn_s_sleep(8);
        }
        *s_flag = fl;
    }
    __syncthreads();
    const int over = *s_flag;
    __syncthreads();
    if (over) return -1;
    if (t < (int)(sizeof(LMEvalPoint) / 8))
        reinterpret_cast<unsigned long long*>(s_pt)[t] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(pt) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    double acc[kNumAcc];
#ifndef VELO_T2
    sweep_rows<true, PRE>(A, f, *s_pt, bx, nb, acc);
#else
    for (int q = 0; q < kNumAcc; q++) acc[q] = f.p[0].x;
#endif
    block_reduce_store<true>(acc, A.partials + (size_t)bx * kNumAcc, s_scratch);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // this thread's row entries have been written through
    __syncthreads();
    if (t == 0) *s_flag = __hip_atomic_fetch_add(&ctl->arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1 ? 1 : 0;
    __syncthreads();
    const int last = *s_flag;
    __syncthreads();
    return last;
}
// the transition of the one-launch solve as a real function: inlined into the ticket loop it drives the whole kernel over its register
// budget (the sweep would spill); as a call it costs the launch a stack frame -- once per solve, not once per LM iteration
__device__ __attribute__((noinline)) void persist_advance(const LMParams* Q, const LMBatchItem* it, int first, double* s_scratch, LMState* sL, LMEvalPoint* s_pt) {
    lm_advance<true, 64, true>(*Q, it->S, it->A.partials, it->n_rows, first, it->xd, it->n_valid, s_scratch, sL, s_pt, nullptr, 0, nullptr, it->pose_out, it->log);
}
template <int PRE>
__device__ __forceinline__ void lm_solve_persist_body(const LMParams& Q, const LMBatchItem* __restrict__ item, SolveCtl* __restrict__ ctl, const int kmax, const int n_wgs) {
    __shared__ LMEvalPoint s_pt;
    __shared__ LMState sL;
    __shared__ double s_scratch[64 * kNumAcc];
    __shared__ int s_ticket, s_flag;
    const int t = threadIdx.x;
    bool running = true;
    while (running) {
        // The item is re-read (scalar loads) in every round of the loop through a pointer the compiler cannot see through: hoisted out of
        // the loop its ~40 fields stay live across the sweep and the transition, and the kernel spills 100+ registers.
        const LMBatchItem* itp = item;
        asm volatile("" : "+s"(itp));
        const LMBatchItem& it = *itp;
        const EvalArgs& A = it.A;
        const int nb = it.nb_icp;
        LMEvalPoint* pt = const_cast<LMEvalPoint*>(A.pt);
        if (t == 0) s_ticket = __hip_atomic_fetch_add(&ctl->ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int tk = s_ticket;                                          // ticket 0 starts the solve; ticket 1 + k nb + bx = block bx of iteration k
        __syncthreads();
        const int k = (tk - 1) / nb, bx = (tk - 1) - k * nb;
        int advance = 0;                                                  // this workgroup runs lm_advance: 1 = start of the solve, 2 = it completed iteration k
        if (tk == 0) advance = 1;
        else if (k >= kmax) running = false;
        else {
            const int r = persist_sweep<PRE>(A, ctl, k, bx, nb, &s_pt, s_scratch, &s_flag);
            if (r < 0) running = false;
            else if (r > 0) advance = 2;
        }
        if (advance) {                                                    // (workgroup-uniform)
            if (advance == 2 && t == 0) ctl_store(&ctl->arrived, 0);
            // start: what lm_begin_kernel does (state reset, eval point 0); else the transition over this iteration's partial rows
#ifndef VELO_T1
            persist_advance(&Q, itp, advance == 1 ? 1 : 0, s_scratch, &sL, &s_pt);
#endif
            if (t < (int)(sizeof(LMState) / 8)) __hip_atomic_store(reinterpret_cast<unsigned long long*>(it.S) + t, reinterpret_cast<const unsigned long long*>(&sL)[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (t >= 64 && t < 64 + (int)(sizeof(LMEvalPoint) / 8)) __hip_atomic_store(reinterpret_cast<unsigned long long*>(pt) + (t - 64), reinterpret_cast<const unsigned long long*>(&s_pt)[t - 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) ctl_store(&ctl->gen, (advance == 1 ? 1 : k + 2) | (sL.done ? kSolveDone : 0));
            __syncthreads();
        }
    }
    // the last workgroup out leaves the control block as it found it
    __syncthreads();
    if (t == 0 && __hip_atomic_fetch_add(&ctl->exited, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_wgs - 1) {
        ctl_store(&ctl->ticket, 0); ctl_store(&ctl->arrived, 0); ctl_store(&ctl->gen, 0); ctl_store(&ctl->exited, 0);
    }
}
__global__ void __launch_bounds__(kEvalThreads, 3)                    // <= 168 VGPRs: the workgroups stay for a whole solve, beside other groups' association workgroups
lm_solve_persist_batch_kernel(LMParams Q, const LMBatchItem* __restrict__ items, SolveCtl* __restrict__ ctl, int kmax)
#if VELO_DEF_LMB
{
    lm_solve_persist_body<VELO_LEAN_PRE>(Q, items + blockIdx.y, ctl + blockIdx.y, kmax, (int)gridDim.x);
}
#else
;
#endif

// all states of the batch into one contiguous block (one D2H copy per chunk instead of one per context)
__global__ void lm_gather_states_kernel(const LMBatchItem* __restrict__ items, LMState* __restrict__ out, int which)
#if VELO_DEF_LMB
{
    const unsigned* src = reinterpret_cast<const unsigned*>(items[blockIdx.x].S + which);
    unsigned* dst = reinterpret_cast<unsigned*>(out + blockIdx.x);
    for (int k = threadIdx.x; k < (int)(sizeof(LMState) / 4); k += blockDim.x) dst[k] = src[k];
}
#else
;
#endif

// The end of a chained call of a lock-step group in ONE launch instead of a gather launch + three to seven copies: every context's final
// state, the solve logs, the chain's failure flags (and, with visual blocks, the match flags and block counts) written straight into the
// group's page-locked result block -- with four busy queues every queue operation costs ~12 us of stream time whatever it does.
constexpr int kFinishJobs = 8;
struct ChainFinish {
    const LMState* S[kFinishJobs];
    const unsigned char* vflags[kFinishJobs]; const int* vis_counts[kFinishJobs];
    unsigned char* h_vflags[kFinishJobs]; int* h_vis_counts[kFinishJobs];
    int n_vflags[kFinishJobs];
    const SolveLog* logs; const int* fail;
    LMState* h_states; SolveLog* h_logs; int* h_fail;
    int n, n_logs, n_counts;
};
__global__ void __launch_bounds__(256) chain_finish_kernel(ChainFinish F)
#if VELO_DEF_LMB
{
    const int i = blockIdx.x, t = threadIdx.x;
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(F.S[i]);
        unsigned* dst = reinterpret_cast<unsigned*>(F.h_states + i);
        for (int k = t; k < (int)(sizeof(LMState) / 4); k += 256) dst[k] = src[k];
    }
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(F.logs + (size_t)i * F.n_logs);
        unsigned* dst = reinterpret_cast<unsigned*>(F.h_logs + (size_t)i * F.n_logs);
        for (int k = t; k < (int)(sizeof(SolveLog) / 4) * F.n_logs; k += 256) dst[k] = src[k];
    }
    if (t == 0) F.h_fail[i] = F.fail[i];
    if (F.n_vflags[i] > 0) {
        for (int k = t; k < F.n_vflags[i]; k += 256) F.h_vflags[i][k] = F.vflags[i][k];
        for (int k = t; k < F.n_counts; k += 256) F.h_vis_counts[i][k] = F.vis_counts[i][k];
    }
}
#else
;
#endif

// A whole ceres::Solve in ONE single-workgroup launch, for problems whose sweep is at most kSmallRows workgroups anyway (the
// reference's own configuration: icp_skip = 200 -> 640 queries = one workgroup).  There a solve is ~8 LM iterations x two
// launches of a few microseconds of work each plus host round trips: pure launch latency (1.1 ms per frame_to_frame call).
// Here the workgroup walks the virtual blocks of the sweep one after the other -- the SAME eval bodies with the same block
// index and block count, so every partial row is bit-identical to the multi-launch path -- reduces them in the same order,
// thread 0 does the transition, and the loop continues on the device.  State and partial rows live in LDS for the duration.
constexpr int kSmallRows = 4;
template <bool VIS = true>                                             // VIS = false: no visual blocks -- the instantiation without their code (and its registers)
__device__ __forceinline__ void
lm_solve_small_body(const EvalArgs& A, const LMParams& Q, LMState* Sg, const double* __restrict__ x_in, const int* __restrict__ n_valid,
                    int nb_icp, int nb_vis, int max_sweeps, PoseRecord* __restrict__ pose_out, SolveLog* __restrict__ log) {
    __shared__ double rows[kSmallRows][kNumAcc];
    __shared__ double part[8][kNumAcc];
    __shared__ double E[kNumAcc];
    __shared__ double s_x[6];
    __shared__ LMState sL;
    __shared__ int s_done;
    const int t = threadIdx.x;
    if (t == 0) {                                            // lm_begin_kernel
        LMState L = *Sg;
        if (x_in) for (int i = 0; i < 6; i++) L.x[i] = x_in[i];
        L.n_valid = n_valid ? *n_valid : 0;
        L.phase = PHASE_INIT; L.done = 0; L.termination = 1; L.iter = 0; L.evals = 0; L.invalid = 0; L.reuse_diag = 0;
        sL = L;
        for (int i = 0; i < 6; i++) s_x[i] = L.x[i];
        s_done = 0;
        if (pose_out) pose_out->ready = 0;                   // chain mode: the next round's association waits for this solve
    }
    __syncthreads();
    EvalArgs B = A;
    B.x_override = s_x;                                      // the sweeps read the point from LDS, not from the global state
    B.partials = &rows[0][0];
    B.vis_row0 = nb_icp;
    const int n_rows = nb_icp + nb_vis;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        if (s_done) break;
        for (int bx = 0; bx < nb_icp; bx++) eval_icp_body(B, bx, nb_icp);
        if (VIS) for (int bx = 0; bx < nb_vis; bx++) eval_visual_body(B, bx, nb_vis);
        __syncthreads();
        if (t < 8 * kNumAcc) {                               // the reduction of lm_transition, same order
            const int k = t % kNumAcc, p = t / kNumAcc;
            double v = 0.0;
            for (int b = p; b < n_rows; b += 8) v += rows[b][k];
            part[p][k] = v;
        }
        __syncthreads();
        if (t < kNumAcc) { double v = 0.0; for (int p = 0; p < 8; p++) v += part[p][t]; E[t] = v; }
        __syncthreads();
        if (t == 0) {
            LMState L = sL;
            lm_transition_local(Q, &L, E);
            sL = L;
            const bool cand = L.phase == PHASE_CAND;
            for (int i = 0; i < 6; i++) s_x[i] = cand ? L.xc[i] : L.x[i];
            s_done = L.done;
        }
        __syncthreads();
    }
    if (t == 0) {
        *Sg = sL;
        if (sL.done) {                                       // chain mode: what the host and the next round need (see lm_advance)
            if (log) {
                for (int i = 0; i < 6; i++) log->x[i] = sL.x[i];
                log->initial_cost = sL.initial_cost; log->final_cost = sL.cost;
                log->termination = sL.termination; log->iter = sL.iter; log->evals = sL.evals; log->n_valid = sL.n_valid;
            }
            if (pose_out) {
                double xf[6];
                for (int i = 0; i < 6; i++) xf[i] = sL.x[i];
                PoseScalars S;
#ifndef VELO_X2
                pose_scalars_compute(xf, &S);
#endif
                pose_out->P = S;
                __threadfence();
                pose_out->ready = 1;
            }
        }
    }
}
__global__ void __launch_bounds__(kEvalThreads)
lm_solve_small_kernel(EvalArgs A, LMParams Q, LMState* Sg, const double* __restrict__ x_in, const int* __restrict__ n_valid,
                      int nb_icp, int nb_vis, int max_sweeps, PoseRecord* __restrict__ pose_out, SolveLog* __restrict__ log)
#if VELO_DEF_LMB
{
    lm_solve_small_body(A, Q, Sg, x_in, n_valid, nb_icp, nb_vis, max_sweeps, pose_out, log);
}
#else
;
#endif
// the solves of a lock-step group, one workgroup per context (the reference's constants in batches: 640 queries per pair).  The items
// come BY VALUE, up to kItemsByValue per launch: read through a pointer their fields stay live across the body's stores (424 bytes of
// scratch, ~11 us per launch); in the argument segment they are constants the compiler reloads where it needs them.
constexpr int kItemsByValue = 4;
struct LMBatchPack { LMBatchItem item[kItemsByValue]; };
static_assert(sizeof(LMBatchPack) + 256 <= 4096, "the pack must fit the kernel argument segment");
__global__ void __launch_bounds__(kEvalThreads)
lm_solve_small_batch_kernel(LMParams Q, LMBatchPack P, int max_sweeps)
#if VELO_DEF_LMB
{
    const LMBatchItem& it = P.item[blockIdx.x];
    lm_solve_small_body(it.A, Q, it.S, it.xd, it.n_valid, it.nb_icp, it.nb_vis, max_sweeps, it.pose_out, it.log);
}
#else
;
#endif
__global__ void __launch_bounds__(kEvalThreads)
lm_solve_small_icp_batch_kernel(LMParams Q, LMBatchPack P, int max_sweeps)
#if VELO_DEF_LMB
{
    const LMBatchItem& it = P.item[blockIdx.x];
    lm_solve_small_body<false>(it.A, Q, it.S, it.xd, it.n_valid, it.nb_icp, 0, max_sweeps, it.pose_out, it.log);
}
#else
;
#endif
__global__ void __launch_bounds__(kEvalThreads)
lm_solve_small_icp_kernel(EvalArgs A, LMParams Q, LMState* Sg, const double* __restrict__ x_in, const int* __restrict__ n_valid,
                          int nb_icp, int max_sweeps, PoseRecord* __restrict__ pose_out, SolveLog* __restrict__ log)
#if VELO_DEF_LMB
{
    lm_solve_small_body<false>(A, Q, Sg, x_in, n_valid, nb_icp, 0, max_sweeps, pose_out, log);
}
#else
;
#endif

// ---- seam 2 by value: a batch of residual functors (costfunctions.h:17-220) at one pose -----------------------------------
// One record per functor: kind (ResidualType order 0..3, 4 = cost3DPD) and the constructor arguments widened to double in the
// reference's order.  One thread per record writes the RAW residuals (no loss) and the 6-column Jacobian Ceres' autodiff would
// produce -- the same device functions the LM sweeps use, so a parity check of this kernel is a parity check of theirs.
struct FunctorRec { int kind; int reserved; double c[9]; };
static_assert(sizeof(FunctorRec) == 80, "velo_functor layout");

__global__ void __launch_bounds__(256)
functor_batch_kernel(const FunctorRec* __restrict__ f, int n, const double* __restrict__ xd, double* __restrict__ res, double* __restrict__ jac)
#if VELO_DEF_LMB
{
    __shared__ PoseRot s_R, s_Rinv;
    __shared__ double s_t[3];
    if (threadIdx.x == 0) {
        const double x[6] = {xd[0], xd[1], xd[2], xd[3], xd[4], xd[5]};
        pose_rot_init(x, &s_R);
        const double m[3] = {-x[0], -x[1], -x[2]};
        pose_rot_init(m, &s_Rinv);
        s_t[0] = x[3]; s_t[1] = x[4]; s_t[2] = x[5];
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FunctorRec F = f[i];
    const double t[3] = {s_t[0], s_t[1], s_t[2]};
    double r[3] = {0.0, 0.0, 0.0}, J[18];
#pragma unroll
    for (int k = 0; k < 18; k++) J[k] = 0.0;
    switch (F.kind) {
        case 0: res_3d3d(s_R, t, F.c, F.c + 3, r, J); break;
        case 1: res_3d2d(s_R, t, F.c, F.c + 3, F.c + 5, r, J); break;
        case 2: res_2d3d(s_Rinv, t, F.c, F.c + 3, F.c + 5, r, J); break;
        case 3: res_2d2d(s_R, t, F.c, F.c + 2, F.c + 4, r, J); break;
        default: res_3dpd(s_R, t, F.c, F.c + 3, F.c + 6, r, J); break;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) res[(size_t)3 * i + k] = r[k];
    if (jac) {
#pragma unroll
        for (int k = 0; k < 18; k++) jac[(size_t)18 * i + k] = J[k];
    }
}
#else
;
#endif

// ---- template kernels: instantiated by the unit that owns the family, `extern template` for everybody else --------------------------------
#ifndef VELO_UNIT_LM_ONLY
#if VELO_DEF_LOAD
#define VELO_INST_LOAD template
#else
#define VELO_INST_LOAD extern template
#endif
#if VELO_DEF_ASSOC
#define VELO_INST_ASSOC template
#else
#define VELO_INST_ASSOC extern template
#endif
VELO_INST_LOAD __global__ void scan_lookback_kernel<kLbItemsSmall>(int*, int, unsigned long long*, int*, int*);
VELO_INST_LOAD __global__ void scan_lookback_kernel<kLbItemsLarge>(int*, int, unsigned long long*, int*, int*);
VELO_INST_LOAD __global__ void advance_scan_kernel<kLbItemsSmall>(AdvBatch);
VELO_INST_LOAD __global__ void advance_scan_kernel<kLbItemsLarge>(AdvBatch);
#define VELO_V5_ARGS (PoseScalars, const PoseRecord*, int*, GridView, const float4*, int, int, const float4*, const int*, unsigned, double, int, float, AssocOut, int, const int*, int, int)
#define VELO_V3_ARGS (PoseScalars, GridView, const float4*, const int*, int, int, const float4*, const int*, const int*, unsigned, double, int, float, AssocOut, int, int, int)
#define VELO_CLUSTER_ARGS (GridView, AssocQueue, const float4*, const int*, int, int, const float4*, const int*, const int*, unsigned, double, float, AssocOut, int)
VELO_INST_ASSOC __global__ void assoc_search_v5_kernel<4, 5, false, 2, 0> VELO_V5_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v5_kernel<4, 5, false, 2, 1> VELO_V5_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v5_batch_kernel<4, 5, false, 2, 0>(AssocBatch);
VELO_INST_ASSOC __global__ void assoc_search_v5_batch_kernel<4, 5, false, 2, 1>(AssocBatch);
VELO_INST_ASSOC __global__ void assoc_search_v5_batch_kernel<4, 5, false, 2, 2>(AssocBatch);
#ifdef VELO_DIAGNOSTICS   // the A/B instantiations of the tools' build
VELO_INST_ASSOC __global__ void assoc_search_v5_kernel<4, 5, true, 2, 1> VELO_V5_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v5_kernel<4, 5, false, 4, 1> VELO_V5_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v5_kernel<4, 6, false, 2, 1> VELO_V5_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v5_kernel<4, 6, false, 2, 0> VELO_V5_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v5_kernel<4, 7, false, 2, 0> VELO_V5_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v5_kernel<4, 8, false, 2, 0> VELO_V5_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<1, 1, false> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<1, 1, true> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<2, 1, false> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<2, 1, true> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<4, 5, false> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<4, 5, true> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<4, 6, false> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<4, 6, true> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<4, 7, false> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<4, 7, true> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<4, 8, false> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<4, 8, true> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<8, 8, false> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_search_v3_kernel<8, 8, true> VELO_V3_ARGS;
VELO_INST_ASSOC __global__ void assoc_cluster_kernel<2, 1> VELO_CLUSTER_ARGS;
VELO_INST_ASSOC __global__ void assoc_cluster_kernel<4, 6> VELO_CLUSTER_ARGS;
VELO_INST_ASSOC __global__ void assoc_cluster_kernel<8, 6> VELO_CLUSTER_ARGS;
#endif
#undef VELO_V5_ARGS
#undef VELO_V3_ARGS
#undef VELO_CLUSTER_ARGS
#endif  // VELO_UNIT_LM_ONLY

}  // namespace velo
