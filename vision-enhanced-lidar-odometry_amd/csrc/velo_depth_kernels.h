// velo_depth_kernels.h -- SURVEY.md 8(f) row 3: projectLidarToCamera + featureDepthAssociation (reference velo.h:329-497)
// on the device.  Included by velo_hip.hip after velo_kernels.h (uses its scan kernels); gfx950 only.
//
// Data layout: the occlusion stacks of all rings live in two float4 arrays parallel to the ring-major cloud --
// ring s owns slots [off[s], off[s] + cnt[s]) of
//     pstack[j] = {c.x, c.y, shifted z, bits(index of the point inside its ring)}      (`projection`, `projected_points`)
//     vstack[j] = {x, y, z, 0} of the UN-shifted point                                  (`scans_valid`, velo.h:368)
// so a ring's list can never outgrow the ring and no compaction is needed before the keypoint search.
#pragma once
#include <hip/hip_runtime.h>

namespace velo {

struct CamWindow { float tx, ty, tz; double min_x, max_x, min_y, max_y; };

// One 256-thread workgroup per ring.  The pass over a ring is sequential by definition (every decision looks at the current
// stack top), so the work is split: all threads project + window-test 1,024 points at a time (4 consecutive points per thread)
// and compact the survivors, in ring order, into LDS with a workgroup scan; wave 0 then replays the stack rule 64 survivors at
// a time.  A run whose x never decreases (and does not start left of the stack top) can neither pop nor be dropped
// (velo.h:351-365 both require c.x < top.x), so it is pushed by 64 lanes at once; only runs that contain a descent -- depth
// discontinuities seen with parallax -- are replayed entry by entry by lane 0.
constexpr int kProjChunk = 1024;
__global__ void __launch_bounds__(256)
project_ring_kernel(const float4* __restrict__ pts, const int* __restrict__ off, int n_rings, CamWindow W,
                    float4* __restrict__ pstack, float4* __restrict__ vstack, int* __restrict__ cnt)
#if VELO_DEF_LOAD
{
    __shared__ float4 s_e[kProjChunk];                               // {c.x, c.y, shifted z, bits(index inside the ring)}
    __shared__ float4 s_p[kProjChunk];                               // un-shifted point
    const int ring = blockIdx.x;
    if (ring >= n_rings) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int base = off[ring], n = off[ring + 1] - base;
    int top = 0;                    // stack height (wave 0, uniform)
    float top_x = 0.f, top_z = 0.f; // projection x / shifted z of the stack top (wave 0, uniform)
    for (int c0 = 0; c0 < n; c0 += kProjChunk) {
        float4 p[4], e[4];
        int in[4], mine = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = pts[base + min(c0 + 4 * tid + u, n - 1)];      // all four loads in flight
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = c0 + 4 * tid + u;
            const float px = p[u].x + W.tx, py = p[u].y + W.ty, pz = p[u].z + W.tz;      // velo.h:346 (float adds)
            const float cx = px / pz, cy = py / pz;                                     // velo.h:347
            in[u] = (i < n && pz > 0.f && (double)cx >= W.min_x && (double)cx < W.max_x && (double)cy >= W.min_y && (double)cy < W.max_y) ? 1 : 0;   // velo.h:348-349
            e[u] = make_float4(cx, cy, pz, __int_as_float(i));
            mine += in[u];
        }
        int total;
        int pos = block_exclusive_scan(mine, &total);
#pragma unroll
        for (int u = 0; u < 4; u++) if (in[u]) { s_e[pos] = e[u]; s_p[pos] = make_float4(p[u].x, p[u].y, p[u].z, 0.f); pos++; }
        __syncthreads();
        if (tid < 64) {
            for (int k0 = 0; k0 < total; k0 += 64) {
                const int m = min(64, total - k0);
                const bool valid = lane < m;
                const float4 ek = s_e[k0 + (valid ? lane : 0)], pk = s_p[k0 + (valid ? lane : 0)];
                const float prev_x = __shfl_up(ek.x, 1);
                const bool descent = valid && (lane == 0 ? (top > 0 && ek.x < top_x) : ek.x < prev_x);
                if (__ballot(descent) == 0ull) {                     // nothing can pop or be dropped: push the whole run
                    if (valid) { pstack[base + top + lane] = ek; vstack[base + top + lane] = pk; }
                    __threadfence_block();                           // a later pop (lane 0) may read these entries back
                    top += m;
                    top_x = __shfl(ek.x, m - 1); top_z = __shfl(ek.z, m - 1);
                } else {
                    if (lane == 0) {
                        for (int k = k0; k < k0 + m; k++) {
                            const float4 en = s_e[k];
                            while (top > 0 && en.x < top_x && en.z < top_z) {    // velo.h:351-358: pop what the new point occludes
                                top--;
                                if (top > 0) { const float4 t = pstack[base + top - 1]; top_x = t.x; top_z = t.z; }
                            }
                            if (top > 0 && en.x < top_x && en.z > top_z) continue;   // velo.h:360-365: the new point is occluded
                            pstack[base + top] = en;
                            vstack[base + top] = s_p[k];
                            top++; top_x = en.x; top_z = en.z;
                        }
                    }
                    top = __shfl(top, 0); top_x = __shfl(top_x, 0); top_z = __shfl(top_z, 0);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) cnt[ring] = top;
}
#else
;
#endif

// util::linterpolate (utility.h:7-29), float arithmetic, products and sum left unfused
__device__ __forceinline__ float lerp_f(float p1, float p2, float start, float end, float mid) {
    const float a = (mid - start) / (end - start);
    const float b = 1.f - a;
    return p1 * b + p2 * a;
}
__device__ __forceinline__ void lerp_p(const float p1[3], const float p2[3], float start, float end, float mid, float out[3]) {
    const float a = (mid - start) / (end - start);
    const float b = 1.f - a;
    out[0] = p1[0] * b + p2[0] * a;
    out[1] = p1[1] * b + p2[1] * a;
    out[2] = p1[2] * b + p2[2] * a;
}

// One wave per keypoint, one lane per ring (64 rings at a time).  The reference walks the rings in order carrying
// `last_interp` (velo.h:394-491); that state is just "ring s-1 bracketed the keypoint, at segment mid" -- a pure function
// of ring s-1 -- so every lane runs the reference's bisection on its own ring, lane s reads lane s-1's result, and the
// first lane whose test passes is the ring at which the reference stops.
__global__ void __launch_bounds__(256)
depth_assoc_kernel(const float2* __restrict__ kps, int n_kp, const float4* __restrict__ pstack, const float4* __restrict__ vstack,
                   const int* __restrict__ off, const int* __restrict__ cnt, int n_rings, double thresh,
                   float4* __restrict__ kp_point, int* __restrict__ flag)
#if VELO_DEF_LOAD
{
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n_kp) return;
    const float2 kp = kps[k];
    int carry_found = 0, carry_mid = 0;
    int hit = 0;
    for (int r0 = 0; r0 < n_rings; r0 += 64) {
        const int s = r0 + lane;
        int found = 0, mid = 0, base = 0;
        if (s < n_rings) {
            base = off[s];
            const int n = cnt[s];
            if (n > 1) {                                             // velo.h:397-400
                int lo = 0, hi = n - 2;
                while (lo <= hi) {                                   // velo.h:401-407, same probes on non-monotone lists too
                    mid = (lo + hi) / 2;
                    if (pstack[base + mid].x > kp.x) hi = mid - 1;
                    else if (pstack[base + mid + 1].x <= kp.x) lo = mid + 1;
                    else { found = 1; break; }
                }
            }
        }
        int pfound = __shfl_up(found, 1), pmid = __shfl_up(mid, 1), pbase = __shfl_up(base, 1);
        if (lane == 0) { pfound = carry_found; pmid = carry_mid; pbase = (s > 0 && s <= n_rings) ? off[s - 1] : 0; }
        bool ok = false;
        float4 a0, a1, b0, b1;
        if (found && pfound && s > 0) {
            a0 = pstack[base + mid]; a1 = pstack[base + mid + 1];
            b0 = pstack[pbase + pmid]; b1 = pstack[pbase + pmid + 1];
            ok = ((a0.y > kp.y) != (b0.y > kp.y)) && (double)fabsf(a0.x - a1.x) < thresh && (double)fabsf(b0.x - b1.x) < thresh;   // velo.h:412-422
        }
        const unsigned long long m = __ballot(ok);
        if (m) {
            if (lane == (int)__ffsll((long long)m) - 1) {
                const float4 va0 = vstack[base + mid], va1 = vstack[base + mid + 1];
                const float4 vb0 = vstack[pbase + pmid], vb1 = vstack[pbase + pmid + 1];
                const float pa0[3] = {va0.x, va0.y, va0.z}, pa1[3] = {va1.x, va1.y, va1.z};
                const float pb0[3] = {vb0.x, vb0.y, vb0.z}, pb1[3] = {vb1.x, vb1.y, vb1.z};
                float i1[3], i2[3], out[3];
                lerp_p(pa0, pa1, a0.x, a1.x, kp.x, i1);               // velo.h:447-452
                lerp_p(pb0, pb1, b0.x, b1.x, kp.x, i2);               // velo.h:453-458
                const float i1y = lerp_f(a0.y, a1.y, a0.x, a1.x, kp.x);   // velo.h:459-464
                const float i2y = lerp_f(b0.y, b1.y, b0.x, b1.x, kp.x);   // velo.h:465-470
                lerp_p(i1, i2, i1y, i2y, kp.y, out);                  // velo.h:472-477
                kp_point[k] = make_float4(out[0], out[1], out[2], 0.f);
            }
            hit = 1;
            break;
        }
        carry_found = __shfl(found, 63); carry_mid = __shfl(mid, 63);
    }
    if (lane == 0) flag[k] = hit;
}
#else
;
#endif

// has_depth[k] = flag ? exclusive count : -1 ; the interpolated points are appended in keypoint order (velo.h:481-483)
__global__ void depth_compact_kernel(const int* __restrict__ flag, const int* __restrict__ excl, const float4* __restrict__ kp_point, int n_kp,
                                     int* __restrict__ has_depth, float4* __restrict__ out)
#if VELO_DEF_LOAD
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_kp) return;
    if (flag[k]) { has_depth[k] = excl[k]; out[excl[k]] = kp_point[k]; }
    else has_depth[k] = -1;
}
#else
;
#endif

}  // namespace velo
