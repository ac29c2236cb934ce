/*
 * velo_hip.h -- C-ABI of the MI355X (gfx950) scan-matching core.
 *
 * This is the drop-in boundary for ONE path of lichunshang/vision-enhanced-lidar-odometry (VELO):
 * the frame-to-frame registration loop  frameToFrame()  (reference velo.h:598-919).  The reference has
 * no FFI/plugin interface; the three de-facto seams the path sits behind are
 *     (1) frameToFrame(...)                     velo.h:598-614, sole caller main.cpp:388-405
 *     (2) the residual functors                 costfunctions.h:17-220
 *     (3) ceres::CostFunction::Evaluate [3P]    used at velo.h:683-689,710-718,744-752,777-785,875-891
 * Each entry point below names the reference lines it replaces.  include/velo_frame_to_frame.hpp is the
 * header-only C++ adaptor that offers seam (1)'s parameter list on top of these calls; INTEGRATION.md
 * shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only; every call returns an int status (VELO_OK == 0, < 0 error)
 * and never throws; the caller owns host buffers, the context owns device buffers; one context per host
 * thread (a context is not internally thread-safe), several contexts may share one GPU.
 * x = (omega[3], t[3]) is the reference's `double transform[6]` (velo.h:610): p_prev = R(omega) p_cur + t.
 *
 * Environment: the library reads exactly five variables, at velo_create / at the first batch call; none of them changes a result
 * (every setting gives bit-identical poses, tables and summaries; tests/test_gpu_parity.py compares them):
 *     VELO_CHAIN=0           frame_to_frame with a host round trip after every solve instead of one chain of launches per call
 *     VELO_CHAIN_MARGIN=n    LM launches enqueued per solve beyond the previous call's count (default: 1..3 from the recent history)
 *     VELO_BATCH_LOCKSTEP=0  velo_frame_to_frame_batch / velo_register_batch with one host thread per context instead of lock-step groups
 *     VELO_BATCH_GROUPS=n    number of lock-step groups of a batch (default: by batch size, at most 4)
 *     VELO_SPIN=1            hipDeviceScheduleSpin for the calling process
 * The reference's own knobs are compile-time constants (kitti.h:3-35) and arrive through velo_params.  Kernel variants, grid shapes and
 * diagnostics are NOT an environment surface of this library: they exist only in the tools' build (-DVELO_DIAGNOSTICS,
 * libvelo_hip_diag.so; tools/README.md), which the variant parity tests and the dev tools load explicitly.
 */
#ifndef VELO_HIP_H_
#define VELO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VELO_OK 0
#define VELO_ERR_INVALID (-1)   /* bad argument (NULL, negative size, empty ring, ...) */
#define VELO_ERR_HIP (-2)       /* a HIP runtime call failed; velo_last_error() has the text */
#define VELO_ERR_STATE (-3)     /* call order violated (e.g. associate before set_target) */
#define VELO_ERR_COMM (-4)      /* RCCL failure */
#define VELO_ERR_NODEVICE (-5)  /* no gfx950 device visible: the product path has NO CPU fallback */

/* residual_type values, same order as the reference enum ResidualType (velo.h:3-8) */
#define VELO_RESIDUAL_3D3D 0
#define VELO_RESIDUAL_3D2D 1
#define VELO_RESIDUAL_2D3D 2
#define VELO_RESIDUAL_2D2D 3
/* velo_functor.kind only: the ICP functor cost3DPD (costfunctions.h:17-55), which has no ResidualType value */
#define VELO_FUNCTOR_3DPD 4

/* ceres::TerminationType [3P] for one LM solve (velo.h:902) */
#define VELO_CONVERGENCE 0
#define VELO_NO_CONVERGENCE 1
#define VELO_FAILURE 2

#define VELO_MAX_SOLVES 64

/* Every tunable of the path.  Defaults = the reference's compile-time constants (kitti.h:8-10,20-32,
 * main.cpp:43-45) and the Ceres defaults the reference leaves untouched at velo.h:897-902. */
typedef struct velo_params {
    int32_t icp_skip;              /* kitti.h:8   200 (BASELINE configs 2-5 use 1) */
    int32_t f2f_iterations;        /* kitti.h:9   2 */
    int32_t icp_iterations;        /* kitti.h:10  3 */
    int32_t enable_icp;            /* velo.h:613,806; caller passes true (main.cpp:404) */
    int32_t enable_2d2d;           /* main.cpp:44 ENABLE_2D2D (defined) */
    int32_t enable_3d2d;           /* main.cpp:45 ENABLE_3D2D (defined) */
    int32_t max_num_iterations;    /* ceres default 50 */
    int32_t max_consecutive_invalid_steps; /* ceres default 5 */
    double weight_3D2D;            /* kitti.h:20  10 */
    double weight_2D2D;            /* kitti.h:21  500 */
    double weight_3DPD;            /* kitti.h:22  1 */
    double loss_thresh_3D2D;       /* kitti.h:23  0.01 */
    double loss_thresh_2D2D;       /* kitti.h:24  2e-5 */
    double loss_thresh_3DPD;       /* kitti.h:25  0.1 */
    double loss_thresh_3D3D;       /* kitti.h:26  0.04 */
    double outlier_reject;         /* kitti.h:30  5 */
    double correspondence_thresh_icp; /* kitti.h:31  0.5 (a SQUARED distance, velo.h:829) */
    double icp_norm_condition;     /* kitti.h:32  1e-5 */
    double function_tolerance;     /* ceres default 1e-6 */
    double gradient_tolerance;     /* ceres default 1e-10 */
    double parameter_tolerance;    /* ceres default 1e-8 */
    double initial_trust_region_radius; /* 1e4 */
    double max_trust_region_radius;     /* 1e16 */
    double min_trust_region_radius;     /* 1e-32 */
    double min_relative_decrease;       /* 1e-3 */
    double min_lm_diagonal;             /* 1e-6 */
    double max_lm_diagonal;             /* 1e32 */
} velo_params;

/* What velo.h:627-654 gathers for one feature match before it chooses residual types. */
typedef struct velo_match {
    float p3_1[3];   /* 3-D keypoint, frame1 (current), camera-0 frame     velo.h:650 */
    float p3_2[3];   /* 3-D keypoint or landmark, frame2 (previous)        velo.h:634-648 */
    float p2_1[2];   /* canonical 2-D observation in frame1                velo.h:653 */
    float p2_2[2];   /* canonical 2-D observation in frame2                velo.h:654 */
    float t_cam[3];  /* cam_trans[cam]                                     kitti.h:76-78 */
    int32_t cam;
    int32_t point1;  /* passed through into good_matches                   velo.h:628 */
    int32_t point2;  /*                                                    velo.h:629 */
    uint8_t d1;      /* has_depth in frame1                                velo.h:631 */
    uint8_t d2;      /* has_depth in frame2 or landmark                    velo.h:632,644 */
    uint8_t pad[2];
} velo_match;

/* One entry of good_matches[cam] / residual_type[cam] (velo.h:691-692,719-720,754-755,787-788). */
typedef struct velo_good_match {
    int32_t cam;
    int32_t point1;
    int32_t point2;
    int32_t residual_type;
} velo_good_match;

/* One point-to-plane correspondence as the association loop builds it (velo.h:822-874). */
typedef struct velo_corr {
    int32_t valid;       /* 0 = skipped by one of the `continue`s at velo.h:849-851,873 */
    int32_t ring_i;      /* np_s_i */
    int32_t idx_i;       /* np_i   */
    int32_t ring_j;      /* np_s_j */
    int32_t idx_j;       /* np_j   */
    int32_t idx_k;       /* np_k   */
    int32_t src_ring;    /* sm  */
    int32_t src_idx;     /* smi */
    float dist_i;        /* np_dist_i (squared, float) */
    float dist_j;        /* np_dist_j */
    float p[3];          /* pointM_untransformed  velo.h:809 */
    float n[3];          /* N                     velo.h:872-874 */
    float v0[3];         /* v0                    velo.h:869 */
} velo_corr;

/* Target-sharded mode (BASELINE config 5): what one rank knows about a query after searching ONLY the
 * whole target rings it holds.  Ring ownership is disjoint, so the query's owner merges `world` such records with
 * best1 = min key1, best2 = min(key1 of the other ranks, key2 of the winner) and builds the plane from the payload. */
typedef struct velo_partial {
    uint64_t key1, key2;   /* (float bits of d^2) << 32 | global ring-major point index; "absent" compares above every real key */
    int32_t ring1, ring2;  /* global ring ids, -1 when absent */
    int32_t idx1;          /* np_i   (index inside ring1) */
    int32_t idx_k;         /* np_k   (ring neighbour of np_i, velo.h:852-863) */
    int32_t idx2;          /* np_j   (index inside ring2) */
    int32_t pad;
    float v0[3];           /* scans_S[ring1][idx1] */
    float v2[3];           /* scans_S[ring1][idx_k] */
    float v1[3];           /* scans_S[ring2][idx2] */
    float pad2;
} velo_partial;

typedef struct velo_solve_summary {
    int32_t termination;       /* VELO_CONVERGENCE / NO_CONVERGENCE / FAILURE */
    int32_t lm_iterations;     /* trust-region iterations, successful or not */
    int32_t evaluations;       /* residual(+Jacobian) sweeps over all blocks */
    int32_t n_icp_valid;       /* point-to-plane blocks in this solve */
    int32_t n_visual_blocks;   /* visual residual blocks in this solve */
    int32_t n_visual_residuals;
    double initial_cost;
    double final_cost;
} velo_solve_summary;

/* --- "next" row 4 of SURVEY.md 8(f): batched triangulatePoint (velo.h:1027-1130) ------------------------------ */
#define VELO_TRI_OBS_3D 0   /* triangulation3D (costfunctions.h:338-375), TrivialLoss (velo.h:1078) */
#define VELO_TRI_OBS_2D 1   /* triangulation2D (costfunctions.h:288-336), ScaledLoss(CauchyLoss(loss_thresh_3D2D), weight_3D2D) (velo.h:1116-1119) */
typedef struct velo_tri_obs {  /* one entry of keypoint_obs3[cam] / keypoint_obs2[cam] of one landmark: 24 bytes */
    int32_t kind;              /* VELO_TRI_OBS_3D / VELO_TRI_OBS_2D */
    int32_t frame;             /* the map key: index into camera_poses */
    int32_t cam;               /* camera; a 2-D observation uses cam_trans[cam] */
    float s[3];                /* 3-D: the observed point (camera-0 frame of `frame`); 2-D: canonical (x, y), s[2] unused */
} velo_tri_obs;
typedef struct velo_tri_result {   /* per landmark, optional: 24 bytes */
    int32_t n_solves;          /* ceres::Solve calls the reference makes for it: 0 (no observation), 1 or 2 (velo.h:1080-1083,1123) */
    int32_t termination;       /* of the last solve */
    int32_t lm_iterations;     /* of the last solve */
    int32_t evaluations;       /* over all solves */
    double final_cost;
} velo_tri_result;

/* residualStats (velo.h:921-1025): what the reference prints after every f2f iteration (velo.h:909) -- per residual type the
 * median (sorted[size / 2]), mean and count of the block norms at the current pose, loss functions NOT applied
 * (3D3D: |r|_2 of 3; 3D2D / 2D3D: |r|_2 of 2; 2D2D, 3DPD: |r|), plus the cost 1/2 sum r^2 of that evaluation.
 * type[k]: k = VELO_RESIDUAL_3D3D, _3D2D, _2D3D, _2D2D, VELO_FUNCTOR_3DPD.  Computed on the device (radix select, no sort). */
typedef struct velo_residual_stat { double median, mean; int64_t count; } velo_residual_stat;
typedef struct velo_residual_stats {
    velo_residual_stat type[5];
    double cost;
    int32_t n_blocks, n_residuals;     /* problem.NumResidualBlocks(), problem.NumResiduals() */
} velo_residual_stats;                 /* 136 bytes */
#define VELO_MAX_STATS 4               /* f2f iterations whose statistics a summary keeps */

typedef struct velo_summary {
    int32_t n_solves;
    int32_t n_assoc_rounds;
    int32_t n_queries;                 /* Nq per round */
    int32_t n_target;                  /* Nt */
    uint64_t algorithmic_bytes;        /* sum B_assoc + sum B_eval */
    uint64_t assoc_bytes;              /* the B_assoc part */
    double assoc_kernel_ms;            /* sum of association-search launch durations (HIP events) when timing is on */
    int32_t assoc_kernel_launches;
    int32_t eval_kernel_launches;
    double eval_kernel_ms;
    velo_solve_summary solves[VELO_MAX_SOLVES];
    int32_t n_residual_stats;          /* f2f iterations recorded below (0 unless velo_set_residual_stats(ctx, 1)) */
    int32_t reserved;
    velo_residual_stats residual_stats[VELO_MAX_STATS];   /* [iter - 1]: residualStats at the end of f2f iteration iter */
} velo_summary;

typedef struct velo_ctx velo_ctx;

/* --- lifetime ------------------------------------------------------------------------------------ */
/* Binds a context (device buffers + one HIP stream) to `device`.  Replaces cv::cuda::setDevice (main.cpp:64)
 * as the device selection of the path.  Fails with VELO_ERR_NODEVICE when no GPU is visible. */
int velo_create(velo_ctx** out, int device);
int velo_destroy(velo_ctx* ctx);
const char* velo_last_error(void);
const char* velo_version(void);

/* --- configuration: kitti.h:8-10,20-32 + ceres::Solver::Options defaults (velo.h:897-901) ----------- */
int velo_default_params(velo_params* p);
int velo_set_params(velo_ctx* ctx, const velo_params* p);
int velo_get_params(const velo_ctx* ctx, velo_params* p);
/* Launch timing with HIP events (off by default).  enable = 1: the association launches of a call, summed into
 * velo_summary::assoc_kernel_ms.  enable = 2: every instrumented launch -- association, seeds, LM (sweep + step), target index build -- is
 * COUNTED per kernel name, and every 8th launch of a name is bracketed by the kernel's own start / stop events (hipExtLaunchKernelGGL;
 * a bracket costs ~5 us of queue time, so bracketing every launch -- enable = 3 -- slows a chain of 20-us launches by a quarter).  The
 * brackets are read after the call's final synchronisation and accumulated until velo_get_kernel_times collects them: `ms` = bracketed
 * time scaled by launches / sampled (what bench.py's `kernels` list reports).  algorithmic_bytes: SURVEY.md 8(d)'s figure for the
 * launches of that kernel -- B_assoc per association round served, B_eval per LM evaluation -- and, for the index-build and seed kernels,
 * what the kernel must move given the data layout.  Launches shared by the contexts of a lock-step group are logged on the group's
 * first context.  (At levels 2 and 3 the association launches are bracketed one by one, as at level 1.) */
int velo_set_timing(velo_ctx* ctx, int enable);
typedef struct velo_kernel_time {
    char name[48];
    double ms;                         /* estimated sum of launch durations: bracketed time x launches / sampled */
    int64_t launches;
    int64_t sampled;                   /* launches that were bracketed */
    uint64_t algorithmic_bytes;
} velo_kernel_time;                    /* 80 bytes */
int velo_get_kernel_times(velo_ctx* ctx, velo_kernel_time* out, int32_t capacity, int32_t* n, int32_t reset);
/* residualStats after every f2f iteration (velo.h:909) into velo_summary::residual_stats (off by default: the reference only
 * prints them; switched on, a call evaluates between the iterations and is therefore driven round by round from the host). */
int velo_set_residual_stats(velo_ctx* ctx, int enable);

/* --- inputs --------------------------------------------------------------------------------------- */
/* Target = frame2 rings.  Replaces `scans_S` + `kd_trees` (velo.h:606-607) and the per-ring
 * KdTreeFLANN::setInputCloud of ScanData (lru.h:17-20): the spatial index is built on the device here.
 * xyz: first float of point 0; stride_bytes between points (12 packed, 16 for pcl::PointXYZ);
 * ring_offsets[n_rings+1]: ring r = points [ring_offsets[r], ring_offsets[r+1]), each ring an ordered cyclic
 * polyline (kitti.h:158-183).  on_device != 0: xyz is already a device pointer (stays caller-owned, copied).
 * Device pointers (every entry point with an on_device flag, velo_scan_ref included): the memory must be on the context's
 * device, float32, the three coordinates of a point contiguous, and COMPLETE before the call -- the library reads it on the
 * context's own non-blocking stream, which is not ordered against the stream that produced the data: synchronise that
 * stream (or wait for its event) first, and do not overwrite the buffer until the call has returned. */
int velo_set_target(velo_ctx* ctx, const float* xyz, int64_t stride_bytes, const int32_t* ring_offsets,
                    int32_t n_rings, int on_device);
/* Target-sharded variant: this context holds only `n_rings` WHOLE rings of the target (keeps the +-1 ring neighbour of
 * velo.h:852-856 local); `first_ring` is the global id of the first of them and `first_point` the global ring-major index of
 * its first point.  ring_offsets are local ([0] == 0).  velo_set_target == velo_set_target_part(..., 0, 0, ...). */
int velo_set_target_part(velo_ctx* ctx, const float* xyz, int64_t stride_bytes, const int32_t* ring_offsets,
                         int32_t n_rings, int32_t first_ring, int32_t first_point, int on_device);
/* Source = frame1 rings (`scans_M`, velo.h:605).  Queries are every icp_skip-th point of each ring (velo.h:807). */
int velo_set_source(velo_ctx* ctx, const float* xyz, int64_t stride_bytes, const int32_t* ring_offsets,
                    int32_t n_rings, int on_device);
/* "Next" row 1 of SURVEY.md 8(f): the step immediately before the path, on the device.  Replaces loadPoints + segmentPoints
 * (kitti.h:121-185) for one frame: `xyzr` are raw Velodyne records in FILE order (x, y, z, reflectance -> stride 16 for a
 * KITTI .bin), a new ring starts where x > 0 and the sign of y flips (kitti.h:166), points are stored in the camera-0 frame
 * (velo_to_cam: row-major 4x4, float arithmetic like pcl::transformPointCloud) and every ring is reordered
 * new[i] = old[n-1-((i + n/2) % n)] (kitti.h:180).  The result becomes this context's source (as_target == 0) or target. */
int velo_set_scan_velodyne(velo_ctx* ctx, int32_t as_target, const float* xyzr, int64_t stride_bytes, int32_t n_points,
                           const float velo_to_cam[16], int on_device);
/* The scan cache of the odometry loop (replaces the ScansLRU look-up of the previous frame, lru.h:31-61, main.cpp:233,380): the cloud
 * this context holds as SOURCE (frame k, already segmented on the device) becomes its TARGET for the next registration -- device
 * buffers are swapped, nothing is uploaded or segmented again; only the target index (ring ids, grid) is built.  Afterwards the
 * context has no source until the next velo_set_source / velo_set_scan_velodyne(as_target = 0). */
int velo_source_to_target(velo_ctx* ctx);
/* Scan-to-map batches (BASELINE configs 4-5: many scans against ONE accumulated map): `dst` takes the target `src` holds -- cloud
 * and search index -- BY REFERENCE: no copy, no second index build, one 110 MB map in HBM instead of one per context.  The shared
 * target is read-only; it lives as long as any context holds it, and a context that loads a new target simply lets go of it.
 * Both contexts must be on the same device and work with the same gates. */
int velo_share_target(velo_ctx* dst, velo_ctx* src);
/* Device-resident scan cache: the ScansLRU of the reference (lru.h:31-61, `size = 50`; look-ups main.cpp:216,349-350,544) with the
 * scans kept in HBM instead of host memory -- the cloud as the path stores it (camera-0 frame, ring-major) and, for scans stored from
 * a context's TARGET side, the search index built for it (the role of ScanData::trees, lru.h:9,17-20), so a frame that is matched again
 * (frame - dframe for every dframe, main.cpp:306-350) is neither read, segmented, uploaded nor indexed a second time.
 * store: device-to-device copy of the scan a context holds (of_target: its target incl. index / its source) under `frame`; a frame that
 *        is already cached is replaced; the least recently used entry is dropped beyond `capacity` (lru.h:52-57).
 * load:  device-to-device copy into a context as its target or source, and the entry becomes the most recent one (lru.h:42-47).
 *        Loading as target reuses the cached index when it was built for the same gates, otherwise (entries stored from a source,
 *        other params) the index is built from the cached cloud.  Any number of contexts may load the same entry.
 * VELO_ERR_STATE when `frame` is not cached (the caller then reads the scan and stores it: what lru.h:48-58 does).
 * Like a context, a cache is not internally thread-safe: one call at a time (the reference's loop is single-threaded, velo.h:900). */
typedef struct velo_scan_cache velo_scan_cache;
int velo_cache_create(velo_scan_cache** out, int32_t device, int32_t capacity);
int velo_cache_destroy(velo_scan_cache* cache);
int velo_cache_store(velo_scan_cache* cache, int32_t frame, velo_ctx* ctx, int32_t of_target);
int velo_cache_load(velo_scan_cache* cache, int32_t frame, velo_ctx* ctx, int32_t as_target);
int velo_cache_contains(const velo_scan_cache* cache, int32_t frame);                      /* 1 / 0 */
/* frames from most to least recently used; returns the number of cached scans */
int velo_cache_frames(const velo_scan_cache* cache, int32_t* frames_out, int32_t capacity);
/* ring offsets / camera-frame points the context currently holds (for callers that segmented on the device) */
int velo_get_ring_offsets(velo_ctx* ctx, int32_t of_target, int32_t* out, int32_t capacity, int32_t* n_rings);
int velo_get_cloud(velo_ctx* ctx, int32_t of_target, float* xyz_out, int32_t capacity_points, int32_t* n_points);
/* Visual matches of both cameras, in the reference's iteration order cam-major (velo.h:622-627). n may be 0. */
int velo_set_visual(velo_ctx* ctx, const velo_match* matches, int32_t n);

/* --- the pieces (each also usable on its own; tests call them one by one) ---------------------------- */
/* One association round at pose x with the gate of outer iteration `iter` (1-based): velo.h:806-874.
 * Leaves the correspondence table on the device for velo_evaluate / velo_solve. */
int velo_associate(velo_ctx* ctx, const double x[6], int32_t iter, int32_t* n_valid);
/* Target-sharded pieces (tests drive them one by one; with a communicator in target-sharded mode velo_associate runs
 * them as  partial search -> all-to-all of the record slices -> merge  internally):
 *   velo_associate_partial: ALL queries against the local rings, one velo_partial per query left on the device;
 *   velo_get_partials:      copy them to the host;
 *   velo_merge_partials:    merge `world` host tables (each n_queries records) for THIS context's query share into its
 *                           correspondence table (rows A3-A6 finished by the query's owner). */
int velo_associate_partial(velo_ctx* ctx, const double x[6], int32_t iter);
int velo_get_partials(velo_ctx* ctx, velo_partial* out, int32_t capacity, int32_t* n_queries);
int velo_merge_partials(velo_ctx* ctx, const velo_partial* const* tables, int32_t world, int32_t* n_valid);
/* Copies the table back (one record per query, in query order sm-major / smi ascending). */
int velo_get_correspondences(velo_ctx* ctx, velo_corr* out, int32_t capacity, int32_t* n_queries);
/* Visual block selection + outlier gate G1 at pose x: velo.h:622-792. */
int velo_build_visual(velo_ctx* ctx, const double x[6], int32_t iter, int32_t* n_blocks);
int velo_get_good_matches(velo_ctx* ctx, velo_good_match* out, int32_t capacity, int32_t* n);
/* Robustified cost + normal equations of all current blocks at x: what one Ceres evaluation produces
 * (functors costfunctions.h:39-54,76-87,111-126,151-168,192-216; losses velo.h:688,714-717,748-751,
 * 781-784,887-890).  JtJ is the full symmetric 6x6 row-major, Jtr = J^T r, cost = sum 1/2 rho(s). */
int velo_evaluate(velo_ctx* ctx, const double x[6], double* cost, double JtJ[36], double Jtr[6]);
/* Optional row-level output for a ceres::CostFunction adaptor (seam 3): residuals[n_res] and row-major
 * jacobian[n_res*6], robustifier already applied.  Order: visual blocks, then ICP blocks (velo.h order). */
int velo_evaluate_rows(velo_ctx* ctx, const double x[6], double* residuals, double* jacobian,
                       int32_t capacity_rows, int32_t* n_rows);
/* Seam 2 (the functor concept, costfunctions.h:17-220: `template<class T> bool operator()(const T* x, T* residual) const`
 * instantiated by ceres::AutoDiffCostFunction<F, dim, 6>) by value: n functors at one pose in one launch.
 * kind = VELO_RESIDUAL_* or VELO_FUNCTOR_3DPD; c = the functor's constructor arguments widened to double, in the reference's
 * order: cost3D3D m(3) s(3) (costfunctions.h:62-74) | cost3D2D m(3) s(2) t(3) (94-110) | cost2D3D m(3) s(2) t(3) (134-150)
 * | cost2D2D m(2) s(2) t(3) (176-190) | cost3DPD point(3) normal(3) offset(3) (19-37).
 * residuals: n x 3 doubles (rows beyond the functor's dimension are 0), the RAW residuals -- no loss applied;
 * jacobians: n x 18 doubles, row-major dim x 6 per functor = what autodiff returns (may be NULL). */
typedef struct velo_functor {
    int32_t kind;
    int32_t reserved;
    double c[9];
} velo_functor;   /* 80 bytes */
int velo_evaluate_functors(velo_ctx* ctx, const velo_functor* functors, int32_t n, const double x[6],
                           double* residuals, double* jacobians);
/* residualStats (velo.h:921-1025) of the current blocks at x. */
int velo_residual_stats_at(velo_ctx* ctx, const double x[6], velo_residual_stats* out);
/* One ceres::Solve (velo.h:897-902) on the current blocks, x in/out. */
int velo_solve(velo_ctx* ctx, double x[6], velo_solve_summary* summary);

/* --- the path -------------------------------------------------------------------------------------- */
/* frameToFrame (velo.h:598-919): f2f_iterations x [visual blocks; icp_iterations x (associate; solve)].
 * x: in = initial guess, out = solution.  T: 4x4 row-major of the solution (util::pose_mat2vec, utility.h:67-82). */
int velo_frame_to_frame(velo_ctx* ctx, double x[6], double T[16], velo_summary* summary);
/* How the calls above ran.  A single-GPU, LiDAR-only call is enqueued as ONE chain of launches with one host synchronisation at its
 * end: the pose of round r+1's association and every solve's summary stay on the device, and the number of LM launches per solve is
 * the previous call's count plus a margin.  A solve that needs more is detected on the device (the next association finds its pose
 * record not ready); the call is then repeated with a host round trip per solve -- same kernels, bit-identical results.
 * calls = calls that went down the chain, misses = calls that had to be repeated.  VELO_CHAIN=0 (read by velo_create) switches the
 * chain off, VELO_CHAIN_MARGIN=n sets the margin (default 2). */
int velo_chain_stats(const velo_ctx* ctx, int32_t* calls, int32_t* misses);
/* Several independent scan pairs in flight, one context each (what run.fish:2 does with one process per sequence): equivalent
 * to n velo_frame_to_frame calls, results bit-identical.  Contexts on one device are advanced in lock-step groups (one sweep /
 * LM-step launch serves a whole group), otherwise one host thread per context.  Every context may appear once (VELO_ERR_INVALID
 * for duplicates or null entries); the first failing context's status is returned. */
int velo_frame_to_frame_batch(velo_ctx** ctxs, int32_t n, double* x /* n*6 */, double* T /* n*16 */,
                              velo_summary* summaries /* n or NULL */);
/* The same with the scans of every job handed over in the call (seam 1 takes scans_M / scans_S per call, velo.h:606-607): job i's
 * target and source go into context i exactly as velo_set_target / velo_set_source would put them, on the thread that then
 * drives that context's group -- a group starts registering as soon as ITS scans are indexed, no barrier across the batch.
 * targets / sources may be NULL (keep what the contexts hold).  n == 1 is velo_set_target + velo_set_source + velo_frame_to_frame
 * in one call (the single-pair path: one chain of launches, one host synchronisation).  A target loaded here for n >= 2 is indexed
 * with at most 2^24 cells instead of 2^25 (only a 2M-point map reaches either; any cell size gives the same results). */
#define VELO_SCAN_ON_DEVICE 1   /* xyz is a device pointer */
#define VELO_SCAN_SHARED 2      /* targets only: jobs with IDENTICAL descriptors carrying this flag share one device copy and one
                                 * index (scan-to-map: many scans against one map) -- built once, held by reference */
#define VELO_SCAN_PROMOTE 4     /* targets only: this job's target is the scan the context holds as its SOURCE (velo_source_to_target:
                                 * the previous frame of a drive becomes sd_prev, main.cpp:233,380 -- buffer swap + index build, no
                                 * upload, no second segmentation); xyz / ring_offsets of the descriptor are ignored.  The job's source
                                 * descriptor then brings the new frame. */
typedef struct velo_scan_ref {
    const float* xyz;
    int64_t stride_bytes;
    const int32_t* ring_offsets;
    int32_t n_rings;
    int32_t on_device;          /* VELO_SCAN_* flags */
} velo_scan_ref;   /* 32 bytes */
int velo_register_batch(velo_ctx** ctxs, int32_t n, const velo_scan_ref* targets, const velo_scan_ref* sources,
                        double* x /* n*6 */, double* T /* n*16 */, velo_summary* summaries /* n */);
/* The same with the jobs' visual matches handed over in the call as well (seam 1 takes `matches` / `keypoints` per call, velo.h:599-605):
 * job i's n_matches[i] records go into context i as velo_set_visual would put them (0: the context registers LiDAR-only), without a
 * host synchronisation per context -- the step of a tightly-coupled drive (BASELINE configs[2]) is ONE call. */
int velo_register_batch_visual(velo_ctx** ctxs, int32_t n, const velo_scan_ref* targets, const velo_scan_ref* sources,
                               const velo_match* const* matches /* n pointers */, const int32_t* n_matches /* n */,
                               double* x /* n*6 */, double* T /* n*16 */, velo_summary* summaries /* n or NULL */);
/* The drive loop of n sequences for n_frames frames in ONE call -- main.cpp:207-413 (the per-frame loop: load the frame, predict the motion
 * main.cpp:311-331, frameToFrame main.cpp:388-405, chain the pose main.cpp:408) for n sequences at a time; the reference runs sequences as
 * independent processes (run.fish:2).  Context i holds the frame its drive starts from as SOURCE (velo_set_source, or the last frame of an
 * earlier call).  For f = 0 .. n_frames-1: that frame is promoted to target (VELO_SCAN_PROMOTE), frames[f * n + i] is loaded as the new
 * source (with matches[f * n + i] / n_matches[f * n + i] when given), the pair is registered from x_guess[i], x_out / T_out / summaries
 * [f * n + i] receive what velo_register_batch returns for it, and poses[i] / x_guess[i] are handed over exactly as velo_pose_handoff
 * does it.  The lock-step groups walk their drives' frames independently (no barrier across the batch between frames); per pair the
 * results are bit-identical to the frame-by-frame calls.  poses: n 4x4 row-major in/out; x_guess: n*6 in/out (the start-up guess
 * {0,0,0,0,0,1}, main.cpp:170, for a drive's first pair).  flags: VELO_SEQ_LOCKSTEP -- the groups start every frame together (a barrier
 * between frames: the timing of one velo_register_batch call per frame, without the caller in the loop); 0 -- no barrier.
 * VELO_SEQ_ANNOUNCE -- `frames` holds ONE MORE frame per sequence, frames[n_frames * n + i], which is not registered but announced
 * (velo_hint_next_frame): its loads run behind the last frame's launches, and the next call, which must start with exactly that frame,
 * finds it in place.  A caller that gets its frames one at a time makes the step of a drive ONE call this way: n_frames = 1 with the frame
 * after it announced -- load, register, chain the pose, predict the motion (main.cpp:305-413 for n sequences). */
#define VELO_SEQ_LOCKSTEP 1
#define VELO_SEQ_ANNOUNCE 2
int velo_register_sequences(velo_ctx** ctxs, int32_t n, int32_t n_frames, const velo_scan_ref* frames /* n_frames*n (+ n with VELO_SEQ_ANNOUNCE) */,
                            const velo_match* const* matches /* n_frames*n or NULL */, const int32_t* n_matches /* n_frames*n or NULL */,
                            double* poses /* n*16 */, double* x_guess /* n*6 */, double* x_out /* n_frames*n*6 */, double* T_out /* n_frames*n*16 or NULL */,
                            velo_summary* summaries /* n_frames*n or NULL */, int32_t flags);
/* main.cpp:216,349 load a scan per frame.  A caller that knows which HOST cloud it will hand over as the context's NEXT source announces it
 * here; the library uploads it on a copy stream of its own while the current registration's launches run (the upload is issued by the
 * thread that is about to wait for them), and the very next velo_set_source / batch job / sequence frame that names the same pointer and
 * size takes the uploaded copy instead of uploading again.  A different source drops the hint; device clouds ignore it; results never
 * change.  velo_register_sequences does this by itself for the frames it is given. */
int velo_hint_next_source(velo_ctx* ctx, const velo_scan_ref* next);
/* The whole next STEP of a drive announced one call ahead (main.cpp:216,233,349,380): the next job on this context will promote its source
 * (VELO_SCAN_PROMOTE) and bring `next` as the new source.  A chained registration (velo_register_batch[_visual], velo_frame_to_frame through
 * the batch entries) then enqueues that promotion, the ingest and the index build BEHIND its own launches before its thread waits for them:
 * they run while the host reads the results and hands the pose over, and the next call finds the frame in place.  Until then the context is
 * one frame ahead -- any other job on it returns VELO_ERR_STATE.  A call that had to be repeated host-driven first gets its own pair back
 * (the old target's cloud is kept for that).  Results never change.  `next` itself is copied; the cloud and the ring table it names must
 * stay valid and unchanged until the call that brings the frame has returned (they are read during the announcing call's registration and
 * compared by the next one).  A scan that cannot be promoted or an announcement the loaders would refuse is not loaded ahead: the call that
 * brings it reports the error, as without the announcement. */
int velo_hint_next_frame(velo_ctx* ctx, const velo_scan_ref* next);

/* --- pose helpers (utility.h:67-96; note the reference's swapped names) ---------------- */
int velo_pose_vec_to_mat(const double x[6], double T[16]);  /* util::pose_mat2vec */
int velo_pose_mat_to_vec(const double T[16], double x[6]);  /* util::pose_vec2mat */
/* The pose hand-off of the drive loop (main.cpp:311-331,408) for n sequences at once, host arithmetic only: poses[i] (row-major 4x4,
 * in/out) <- poses[i] * dpose[i] (ceres_poses_mat[frame] = ceres_poses_mat[frame-1] * dpose), and x_next[i] <- the 6-vector of
 * poses_old[i]^-1 * poses_new[i] -- the constant-velocity guess the next frame's frameToFrame starts from.  x_next may be NULL. */
int velo_pose_handoff(int32_t n, double* poses /* n*16 */, const double* dpose /* n*16 */, double* x_next /* n*6 or NULL */);

/* --- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------------- */
/* Query-sharded mode: every rank holds the whole target and a contiguous 1/world share of the query list;
 * each evaluation all-reduces the 28-double block (21 JtJ + 6 Jtr + cost) so all ranks take the same LM
 * decisions.  The 128-byte id comes from rank 0 and is distributed by the host program. */
int velo_comm_unique_id(char id[128]);
int velo_comm_init(velo_ctx* ctx, const char id[128], int32_t rank, int32_t world);
int velo_comm_destroy(velo_ctx* ctx);
/* The same mode without a collective library call per evaluation ("peer-mapped one-shot all-reduce"): every
 * rank owns a small slab in device memory; velo_comm_peer_export creates it and returns its 64-byte IPC handle
 * (hipIpcMemHandle_t); the host program gathers the handles of all ranks (rank order) and hands them to velo_comm_peer_attach,
 * which maps the peers' slabs.  Each LM step then writes its 28 doubles straight into every peer's slab and adds the world's
 * blocks in rank order inside the step kernel: deterministic, identical on all ranks, a few microseconds over xGMI instead of a
 * collective launch.  world <= 8.  Ranks may be processes on different GPUs of one node or -- for tests -- on the same GPU.
 * A peer that never arrives makes the wait time out (5 s): the call in flight (velo_frame_to_frame, velo_solve, velo_evaluate,
 * velo_associate in target-sharded mode) returns VELO_ERR_COMM, and the communicator must be exported and attached again on every rank
 * (the slabs' sequence numbers are out of step).  Every velo_comm_peer_export hands out a NEW, cleared slab (the previous one is
 * retired, not re-used: a slow peer's timed-out call may still be storing into it): every rank must have exported before any rank
 * attaches -- gathering the handles is that barrier -- and no barrier is needed between attach and the first call.
 * A chained call enqueues a predicted number of LM launches per solve, and over peers that number must be the same on every rank: the
 * ranks agree on it at the start of every chained call (the maximum over the ranks' own predictions, exchanged through the slabs),
 * so contexts with different call histories or VELO_CHAIN_MARGIN settings may share a communicator. */
int velo_comm_peer_export(velo_ctx* ctx, char handle[64]);
int velo_comm_peer_attach(velo_ctx* ctx, const char* handles /* world * 64 bytes, rank order */, int32_t rank, int32_t world);
/* Target-sharded mode over the same peers (instead of RCCL send/recv): the receive area for the per-query records of up to
 * max_queries queries (2 x (max_queries + 8 world) x 80 bytes), exported and attached like the slab, after velo_comm_peer_attach.
 * With it, velo_comm_set_target_sharded(ctx, 1) exchanges the records by direct stores into the owners' areas. */
int velo_comm_peer_export_records(velo_ctx* ctx, int32_t max_queries, char handle[64]);
int velo_comm_peer_attach_records(velo_ctx* ctx, const char* handles /* world * 64 bytes, rank order */, int32_t max_queries);
/* What is attached: kind 0 = nothing, 1 = RCCL communicator (world read back with ncclCommCount), 2 = peer slabs. */
int velo_comm_info(const velo_ctx* ctx, int32_t* kind, int32_t* rank, int32_t* world);
/* enable != 0: the communicator's ranks hold disjoint ring blocks of the target (velo_set_target_part) instead of
 * replicas; association then exchanges per-query top-2 records (all-to-all) before the query-sharded evaluation. */
int velo_comm_set_target_sharded(velo_ctx* ctx, int enable);
/* Restrict this context's queries to share `rank` of `world` WITHOUT a communicator (tests / replicas). */
int velo_set_query_shard(velo_ctx* ctx, int32_t rank, int32_t world);

/* Blocks until everything queued on the context's stream is done. */
int velo_synchronize(velo_ctx* ctx);

/* --- "next" row 3 of SURVEY.md 8(f): the producer of the 3-D keypoints rows R2-R4 consume ---------------- */
/* projectLidarToCamera (velo.h:329-374) for one camera, on the rings this context holds as source (of_target == 0) or target:
 * every point is shifted by cam_t (float, = cam_trans[cam]), projected c = (x/z, y/z) in float, kept when z > 0 and c lies in
 * [bounds[0], bounds[1]) x [bounds[2], bounds[3]) (= min_x, max_x, min_y, max_y of kitti.h:85-97, doubles), and pushed on the
 * ring's occlusion stack: entries behind it in x and farther in z are popped (velo.h:351-358), the point is dropped when the
 * stack top is right of it and nearer (velo.h:360-365).  Rings are independent; each is one sequential pass like the reference.
 * The surviving (`projection`, `scans_valid`) lists stay on the device for velo_depth_association. */
int velo_project_lidar(velo_ctx* ctx, int32_t of_target, const float cam_t[3], const double bounds[4], int32_t* n_valid_total);
/* copies the lists back: proj_xy [n][2], points_xyz [n][3] (the un-shifted points, velo.h:368), ring_offsets [n_rings + 1] */
int velo_get_projection(velo_ctx* ctx, float* proj_xy, float* points_xyz, int32_t capacity_points, int32_t* ring_offsets,
                        int32_t capacity_offsets, int32_t* n_rings);
/* featureDepthAssociation (velo.h:376-497) against the last velo_project_lidar: for every keypoint (canonical coordinates,
 * [n][2] floats) the first ring s whose bracketing segment and that of ring s-1 straddle the keypoint in y and are both
 * narrower than depth_assoc_thresh (kitti.h:28) gives a 3-D point by the reference's float bilinear interpolation.
 * has_depth[k] = -1 or the index into kp_with_depth (appended in keypoint order).  The reference's unqualified abs() on the
 * float widths (velo.h:416,419) is restated as fabs, like the outlier gate. */
int velo_depth_association(velo_ctx* ctx, const float* keypoints_xy, int32_t n_keypoints, double depth_assoc_thresh,
                           float* kp_with_depth_xyz, int32_t capacity_points, int32_t* has_depth, int32_t* n_with_depth);

/* --- "next" row 4: every landmark of a frame triangulated in one call (main.cpp:661-671 loops triangulatePoint over ids) ---
 * Landmark l owns obs[obs_offsets[l] .. obs_offsets[l+1]).  Residual blocks are added like velo.h:1049-1122: all 3-D
 * observations in the order given, then all 2-D observations in the order given (pass them cam-major, frame ascending --
 * the iteration order of the reference's per-camera std::map -- to reproduce its summation order).  Start value: the
 * current point when initial_guess[l] != 0, else (0, 0, 10) and, if there is a 3-D observation, a first solve on the first
 * 3-D block alone (velo.h:1080-1083).  The solver is the context's Ceres-default LM (velo_params), 3 unknowns.
 * camera_poses: [n_frames][6] (angle-axis, translation) as in ceres_poses_vec; cam_trans: [n_cams][3] floats.
 * points_xyz [n_landmarks][3] is read (initial guesses) and written (float, like pcl::PointXYZ); results may be NULL. */
int velo_triangulate_points(velo_ctx* ctx, const double* camera_poses, int32_t n_frames, const float* cam_trans, int32_t n_cams,
                            const velo_tri_obs* obs, const int32_t* obs_offsets, int32_t n_landmarks, float* points_xyz,
                            const uint8_t* initial_guess, velo_tri_result* results);

#ifdef __cplusplus
}
#endif
#endif /* VELO_HIP_H_ */
