// velo_hip.hip -- host side of the C-ABI declared in include/velo_hip.h (gfx950 only; no CPU fallback).
//
// One context = one HIP stream + device-resident target index, source cloud, correspondence table and LM
// state.  frame_to_frame (reference velo.h:598-919) runs as
//     for iter:  visual gate kernel;  for icp_iter:  association kernel;  LM solve = chunks of
//     [eval sweep -> (RCCL all-reduce) -> lm_step] launches that early-exit on the device-side `done` flag,
// with one small D2H status copy per chunk (the only host synchronisation inside a solve).
// velo_frame_to_frame_batch advances several contexts in lock-step groups with shared LM launches (f2f_batch_lockstep);
// problems whose sweep is a few workgroups run a whole solve in one launch (lm_solve_small_kernel).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <list>
#include <memory>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <chrono>

#include "../../include/velo_hip.h"
#include "velo_kernels.h"
#include "velo_depth_kernels.h"
#include "velo_tri_kernels.h"

using namespace velo;

// velo_lm_ag.hip: the all-gather solve's launcher (a translation unit of its own, see velo_lm_ag_kernels.h)
extern "C" int velo_launch_lm_solve_ag(int nb_max, int n, void* stream, const void* lm_params, size_t lm_params_bytes, const void* pack, size_t pack_bytes,
                                       void* ctl, int kmax, size_t half, void* a, void* b);

namespace {

thread_local std::string g_err;
std::string g_err_shared;   // last error of any thread (read by velo_last_error when the caller's own is empty)
std::mutex g_err_mutex;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    { std::lock_guard<std::mutex> lk(g_err_mutex); g_err_shared = buf; }
    return code;
}

#ifdef VELO_DIAGNOSTICS
// dev aid (VELO_API_TRACE=<us>, diagnostics build): every runtime call that keeps its caller longer than that is reported with its text --
// how the copies that block for milliseconds were found.  Synchronisations are expected to wait and are not reported.
static const double g_api_trace_us = getenv("VELO_API_TRACE") ? atof(getenv("VELO_API_TRACE")) : 0.0;
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        const auto t__ = g_api_trace_us > 0.0 ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point(); \
        hipError_t e__ = (expr);                                                                        \
        if (g_api_trace_us > 0.0) {                                                                     \
            const double us__ = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t__).count(); \
            if (us__ > g_api_trace_us && !strstr(#expr, "Synchronize")) fprintf(stderr, "[velo api] %.0f us in %s (line %d)\n", us__, #expr, __LINE__); \
        }                                                                                               \
        if (e__ != hipSuccess) return fail(VELO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#else
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) return fail(VELO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#endif
#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess) return fail(VELO_ERR_COMM, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r__), __FILE__, __LINE__); \
    } while (0)
#define VELO_TRY(expr)           \
    do {                         \
        int s__ = (expr);        \
        if (s__ != VELO_OK) return s__; \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    ~DevBuf() { release(); }             // every buffer a context owns goes with it (velo_destroy -> delete)
    int reserve(size_t n) {
        if (n <= cap) return VELO_OK;
        const size_t cap_before = cap; (void)cap_before;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 8 + 64;
#ifdef VELO_DIAGNOSTICS
        // A/B (VELO_ALLOC_2MB=1): buffers of 256 KB and more padded to whole 2 MB -- does the page-table fragment size matter to the gathers?
        static const bool pad2m = getenv("VELO_ALLOC_2MB") && atoi(getenv("VELO_ALLOC_2MB")) != 0;
        if (pad2m && want * sizeof(T) >= (256u << 10)) want = ((want * sizeof(T) + (2u << 20) - 1) / (2u << 20)) * (2u << 20) / sizeof(T);
#endif
#ifdef VELO_DIAGNOSTICS
        static const bool alloc_trace = getenv("VELO_ALLOC_TRACE") != nullptr;   // dev aid: a (re)allocation synchronises the device -- which buffers still grow in a warm loop?
        if (alloc_trace) fprintf(stderr, "[velo alloc] device buffer of %zu-byte elements: %zu -> %zu elements\n", sizeof(T), cap_before, want);
#endif
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e != hipSuccess) return fail(VELO_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return VELO_OK;
    }
    // for buffers whose size moves from frame to frame (the index table follows the scan's bounding box): when it has to grow, grow by `extra`
    // elements more -- a reallocation synchronises the device, and the queues stall for 6-7 ms one step later (measured: tools/step_times.py)
    int reserve_roomy(size_t n, size_t extra) { return n <= cap ? VELO_OK : reserve(n + extra); }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Grid {
    GridDesc d{};
    double gate = 0.0;          // squared-distance gate the cell size was derived from
    double h = 0.0;             // cell size
    DevBuf<int> cell_start;     // 3 + ncells + 1: the table starts at element 3, so that table + 1 -- what the one-pass scan and the scatter work on -- is 16-byte aligned
    int* table() const { return cell_start.p + 3; }
    DevBuf<float4> sorted;      // cell-sorted copy {x,y,z,bits(gidx)} (+ kGridPad sentinels)
    DevBuf<int> sring;
    // compressed table (VELO_GRID_COMPRESS=1 in the diagnostics build, see GridView and build_grid): occupancy bits + occupied cells before every word;
    // cell_start then holds the start of every OCCUPIED cell (at most n + 1 entries)
    DevBuf<unsigned long long> wmask;
    DevBuf<int> wprefix;
    int wpr = 0;                // words per grid row; 0 = dense table
    size_t table_len() const { return wpr > 0 ? (size_t)n_points_cap + 1 : (size_t)d.ncells + 1; }   // entries of table() in use
    size_t n_words() const { return (size_t)wpr * (size_t)d.ny * (size_t)d.nz; }
    int n_points_cap = 0;       // compressed: the target's point count when the table was built
    bool built = false;
    void view(GridView* V) const {
        V->d = d; V->cell_start = table(); V->sorted = sorted.p; V->sring = sring.p;
        V->wmask = wpr > 0 ? wmask.p : nullptr; V->wprefix = wpr > 0 ? wprefix.p : nullptr; V->wpr = wpr;
    }
};

struct HostStatus {   // pinned; one D2H copy per LM chunk
    LMState s;
};

void default_params(velo_params* p) {
    std::memset(p, 0, sizeof(*p));
    p->icp_skip = 200; p->f2f_iterations = 2; p->icp_iterations = 3;          // kitti.h:8-10
    p->enable_icp = 1; p->enable_2d2d = 1; p->enable_3d2d = 1;                  // main.cpp:43-45,404
    p->max_num_iterations = 50; p->max_consecutive_invalid_steps = 5;
    p->weight_3D2D = 10; p->weight_2D2D = 500; p->weight_3DPD = 1;              // kitti.h:20-22
    p->loss_thresh_3D2D = 0.01; p->loss_thresh_2D2D = 0.00002;                  // kitti.h:23-24
    p->loss_thresh_3DPD = 0.1; p->loss_thresh_3D3D = 0.04;                      // kitti.h:25-26
    p->outlier_reject = 5.0; p->correspondence_thresh_icp = 0.5;                // kitti.h:30-31
    p->icp_norm_condition = 1e-5;                                               // kitti.h:32
    p->function_tolerance = 1e-6; p->gradient_tolerance = 1e-10; p->parameter_tolerance = 1e-8;
    p->initial_trust_region_radius = 1e4; p->max_trust_region_radius = 1e16;
    p->min_trust_region_radius = 1e-32; p->min_relative_decrease = 1e-3;
    p->min_lm_diagonal = 1e-6; p->max_lm_diagonal = 1e32;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Environment surface.  The PRODUCT library reads five documented, result-preserving knobs (include/velo_hip.h, "environment"):
// VELO_CHAIN, VELO_CHAIN_MARGIN, VELO_BATCH_GROUPS, VELO_BATCH_LOCKSTEP, VELO_SPIN.  Every other switch -- kernel variants, grid
// shapes, A/B paths, diagnostics -- exists only in the tools' build (-DVELO_DIAGNOSTICS, libvelo_hip_diag.so): a drop-in
// frameToFrame whose kernel selection followed leaked environment variables would not be a product surface (the reference's knobs
// are compile-time constants, kitti.h:3-35).  The parity tests that sweep variants load the diagnostics library.
inline const char* dev_env(const char* name) {
#ifdef VELO_DIAGNOSTICS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace

// What set_target builds: the target cloud and its search index.  Held through a shared_ptr: contexts that register different
// scans against the same map (scan-to-map batches) share ONE copy -- 110 MB for a 2M-point map instead of one per context, one index
// build instead of one per context.  Read-only once built; a context that loads a new target while others still hold this one
// starts a fresh TargetData instead of overwriting it.
struct TargetData {
    int n_tgt = 0, n_tgt_rings = 0;
    int tgt_first_ring = 0, tgt_first_point = 0;   // target-sharded mode: global ids of the first local ring / point
    DevBuf<float4> tgt;
    DevBuf<float4> tgt_pad;              // ring-major copy with a wrap-around sentinel on either side of every ring (pad_rings_kernel)
    DevBuf<int> tgt_off, tgt_ring_of, tgt_cell_of;
    std::vector<int> h_tgt_off;
    std::vector<Grid> grids;             // one per distinct gate among iter = 1..f2f_iterations
    float bbox[6] = {0, 0, 0, 0, 0, 0};
    DevBuf<unsigned long long> dimg;     // direction image (kDimgW x kDimgH nearest-point keys): where the seeds of a round without good predecessors come from
    bool dimg_built = false;
};

struct velo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    velo_params P;
    int timing = 0;                      // 0 off, 1 association launches (velo_summary::assoc_kernel_ms), 2 every instrumented launch by kernel name
    int assoc_variant = -1;              // VELO_ASSOC_VARIANT: -1 = default (tube kernel 5); 0 = per-lane reference kernel; 1/2/4/8 = waves per group
                                         // of the box walk; 5 = tube kernel
    int cluster_w = 6;                   // cluster radius of the box kernels, in cells of the default grid (VELO_CLUSTER_W)
    bool cluster_w_set = false;          // the tube kernel keeps one cluster per group unless VELO_CLUSTER_W is given
    int persistent_wgs = 2048;           // workgroups of the persistent association kernel (VELO_PERSISTENT_WGS)
    int xcd_map = 0;                     // box kernels: XCD-contiguous group mapping (VELO_XCD_MAP=1): measured slower
    int tube_map = -1;                   // tube kernel (VELO_TUBE_MAP): -1 = groups in ring order (default); 1 = XCD k works on the k-th eighth of
                                         // every ring (a wedge of the scene): L2 hit rate 57 % -> 73 %, yet 8 % SLOWER (the kernel is bound by VALU
                                         // issue and per-group latency chains, not by L2 misses); 0 = ring order through the same table
                                         // (dense bottom rings all land on one XCD); round-robin placement balances better
    int debug_skip = 0;                  // timing experiments only (VELO_DEBUG_SKIP): results are wrong when non-zero

    // target (frame2): cloud + search index, shareable between contexts (velo_share_target: many scans against one map)
    std::shared_ptr<TargetData> T = std::make_shared<TargetData>();
    DevBuf<int> vis_counts;                       // chain mode: [f2f iteration][blocks, residuals] selected by the device-side gate
    DevBuf<int> scan_tiles, cursor, scan_total;   // scratch of an index build / of the segmenter's scans
    struct { const char* dsrc = nullptr; int64_t stride = 0; bool on = false; } src_raw;   // set_source: the records the fused ingest launch still has to read (source_finalize)
    bool src_bbox_valid = false;         // h_int[16..21] hold the bounding-box keys of the source cloud (source_ingest; on the host once src_bbox_ev has passed)
    hipEvent_t src_bbox_ev = nullptr;    // recorded behind the keys' copy: a promotion waits on it (a no-op after any completed call) before it trusts them
    bool target_early = false;           // promote_begin already sized and enqueued the index (box known): target_finalize_end has nothing to wait for
    int lb_zeroed = 0;                            // status words of the one-pass scan that target_ingest_kernel cleared for the next build (0: build_grid clears them)
    bool batch_load = false;                      // set while velo_register_batch loads this context's scans for a batch of two or more (see build_grid)
    DevBuf<unsigned long long> lb_status;         // one-pass scan: tile status words + ticket
    DevBuf<unsigned> bbox_keys;
    bool have_target = false;

    // source (frame1)
    int n_src = 0, n_src_rings = 0, n_q = 0;
    DevBuf<float4> src;
    DevBuf<int> src_off, q_off, q_src;
    std::vector<int> h_src_off, h_q_off;
    int src_skip = 0;                    // icp_skip the query list was built with
    DevBuf<float4> qpts_buf;             // query points by query index when icp_skip > 1 (with icp_skip == 1 the source cloud is the list)
    const float4* qpts = nullptr;
    DevBuf<float4> prev_a;               // tube kernel warm start: last round's two winners per query with their coordinates (index -1 = none; [0, nq) best, [nq, 2 nq) second),
    DevBuf<int2> prev_r;                 // and their rings; reset with every new source / target
    bool prev_ready = false;             // the seed arrays hold n_q initialised entries for the current source and target
    bool prev_filled = false;            // ... they hold "no previous winner" everywhere already (written by the launch that loaded the source: advance_ingest_kernel)
    int prev_filled_nq = -1;             // ... for this many queries
    int warm_start = 1;                  // VELO_WARM_START=0 turns the seeds off (A/B; results are identical either way)
    int assoc_lds_pad = 0;               // bytes of unused dynamic LDS per association workgroup -- caps the association kernel's workgroups per
                                         // CU so that LM workgroups of other pairs in flight find room at once.  Set by the lock-step batch
                                         // driver when several groups share the chip (kAssocPadShared); VELO_ASSOC_LDS_PAD (diagnostics build) fixes it
    bool assoc_lds_pad_fixed = false;
    int lm_persist = 0;                  // VELO_LM_PERSIST=1 (diagnostics build): lock-step groups in chain mode run a whole solve as ONE launch
                                         // (lm_solve_persist_batch_kernel).  Exact (bit-identical), measured, NOT a gain: an iteration inside the launch
                                         // takes 28 us alone (six agent-scope hand-overs between workgroups) against 19 us for a launch; 8 pairs in
                                         // flight 2,742 vs 3,376 pairs/s -- the waiting workgroups hold registers the association kernels want
    int lm_persist_wgs = 0;              // workgroups per context of that launch (0 = one per virtual block); VELO_LM_PERSIST_WGS (diagnostics build)
    DevBuf<SolveCtl> solve_ctl;          // its per-context control blocks (owned by the first context of a group; zero between launches)
    DevBuf<AgCtl> ag_ctl;                // all-gather solve (lm_solve_ag_batch_kernel, lm_persist == 2): epoch, abort word and flags per context of a group
    int lm_lean = -1;                    // lean fused LM kernel in lock-step groups: -1 = when several groups share the chip; VELO_LM_LEAN (diagnostics build) forces 0 / 1
    // chain mode: a whole frame_to_frame as ONE chain of launches (pose scalars of the next round and the solve summaries stay on the device)
    int patch_order = 1;                 // query list in patch order (VELO_PATCH_ORDER=0: the reference's ring order)
    bool q_patch = false;                // the current list is in patch order
    int patch_rings = kPatchRingsDefault, patch_len = kPatchLenDefault;   // VELO_PATCH_SHAPE=rings,points
    bool ring_order_forced = false;      // this context exchanges per-query records with others (target-sharded workflow): the list stays in the reference's order
    int direct_max = 12288;              // sparse rounds (icp_skip >= direct_skip) of at most this many queries search one wave per query
    int direct_skip = 4;                 // (VELO_ASSOC_DIRECT_MAX, 0 = never; VELO_ASSOC_DIRECT_SKIP)
    // Density-shrunk grids (scan-to-map), single calls: a group whose phase-1 boxes span more than dense_rows grid rows -- queries strung
    // along a wall that thirty scans have sampled: 12,000-17,000 staged candidates -- is searched query by query as a whole, provided at
    // most dense_far of its members have a bound beyond four cells (see assoc_search_v5_body).  Measured on the 2M-point map, us per round
    // of a call: 450 / 402 / 305 / 231 / 230 / 137 -> 435 / 396 / 215 / 149 / 165 / 102, single registration 2.56 -> 2.35 ms.  A launch
    // ends with its slowest group, and these are the slowest; with 8 pairs in flight other groups' kernels fill that tail anyway and the
    // step does not move (1,242 vs 1,235 pairs/s), so lock-step batches keep the tile path (dense_batch).  VELO_DENSE_ROWS / VELO_DENSE_FAR /
    // VELO_DENSE_BATCH (diagnostics build) override.
    int dense_rows = 384, dense_far = 2, dense_batch = 0;
    int asker_queue = 1;                 // shrunk grid: asking queries go to assoc_asker_kernel (VELO_ASKER_QUEUE=0: searched inside their group's workgroup)
    int ask_map = 0;                     // VELO_ASK_MAP: which list entries a wave of the asker kernel takes (0 strided, 1 contiguous + XCD-chunked)
    DevBuf<int> ask_count, ask_list;
    DevBuf<unsigned long long> ask_keys;
    DevBuf<int2> ask_rings;
    int ask_idx = 0;
    bool ask_clean[2] = {false, false};
    bool lm_trace_vis_off = false;       // VELO_LM_MERGED_VIS=0: calls with visual blocks keep sweep + visual sweep + step as three launches (A/B)
    bool want_stats = false;             // velo_set_residual_stats
    DevBuf<double> stat_vals, stat_part;
    DevBuf<signed char> stat_types;
    DevBuf<int> stat_hist;
    DevBuf<StatWork> stat_work;
    DevBuf<velo_residual_stats> stat_out;
    int assoc_lane = 0;                  // VELO_ASSOC_LANE=1: rounds that start from seeds use the lane kernel (A/B; slower, see assoc_lane_body)
    int seed_rounds = 0;                 // association rounds since the seeds were last cleared
    int xcd_chunks = 0;                  // VELO_XCD_CHUNKS=1 (diagnostics build): XCD k searches the k-th eighth of the query list (see assoc_search_v5_body)
    int cu_mask_mode = 0;                // VELO_CU_MASK=1|2 (diagnostics build): this context's stream is confined to a quarter of the CUs (1: bits 64 q .. 64 q + 63,
                                         // 2: the bits i with (i % 8) / 2 == q), q = (creation order / 2) % 4 -- the experiment of giving every lock-step group its own CUs
    int dimg_seeds = 0;                  // VELO_DIMG_SEEDS=1 (diagnostics build): seeds of iteration-1 rounds from the target's direction image (seed_kernel).
                                         // Measured, exact, NOT a gain: C2 rounds 95 / 80 / 78 -> 90 / 86 / 85 us (the cold round's cost is the true
                                         // second-ring distance of the far queries, not poor seeds; the seed launch costs 7 us), 8 pairs in flight
                                         // 3,305 -> 3,354 pairs/s (noise); 2M-point map 1,233 -> 1,050 (0.35-degree buckets are 4 cells wide there)
    int chain_calls = 0;                 // calls that went down the chain
    int chain_margin = 2;                // LM launches enqueued per solve beyond the previous call's count (VELO_CHAIN_MARGIN)
    int chain_misses = 0;                // calls whose chain was too short and were repeated by the host-driven path
    int chain = 1;                       // VELO_CHAIN=0: host round trip after every solve (A/B; identical results)
    DevBuf<PoseRecord> pose_rec;
    DevBuf<SolveLog> solve_log;
    DevBuf<int> chain_fail;
    SolveLog* h_log = nullptr;           // pinned: VELO_MAX_SOLVES logs + the failure flag behind them
    int lm_fused = 1;                    // VELO_LM_FUSED=0: the lock-step batch driver launches sweep and LM step separately (A/B, identical results)
    int lm_vis_merged = 1;               // chained batch solves with visual blocks run them INSIDE the fused sweep + step launch (extra workgroups, eval_step_batch_(lean_)vis_kernel);
                                         // VELO_LM_VIS_MERGED=0: a launch of their own ahead of it (A/B, identical results)
    int lm_iter = 0;                     // VELO_LM_ITER=1: chained batch solves launch the lean one-launch iteration (every workgroup advances the state itself) instead of the fused sweep + step (A/B, identical results; measured slower: 3,204 vs 3,418 pairs/s)
    int lm_merged = 1;                   // VELO_LM_MERGED=0: sweep and LM step as two launches per iteration also where one would do (A/B, identical results)
    int small_solve = 1;                 // VELO_SMALL_SOLVE=0: small problems go through the launch-per-iteration path too (A/B, identical results)
    int asker_rows = -1;                 // tube kernel (VELO_ASKER_ROWS): phase 2 goes query by query when the asking queries' boxes have more
                                         // rows than this in total.  -1 = by target density: never on a regular scan (120k points: the tile pass
                                         // is 62 vs 105-115 us), always when the grid had to be density-shrunk (2M-point map: 244 vs 420 us)
    DevBuf<FunctorRec> fn_in;            // velo_evaluate_functors: the records of one call
    DevBuf<int> group_perm;              // workgroup -> 64-query group, XCD-aware (see build_group_perm)
    int perm_qb = -1, perm_qe = -1, perm_nq = -1, perm_mode = -1;
    bool have_source = false;

    DevBuf<char> staging;                // raw host clouds land here before packing
    // velo_hint_next_source: the NEXT frame's raw host cloud is uploaded on a copy stream of its own while this frame's chain of launches runs
    // (main.cpp:216 loads a scan per frame): two landing buffers, so the upload for frame k + 2 never touches what frame k + 1's ingest reads
    struct Prefetch {
        const void* host = nullptr; size_t bytes = 0; bool hinted = false, ready = false; int buf = 0;
        DevBuf<char> land[2]; hipStream_t stream = nullptr; hipEvent_t ev = nullptr;
        char* pin[2] = {nullptr, nullptr}; size_t pin_cap[2] = {0, 0}; bool in_pin = false;   // the announced cloud in page-locked memory of the library's own (see prefetch_issue)
    } pf;
    // velo_hint_next_frame: the NEXT frame of a drive -- promote the scan held as source, load the announced scan as the new source, build the
    // index -- is enqueued BEHIND the current registration's chain of launches, before the calling thread waits for it: the loads of frame
    // k + 1 run while the host wakes up, reads frame k's results and hands the pose over (the step boundary, where every queue used to
    // drain).  The old target's cloud is kept until the call is known to be good: a call that has to be repeated host-driven gets its pair back.
    struct NextFrame {
        enum State { NONE = 0, LOADED = 2, CONSUMED = 3 };
        int state = NONE;                   // of the frame loaded ahead: LOADED until the job that brings it arrives, CONSUMED while that job runs
        velo_scan_ref ref{};                // ... and its descriptor
        bool hint_valid = false;            // an announcement waiting for the end of the current call's enqueue (it may arrive while `state` is LOADED:
        velo_scan_ref hint{};               //  the caller announces frame k + 1 before the job of frame k, loaded ahead one call ago, has been handed over)
        DevBuf<float4> undo_cloud; std::vector<int> undo_off; int undo_n = 0, undo_rings = 0;   // the old target's cloud: its BUFFER, rotated out (no copy)
        DevBuf<unsigned> keys; int parity = 0;   // group-batched loads (AdvJob): the source's bounding-box keys in two slots, used alternately
        hipEvent_t call_done = nullptr;     // behind the call's last read-back copy: what the calling thread waits for when more has been enqueued behind it
    } nf;
    double last_chain_us = 0.0;          // the previous chained lock-step call this context led: enqueue -> results in (sizes the stagger of the groups' starts)
    AdvJob* adv = nullptr;               // set while a group's loads are being COLLECTED (preload_group): target_ingest / build_grid / source_ingest fill it instead of launching
    DevBuf<int> seg_flag, seg_excl, seg_ring, seg_off;   // device-side ring segmentation (velo_set_scan_velodyne)

    // correspondence table
    DevBuf<float4> cp, cn, cv0, aux1;
    DevBuf<int4> aux0;
    DevBuf<int> n_valid;                 // two counters used alternately: the association kernel of one round clears the counter of
    int nv_idx = 0;                      // the next, so no fill launch (and no 18 us launch gap behind it) per round
    bool nv_clean[2] = {false, false};   // counter is zero on the stream's timeline
    DevBuf<unsigned long long> dbg, wg_times;
    int wg_times_n = 0;
    DevBuf<AssocItem> items;             // work queue of the pipelined association
    DevBuf<int> item_counters;
    DevBuf<float4> qpos;
    DevBuf<PartialRec> partials_rec, partials_all;   // target-sharded mode: my records for all queries / every rank's records for my queries
    bool have_partials = false;
    int last_partial_iter = 1;           // gate of the last partial association (the merge needs its key_inf)
    bool target_sharded = false;
    bool have_corr = false;
    int last_n_valid = 0;

    // visual
    int n_matches = 0;
    DevBuf<VisualMatch> vm;
    DevBuf<unsigned char> vflags;
    std::vector<velo_match> h_matches;
    std::vector<unsigned char> h_vflags;
    bool vflags_valid = false;

    // LM
    DevBuf<LMState> state;
    DevBuf<LMEvalPoint> eval_pt;         // where the next sweep evaluates: written by lm_begin / the LM step, read by the sweep workgroups
    DevBuf<double> partials, reduced, xdev;
    DevBuf<int> ticket;
    DevBuf<int> batch_tickets;                    // fused sweep + step of a lock-step group: one ticket counter per context (0 at launch boundaries)
    DevBuf<unsigned long long> lm_trace;  // diagnostics build, VELO_LM_TRACE=1: stage stamps of the LM chain (tools/lm_trace.py)
    bool lm_trace_on = false;
    int lm_trace_idx = 0;                 // launches of the current solve so far
    // captured LM chunks (single GPU): key = iterations per chunk; rebuilt when anything baked into the nodes changes
    hipGraphExec_t chunk_graph[2] = {nullptr, nullptr};
    int chunk_graph_iters[2] = {0, 0};
    std::vector<unsigned char> chunk_graph_sig[2];   // bytes of everything baked into the nodes
    bool use_graphs = false;             // LM chunks as hipGraphs (VELO_GRAPHS=1): measured no gain, replay overhead ~ launches saved
    int pred_evals[VELO_MAX_SOLVES];     // evaluations each solve of the previous frame_to_frame needed (chunk sizing)
    int eval_hist[VELO_MAX_SOLVES][4];   // ... and of the last four calls: how far a solve's count moves decides the chain's margin
    int eval_hist_n[VELO_MAX_SOLVES];
    bool chain_margin_fixed = false;     // VELO_CHAIN_MARGIN given: that margin, always
    HostStatus* h_status = nullptr;      // pinned
    double* h_x = nullptr;               // pinned, 8 doubles
    int* h_int = nullptr;                // pinned scratch
    // pinned staging for the small host tables a load sends to the device (source ring offsets, query offsets): the copy is asynchronous
    // and the slot's event says when the host may write the slot again -- no stream synchronisation at the end of a load
    struct PinSlot { int* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
    PinSlot pin[4];                               // 0: source ring offsets (+ query offsets), 1: query offsets, 2: target ring offsets + bounding-box keys,
                                                  // 3: the visual matches on their way to the device (set_visual_impl without a wait)
    DevBuf<int> row_off_vis, row_off_icp;
    DevBuf<double> rows_r, rows_J;

    // camera projection of a scan + keypoint depth association
    DevBuf<float4> pstack, vstack, kp_point, kp_out;
    DevBuf<float2> kps;
    DevBuf<int> proj_off, ring_cnt, kp_flag, kp_excl, kp_has;
    std::vector<int> h_proj_off, h_ring_cnt;   // offsets of the projected cloud (copied: the cloud may be replaced later)
    int proj_rings = 0, proj_points = 0, proj_of_target = 0;
    bool have_projection = false;

    // batched landmark triangulation
    DevBuf<TriFrame> tri_frames;
    DevBuf<double> tri_cam_t;
    DevBuf<velo_tri_obs> tri_obs;
    DevBuf<int> tri_off;
    DevBuf<float> tri_pts;
    DevBuf<unsigned char> tri_init;
    DevBuf<velo_tri_result> tri_res;
    int tri_variant = 1;                 // 1 = one wave per landmark (default), 0 = one thread per landmark (VELO_TRI_VARIANT)

    // lock-step batch driver (velo_frame_to_frame_batch): scratch owned by the FIRST context of a batch
    DevBuf<LMBatchItem> batch_items;
    DevBuf<PoseRecord> batch_pose;       // chain mode of the lock-step driver: per-context records, logs, failure flags
    DevBuf<SolveLog> batch_logs;
    DevBuf<int> batch_fail;
    DevBuf<LMState> batch_states;
    DevBuf<double> batch_x;
    void* h_batch = nullptr;             // pinned: items, x, states
    size_t h_batch_bytes = 0;
    int batch_lockstep = 1;              // VELO_BATCH_LOCKSTEP=0: one host thread per context instead (A/B)

    // sharding / comm
    int shard_rank = 0, shard_world = 1;
    ncclComm_t comm = nullptr;
    // peer-slab all-reduce (velo_comm_peer_export / _attach): my slab, the peers' mappings, my sequence counter and error word
    PeerSlab* peer_slab = nullptr;
    bool peer_on = false;
    PeerComm peer{};
    void* peer_mapped[kMaxPeers] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    DevBuf<unsigned long long> peer_seq;
    DevBuf<unsigned long long> peer_kseq;           // counter of the launch-count agreements (peer_agree_kernel)
    int* h_agree = nullptr;                         // pinned: the agreed launch counts of a chained peer call
    std::vector<void*> peer_retired;                // slabs of earlier exports: a peer's timed-out call may still store into them; freed with the context
    DevBuf<int> peer_err;
    PartialRec* peer_area = nullptr;     // my receive area of the record exchange (fine-grained, exported)
    int peer_area_queries = 0;           // max_queries it was sized for
    void* peer_area_mapped[kMaxPeers] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    PeerRecs peer_recs{};
    bool peer_recs_on = false;
    unsigned long long peer_xseq = 0;    // exchanges so far (all ranks count alike)

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> assoc_events;   // reused pool
    std::vector<std::pair<const char*, uint64_t>> assoc_event_info;  // kernel name + algorithmic bytes of the launch behind each pair
    int assoc_events_used = 0;
    // velo_set_timing(ctx, 2): every instrumented launch (association, LM, index build) is bracketed by the start / stop events of
    // hipExtLaunchKernelGGL; a call's brackets are read after its final synchronisation and added up per kernel name (velo_get_kernel_times)
    struct TimedLaunch { hipEvent_t a = nullptr, b = nullptr; const char* name = nullptr; uint64_t bytes = 0; };
    std::vector<TimedLaunch> klog;
    int klog_used = 0;
    struct KernelAcc { const char* name; double ms; int64_t launches, sampled; uint64_t bytes; };   // ms: of the `sampled` bracketed launches
    int timing_every = 8;                    // level 2 brackets every n-th launch of a kernel name (a bracket costs ~5 us of queue time); level 3: every launch
    std::vector<KernelAcc> kacc;
    const char* lm_kernel_name = nullptr;    // the LM kernel the last call launched (its evaluations' algorithmic bytes are known only afterwards)
    const char* lm_round_name[VELO_MAX_SOLVES] = {};   // ... per solve of a chained call: rounds may take different kernels (small solve / sweep + step)
    // a chained call that misses is repeated host-driven: what the abandoned chain logged (counted launches, brackets) is dropped with it
    struct TimingMark { std::vector<KernelAcc> kacc; int klog_used = 0, assoc_events_used = 0; };
    void timing_mark(TimingMark* m) const { if (timing >= 2) { m->kacc = kacc; m->klog_used = klog_used; m->assoc_events_used = assoc_events_used; } }
    void timing_rewind(const TimingMark& m) { if (timing >= 2) { kacc = m.kacc; klog_used = m.klog_used; assoc_events_used = m.assoc_events_used; } }
};

namespace {

// ---- per-kernel launch times (velo_set_timing(ctx, 2)) ----
velo_ctx::KernelAcc* kacc_find(velo_ctx* c, const char* name) {
    for (auto& a : c->kacc) if (a.name == name || std::strcmp(a.name, name) == 0) return &a;
    c->kacc.push_back({name, 0.0, 0, 0, 0});
    return &c->kacc.back();
}
void kacc_add(velo_ctx* c, const char* name, double ms, int64_t launches, int64_t sampled, uint64_t bytes) {
    if (!name) return;
    velo_ctx::KernelAcc* a = kacc_find(c, name);
    a->ms += ms; a->launches += launches; a->sampled += sampled; a->bytes += bytes;
}
// Association launches keep their own event pool (velo_summary::assoc_kernel_ms): bracketed one by one at levels 1 and 3, sampled like
// every other kernel at level 2 (where the launch is counted here).  -> bracket this launch?
bool assoc_bracket(velo_ctx* c, const char* name, uint64_t bytes) {
    if (c->timing <= 0) return false;
    if (c->timing == 1) return true;
    velo_ctx::KernelAcc* a = kacc_find(c, name);
    a->launches++; a->bytes += bytes;
    return c->timing >= 3 || (a->launches - 1) % c->timing_every == 0;
}
// An event pair for the launch that follows, or null.  Every launch of the kernel is COUNTED (with its bytes); every timing_every-th
// one is bracketed -- a bracket makes the runtime put two more packets into the queue, ~5 us of a chain whose launches take 20 us --
// and velo_get_kernel_times scales the bracketed time up by launches / sampled.
velo_ctx::TimedLaunch* klog_slot(velo_ctx* c, const char* name, uint64_t bytes) {
    if (c->timing < 2 || !name) return nullptr;
    velo_ctx::KernelAcc* a = kacc_find(c, name);
    a->launches++; a->bytes += bytes;
    // (a kernel that has been launched often since the log was reset -- the LM launches: 750 per group in a 20-step region -- is sampled four
    //  times more sparsely from then on: the brackets of every 8th launch cost the C2 headline 2.4 %, 3,905 against 3,995 pairs/s)
    const int every = c->timing >= 3 ? 1 : c->timing_every * (a->launches > 128 ? 4 : 1);
    if ((a->launches - 1) % every != 0 || c->klog_used >= 1024) return nullptr;
    if (c->klog_used >= (int)c->klog.size()) {
        velo_ctx::TimedLaunch t;
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        c->klog.push_back(t);
    }
    velo_ctx::TimedLaunch* t = &c->klog[(size_t)c->klog_used++];
    t->name = name; t->bytes = bytes;
    return t;
}
// launch `kernel` on `stream`, bracketed by the next event pair of context c's log when it times every launch
#define VELO_LAUNCH_T(c, name, bytes, kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                                            \
        velo_ctx::TimedLaunch* tl__ = klog_slot(c, name, bytes);                                                                    \
        hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, tl__ ? tl__->a : nullptr, tl__ ? tl__->b : nullptr, 0, __VA_ARGS__); \
    } while (0)

// a context about to load a NEW target: a TargetData other contexts still hold is left to them
void own_target(velo_ctx* c) {
    if (!c->T || c->T.use_count() > 1) c->T = std::make_shared<TargetData>();
}

int q_range(const velo_ctx* c, int* b, int* e) {
    const int64_t nq = c->n_q;
    *b = (int)(nq * c->shard_rank / c->shard_world);
    *e = (int)(nq * (c->shard_rank + 1) / c->shard_world);
    return VELO_OK;
}

// next valid-counter: returns with c->nv_idx switched to a counter that is zero at this point of the stream
int next_valid_counter(velo_ctx* c) {
    VELO_TRY(c->n_valid.reserve(2));
    c->nv_idx ^= 1;
    if (!c->nv_clean[c->nv_idx]) HIP_TRY(hipMemsetAsync(c->n_valid.p + c->nv_idx, 0, sizeof(int), c->stream));
    c->nv_clean[c->nv_idx] = false;      // about to be counted into
    return VELO_OK;
}

int upload_cloud(velo_ctx* c, const float* xyz, int64_t stride, int n, int on_device, DevBuf<float4>& dst) {
    VELO_TRY(dst.reserve((size_t)std::max(n, 1)));
    if (n == 0) return VELO_OK;
    const char* dsrc = (const char*)xyz;
    if (!on_device) {
        const size_t bytes = (size_t)(n - 1) * (size_t)stride + 12;
        VELO_TRY(c->staging.reserve(bytes));
        HIP_TRY(hipMemcpyAsync(c->staging.p, xyz, bytes, hipMemcpyHostToDevice, c->stream));
        dsrc = c->staging.p;
    }
    VELO_LAUNCH_T(c, "pack_points_kernel", 28ull * (uint64_t)n, pack_points_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, dsrc, stride, n, dst.p);
    HIP_TRY(hipGetLastError());
    return VELO_OK;
}

// the hinted upload (velo_hint_next_source), issued where the calling thread is about to wait for this frame's chain anyway
int prefetch_issue(velo_ctx* c) {
    if (!c->pf.hinted || c->pf.ready || !c->pf.host || c->pf.bytes == 0) return VELO_OK;
    const int nb = c->pf.buf ^ 1;
    // The announced cloud goes into page-locked memory of the library's own, copied by the thread that is about to wait for the running chain
    // (65 us per 1.44 MB, hidden there), and the launch that ingests it reads that memory itself, every record once, with 16-byte loads
    // (advance_ingest_kernel: 2.9 MB for a group of two in ~45 us more, the bus's rate) -- no copy of the runtime's at all: 3,840-3,870 pairs/s
    // in every run, 0.94 x the resident rate.  Handing the runtime the caller's pageable pointer (its staged path, on a copy stream of our
    // own: VELO_PF_PAGEABLE=1 in the diagnostics build) gives 4,080-4,110 when nothing goes wrong, but the call blocks its caller for 5-12 ms
    // once in ~ 200 copies with four busy queues: two runs in five read 3,020-3,050.  The two buffers alternate: the one filled now is read
    // by the ingest enqueued right behind this call's chain, the other one by the ingest of one step ago, which has long run.
    static const bool pageable = dev_env("VELO_PF_PAGEABLE") != nullptr;
    if (!pageable) {
        if (c->pf.pin_cap[nb] < c->pf.bytes) {
            if (c->pf.pin[nb]) { (void)hipHostFree(c->pf.pin[nb]); c->pf.pin[nb] = nullptr; c->pf.pin_cap[nb] = 0; }
            const size_t want = c->pf.bytes + c->pf.bytes / 8 + 4096;
            HIP_TRY(hipHostMalloc((void**)&c->pf.pin[nb], want, hipHostMallocDefault));
            c->pf.pin_cap[nb] = want;
        }
        static const bool slow_trace_p = dev_env("VELO_SLOW_TRACE") != nullptr;
        const auto tm0 = std::chrono::steady_clock::now();
        std::memcpy(c->pf.pin[nb], c->pf.host, c->pf.bytes);
        if (slow_trace_p) fprintf(stderr, "[velo slow] prefetch_issue: memcpy of %zu bytes into the page-locked buffer %.0f us\n", c->pf.bytes,
                                  std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tm0).count());
        c->pf.buf = nb; c->pf.ready = true; c->pf.hinted = false; c->pf.in_pin = true;
        return VELO_OK;
    }
    if (!c->pf.stream) HIP_TRY(hipStreamCreateWithFlags(&c->pf.stream, hipStreamNonBlocking));
    if (!c->pf.ev) HIP_TRY(hipEventCreateWithFlags(&c->pf.ev, hipEventDisableTiming));
    VELO_TRY(c->pf.land[nb].reserve(c->pf.bytes));
    static const bool slow_trace = dev_env("VELO_SLOW_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(c->pf.land[nb].p, c->pf.host, c->pf.bytes, hipMemcpyHostToDevice, c->pf.stream));
    if (slow_trace) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us > 1000.0) fprintf(stderr, "[velo slow] prefetch_issue: hipMemcpyAsync of %zu pageable bytes took %.0f us\n", c->pf.bytes, us);
    }
    HIP_TRY(hipEventRecord(c->pf.ev, c->pf.stream));
    c->pf.buf = nb; c->pf.ready = true; c->pf.hinted = false; c->pf.in_pin = false;
    return VELO_OK;
}

double gate_of_iter(const velo_params& P, int iter) {
    const double it = (double)iter;
    return P.correspondence_thresh_icp / it / it / it / it;     // velo.h:829
}

// largest float f with (double)f <= gate : the reference rejects when (double)dist2 > gate
unsigned gate_bits_of(double gate) {
    if (!(gate >= 0.0)) return 0u;   // negative / NaN gate: only d2 == 0 could pass a ">" test... keep 0
    float f = (float)gate;
    if ((double)f > gate) f = std::nextafterf(f, 0.0f);
    if (std::isinf(f)) f = FLT_MAX;
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}

int build_grid(velo_ctx* c, Grid& G, double gate) {
    G.gate = gate;
    const double radius = std::sqrt(std::max(gate, 0.0));
    double h = std::max(radius * 1.01, 1e-6);
    // Points lie on surfaces, so points-per-cell grows like N * h^2: for clouds denser than one HDL-64E sweep (accumulated
    // maps, BASELINE config 4) shrink the cell like N^-1/2 to keep the per-cell population -- and with it the candidates
    // per query -- at the level the kernel is tuned for.  Any cell size is exact (the box walk handles every gate).
    const double dense_ref = dev_env("VELO_DENSE_REF") ? atof(dev_env("VELO_DENSE_REF")) : 150000.0;
    if (dense_ref > 0.0 && (double)c->T->n_tgt > dense_ref) h *= std::sqrt(dense_ref / (double)c->T->n_tgt);
    h = std::max(h, 1e-6);
    const double ext[3] = {(double)c->T->bbox[3] - c->T->bbox[0], (double)c->T->bbox[4] - c->T->bbox[1], (double)c->T->bbox[5] - c->T->bbox[2]};
    int dims[3];
    for (;;) {   // per-axis <= 8192 cells and <= 2^25 cells in all, else coarsen (still exhaustive: cell >= radius)
        bool ok = true;
        double total = 1.0;
        for (int k = 0; k < 3; k++) {
            const double dk = std::floor(std::max(ext[k], 0.0) / h) + 1.0;
            if (dk > 8192.0) ok = false;
            dims[k] = (int)std::min(dk, 8192.0);
            total *= dk;
        }
        // (a target loaded by velo_register_batch for one of several registrations in flight: 2^24 cells.  Measured on the 2M-point map,
        //  where the cap decides -- 23 M cells of 6.2 cm or 11.5 M of 7.8 cm: one pair 2.29 vs 2.36 ms, eight pairs in flight 1,340-1,370
        //  vs 1,450-1,465 pairs/s: the bigger cells cost the lone search 3 %, the half-size table -- build, and the lines every other
        //  queue's kernels compete with -- is worth 8 % to the batch.  Any cell size is exact.)
        static const double cap_env = dev_env("VELO_GRID_CAP") ? std::max(atof(dev_env("VELO_GRID_CAP")), 4096.0) : 0.0;
        const double cell_cap = cap_env > 0.0 ? cap_env : (c->batch_load ? 16777216.0 : 33554432.0);
        if (ok && total <= cell_cap) break;
        h *= 1.26;
    }
    G.d.ox = c->T->bbox[0]; G.d.oy = c->T->bbox[1]; G.d.oz = c->T->bbox[2];
    G.d.inv_h = (float)(1.0 / h);
    G.h = h;
    G.d.nx = dims[0]; G.d.ny = dims[1]; G.d.nz = dims[2];
    G.d.ncells = dims[0] * dims[1] * dims[2];
    const int nc = G.d.ncells, n = c->T->n_tgt;
    const size_t ns = (size_t)n + kGridPad;
    VELO_TRY(G.sorted.reserve(ns)); VELO_TRY(G.sring.reserve(ns));
    VELO_TRY(c->scan_total.reserve(1));
    // Measured on the 2M-point map (round 4, tools/build_times.py, tools/ab_env.py): the compressed table moves 175 MB per build instead of
    // 466 MB, yet the build takes 218 us instead of 191 (grid_mark 64 us: the atomicOr's of neighbouring points meet on the same 8-byte word;
    // grid_ccount 41 us) and the search pays the dependent load of every look-up with 8 % (association launch 522 vs 466-485 us; 8 pairs in
    // flight 1,428 vs 1,475 pairs/s).  Exact (every full-size parity test passes on it), NOT kept: the dense table stays the default,
    // VELO_GRID_COMPRESS=1 (diagnostics build) switches it on.
    const int compress_env = dev_env("VELO_GRID_COMPRESS") ? atoi(dev_env("VELO_GRID_COMPRESS")) : 0;
    const bool compressed = compress_env != 0;
    if (compressed) {
        // occupancy bits -> occupied cells before every word (one-pass scan over the words) -> points per occupied cell -> their starts
        // (one-pass scan over at most n + 1 entries) -> scatter.  The compact table is sized by the point count: its length on the device
        // (the number of occupied cells) is never needed on the host.
        G.wpr = (G.d.nx + 63) / 64;
        G.n_points_cap = n;
        const size_t nw = G.n_words();
        if (nw + 1 > (size_t)0x7fffffff) return fail(VELO_ERR_INVALID, "grid too large for the compressed table");
        VELO_TRY(G.wmask.reserve(nw + 1)); VELO_TRY(G.wprefix.reserve(nw + 4));
        VELO_TRY(G.cell_start.reserve((size_t)n + 8));
        const int nwi = (int)nw + 1;                                       // scanned entries: every word + the sentinel
        const int tiles_w = cdiv(nwi, lb_tile(kLbItemsSmall));
        const bool large_c = n + 1 >= kLbLargeFrom;
        const int tiles_c = cdiv(n + 1, lb_tile(large_c ? kLbItemsLarge : kLbItemsSmall));
        VELO_TRY(c->lb_status.reserve((size_t)tiles_w + 1 + (size_t)tiles_c + 1));   // two scans, a status region (+ ticket) each
        HIP_TRY(hipMemsetAsync(G.wmask.p, 0, sizeof(unsigned long long) * (nw + 1), c->stream));
        HIP_TRY(hipMemsetAsync(G.cell_start.p, 0, sizeof(int) * ((size_t)n + 8), c->stream));
        HIP_TRY(hipMemsetAsync(c->lb_status.p, 0, sizeof(unsigned long long) * ((size_t)tiles_w + 1 + (size_t)tiles_c + 1), c->stream));
        c->lb_zeroed = 0;
        unsigned long long* st_w = c->lb_status.p;
        unsigned long long* st_c = c->lb_status.p + tiles_w + 1;
        if (n > 0) VELO_LAUNCH_T(c, "grid_mark_kernel", 20ull * (uint64_t)n, grid_mark_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, G.d, c->T->tgt.p, n, c->T->tgt_cell_of.p, G.wmask.p, G.wpr);
        VELO_LAUNCH_T(c, "word_popc_kernel", 12ull * (uint64_t)nw, word_popc_kernel, dim3(cdiv(nwi, 256)), dim3(256), 0, c->stream, (const unsigned long long*)G.wmask.p, (int)nw, G.wprefix.p);
        VELO_LAUNCH_T(c, "scan_lookback_kernel", 8ull * (uint64_t)nwi, scan_lookback_kernel<kLbItemsSmall>, dim3(tiles_w), dim3(kScanThreads), 0, c->stream, G.wprefix.p, nwi, st_w,
                      reinterpret_cast<int*>(st_w + tiles_w), c->scan_total.p);
        if (n > 0) VELO_LAUNCH_T(c, "grid_ccount_kernel", 8ull * (uint64_t)n, grid_ccount_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, G.d, c->T->tgt_cell_of.p, n,
                                 (const unsigned long long*)G.wmask.p, (const int*)G.wprefix.p, G.wpr, G.table());
        if (large_c) VELO_LAUNCH_T(c, "scan_lookback_kernel", 8ull * (uint64_t)(n + 1), scan_lookback_kernel<kLbItemsLarge>, dim3(tiles_c), dim3(kScanThreads), 0, c->stream, G.table() + 1, n, st_c,
                                   reinterpret_cast<int*>(st_c + tiles_c), c->scan_total.p);
        else VELO_LAUNCH_T(c, "scan_lookback_kernel", 8ull * (uint64_t)(n + 1), scan_lookback_kernel<kLbItemsSmall>, dim3(tiles_c), dim3(kScanThreads), 0, c->stream, G.table() + 1, n, st_c,
                           reinterpret_cast<int*>(st_c + tiles_c), c->scan_total.p);
        VELO_LAUNCH_T(c, "grid_scatter_kernel", 44ull * (uint64_t)n, grid_scatter_kernel, dim3(cdiv(std::max(n, kGridPad), 256)), dim3(256), 0, c->stream, c->T->tgt.p, c->T->tgt_cell_of.p, c->T->tgt_ring_of.p, n,
                      G.table() + 1, (const int*)c->scan_total.p, c->T->tgt_first_point, G.sorted.p, G.sring.p);
        HIP_TRY(hipGetLastError());
        G.built = true;
        return VELO_OK;
    }
    G.wpr = 0; G.n_points_cap = 0;
    VELO_TRY(G.cell_start.reserve_roomy((size_t)nc + 4, (size_t)nc / 2));
    // count -> one-pass exclusive scan -> scatter, all in the table itself with an offset of one (grid_count_kernel, scan_lookback_kernel)
    const bool large_tiles = nc >= kLbLargeFrom;
    const int n_tiles = cdiv(nc, lb_tile(large_tiles ? kLbItemsLarge : kLbItemsSmall));
    VELO_TRY(c->lb_status.reserve((size_t)n_tiles + 1));               // tile status words + the ticket counter behind them
    if (!c->adv) HIP_TRY(hipMemsetAsync(G.cell_start.p, 0, sizeof(int) * ((size_t)nc + 4), c->stream));   // (collected loads: the group's clear launch, advance_clear_kernel)
    if (c->lb_zeroed < n_tiles + 1) HIP_TRY(hipMemsetAsync(c->lb_status.p, 0, sizeof(unsigned long long) * ((size_t)n_tiles + 1), c->stream));
    c->lb_zeroed = 0;                                                  // (about to be used)
    if (c->adv) {                                                      // collected: count rides in the group's ingest launch, scan and scatter in the group's
        AdvJob& J = *c->adv;
        J.g = G.d; J.cell_of = c->T->tgt_cell_of.p; J.table = G.table(); J.nc = nc; J.n_tiles = n_tiles;
        J.clear = G.cell_start.p; J.n_clear = (int)std::min((((size_t)nc + 4 + 3) / 4) * 4, G.cell_start.cap);
        J.lb_status = c->lb_status.p; J.lb_ticket = reinterpret_cast<int*>(c->lb_status.p + n_tiles); J.scan_total = c->scan_total.p;
        J.sorted = G.sorted.p; J.sring = G.sring.p; J.first_point = c->T->tgt_first_point; J.nb_sc = cdiv(std::max(n, kGridPad), 256);
        G.built = true;
        return VELO_OK;
    }
    // (bytes: what each kernel must move given this index layout -- count: cloud in, cell ids out; scan: table in + out; scatter: cloud + ids in, sorted copy out)
    if (n > 0) VELO_LAUNCH_T(c, "grid_count_kernel", 20ull * (uint64_t)n, grid_count_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, G.d, c->T->tgt.p, n, c->T->tgt_cell_of.p, G.table());
    if (large_tiles) VELO_LAUNCH_T(c, "scan_lookback_kernel", 8ull * (uint64_t)nc, scan_lookback_kernel<kLbItemsLarge>, dim3(n_tiles), dim3(kScanThreads), 0, c->stream, G.table() + 1, nc, c->lb_status.p,
                                   reinterpret_cast<int*>(c->lb_status.p + n_tiles), c->scan_total.p);
    else VELO_LAUNCH_T(c, "scan_lookback_kernel", 8ull * (uint64_t)nc, scan_lookback_kernel<kLbItemsSmall>, dim3(n_tiles), dim3(kScanThreads), 0, c->stream, G.table() + 1, nc, c->lb_status.p,
                       reinterpret_cast<int*>(c->lb_status.p + n_tiles), c->scan_total.p);
    VELO_LAUNCH_T(c, "grid_scatter_kernel", 44ull * (uint64_t)n, grid_scatter_kernel, dim3(cdiv(std::max(n, kGridPad), 256)), dim3(256), 0, c->stream, c->T->tgt.p, c->T->tgt_cell_of.p, c->T->tgt_ring_of.p, n,
                  G.table() + 1, (const int*)c->scan_total.p, c->T->tgt_first_point, G.sorted.p, G.sring.p);
    HIP_TRY(hipGetLastError());
    G.built = true;
    return VELO_OK;
}

// (re)build THE grid: one fine grid, cell ~ the smallest gate radius among iter = 1..f2f_iterations, serves all gates
int build_grids(velo_ctx* c) {
    double gmin = gate_of_iter(c->P, 1);
    for (int it = 2; it <= c->P.f2f_iterations; it++) gmin = std::min(gmin, gate_of_iter(c->P, it));
    if (const char* e = dev_env("VELO_GRID_GATE")) gmin = atof(e);
    if (c->T->grids.empty()) c->T->grids.resize(1);
    return build_grid(c, c->T->grids[0], gmin);
}

Grid* grid_for_iter(velo_ctx* c, int) { return (!c->T->grids.empty() && c->T->grids[0].built) ? &c->T->grids[0] : nullptr; }

// patch order serves the unsharded list only: query shards are defined on the reference's order (the oracle's shard rule), and the
// placement table of VELO_TUBE_MAP reads ring positions off it
// ... and the regular grid only: on the density-shrunk grid of a big map (cells of 5 cm) a patch spans more rows than a ring segment
// and the rounds get slower (2M-point map: 2.87 vs 2.52 ms per registration), so there the list keeps the reference's order.
bool want_patch(const velo_ctx* c) {
    if (c->patch_order == 0 || c->shard_world != 1 || c->tube_map >= 0 || c->ring_order_forced || c->target_sharded) return false;
    if (c->patch_order >= 2) return true;                             // A/B: patch order whatever the grid
    if (c->have_target && c->T && !c->T->grids.empty() && c->T->grids[0].built) {
        const int reach = (int)std::ceil(std::sqrt(std::max(gate_of_iter(c->P, 1), 0.0)) / (c->T->grids[0].h * 0.999));
        if (reach > 5) return false;
    }
    return true;
}
bool query_list_stale(const velo_ctx* c) {
    return c->src_skip != std::max(c->P.icp_skip, 1) || (c->n_q > 0) != (c->P.enable_icp != 0 && c->h_q_off[c->n_src_rings] > 0) || c->q_patch != want_patch(c);
}

int pin_acquire(velo_ctx* c, int k, size_t n, int** out) {
    velo_ctx::PinSlot& s = c->pin[k];
    if (s.pending) { HIP_TRY(hipEventSynchronize(s.ev)); s.pending = false; }
    if (s.cap < n) {
        static const bool alloc_trace = dev_env("VELO_ALLOC_TRACE") != nullptr;
        if (alloc_trace) fprintf(stderr, "[velo alloc] pinned slot %d: %zu -> %zu ints\n", k, s.cap, n + 64);
        if (s.p) { (void)hipHostFree(s.p); s.p = nullptr; s.cap = 0; }
        HIP_TRY(hipHostMalloc((void**)&s.p, (n + 64) * sizeof(int)));
        s.cap = n + 64;
    }
    if (!s.ev) HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    *out = s.p;
    return VELO_OK;
}
int pin_release(velo_ctx* c, int k) {
    HIP_TRY(hipEventRecord(c->pin[k].ev, c->stream));
    c->pin[k].pending = true;
    return VELO_OK;
}

int build_query_list(velo_ctx* c) {
    const int skip = std::max(c->P.icp_skip, 1);
    c->prev_ready = false;                                            // seeds are indexed by query
    c->h_q_off.assign((size_t)c->n_src_rings + 1, 0);
    for (int r = 0; r < c->n_src_rings; r++) {
        const int n = c->h_src_off[r + 1] - c->h_src_off[r];
        c->h_q_off[r + 1] = c->h_q_off[r] + (n + skip - 1) / skip;        // smi = 0, skip, 2 skip, ... < n  (velo.h:807)
    }
    c->n_q = c->P.enable_icp ? c->h_q_off[c->n_src_rings] : 0;            // velo.h:806 `* enable_icp`
    c->src_skip = skip;
    VELO_TRY(c->q_off.reserve((size_t)c->n_src_rings + 1));
    VELO_TRY(c->q_src.reserve((size_t)std::max(c->n_q, 1)));
    {
        int* pin = nullptr;
        VELO_TRY(pin_acquire(c, 1, (size_t)c->n_src_rings + 1, &pin));
        std::memcpy(pin, c->h_q_off.data(), sizeof(int) * ((size_t)c->n_src_rings + 1));
        HIP_TRY(hipMemcpyAsync(c->q_off.p, pin, sizeof(int) * ((size_t)c->n_src_rings + 1), hipMemcpyHostToDevice, c->stream));
        VELO_TRY(pin_release(c, 1));
    }
    if (c->n_q > 0) {
        hipLaunchKernelGGL(query_list_kernel, dim3(cdiv(c->n_q, 256)), dim3(256), 0, c->stream, c->src_off.p, c->q_off.p, c->n_src_rings, skip, c->n_q,
                           want_patch(c) ? 1 : 0, c->patch_rings, c->patch_len, c->q_src.p);
        HIP_TRY(hipGetLastError());
    }
    c->q_patch = want_patch(c);
    const size_t nq = (size_t)std::max(c->n_q, 1);
    if (skip == 1 && !c->q_patch) c->qpts = c->src.p;                     // q_src[i] == i
    else {
        VELO_TRY(c->qpts_buf.reserve(nq));
        if (c->n_q > 0) hipLaunchKernelGGL(gather_queries_kernel, dim3(cdiv(c->n_q, 256)), dim3(256), 0, c->stream, (const float4*)c->src.p, (const int*)c->q_src.p, c->n_q, c->qpts_buf.p);
        HIP_TRY(hipGetLastError());
        c->qpts = c->qpts_buf.p;
    }
    VELO_TRY(c->cp.reserve(nq)); VELO_TRY(c->cn.reserve(nq)); VELO_TRY(c->cv0.reserve(nq));
    VELO_TRY(c->aux0.reserve(nq)); VELO_TRY(c->aux1.reserve(nq));
    c->have_corr = false;
    return VELO_OK;                                                       // (no host wait: the offsets went through a pinned slot)
}

// Workgroup -> group map of the association kernel.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8) and each
// XCD has its own 4 MB L2, while one launch touches ~17 MB (clouds, sorted copy, cell table, outputs): with groups in ring
// order every XCD streams the whole scene through its L2 (measured hit rate 57 %).  Here XCD k gets the groups whose queries lie
// in the k-th eighth of their ring -- a wedge of the scene across ALL rings, so the load stays balanced (the dense bottom
// rings are shared by all XCDs) and each L2 only has to hold its wedge.  Placement only; results do not depend on it.
int build_group_perm(velo_ctx* c, int qb, int qe, int mode) {
    if (c->perm_qb == qb && c->perm_qe == qe && c->perm_nq == c->n_q && c->perm_mode == mode) return VELO_OK;
    const int groups = cdiv(qe - qb, 64);
    std::vector<int> perm((size_t)std::max(groups, 1));
    constexpr int NX = 8;
    std::vector<std::vector<int>> bucket(NX);
    int ring = 0;
    for (int g = 0; g < groups; g++) {
        const int q = qb + 64 * g + 32;                                  // the group's middle query decides
        const int qq = std::min(q, qe - 1);
        while (ring + 1 < (int)c->h_q_off.size() - 1 && c->h_q_off[(size_t)ring + 1] <= qq) ring++;
        const int len = std::max(c->h_q_off[(size_t)ring + 1] - c->h_q_off[(size_t)ring], 1);
        const int k = mode == 1 ? std::min(NX - 1, (int)((int64_t)(qq - c->h_q_off[(size_t)ring]) * NX / len)) : g % NX;
        bucket[(size_t)k].push_back(g);
    }
    // blockIdx = slot * 8 + k; a bucket that runs dry is refilled from the fullest one (keeps every group exactly once)
    std::vector<size_t> next(NX, 0);
    for (int b = 0; b < groups; b++) {
        int k = b % NX;
        if (next[(size_t)k] >= bucket[(size_t)k].size()) {
            size_t best = 0; int kb = -1;
            for (int j = 0; j < NX; j++) { const size_t left = bucket[(size_t)j].size() - next[(size_t)j]; if (left > best) { best = left; kb = j; } }
            k = kb;
        }
        perm[(size_t)b] = bucket[(size_t)k][next[(size_t)k]++];
    }
    VELO_TRY(c->group_perm.reserve(perm.size()));
    HIP_TRY(hipMemcpyAsync(c->group_perm.p, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                             // perm is a stack-owned host vector
    c->perm_qb = qb; c->perm_qe = qe; c->perm_nq = c->n_q; c->perm_mode = mode;
    return VELO_OK;
}

void pose_scalars(const double x[6], PoseScalars* S) {
    // the point-independent part of ceres::AngleAxisRotatePoint [3P], in double with the host libm
    std::memset(S, 0, sizeof(*S));
    for (int k = 0; k < 3; k++) { S->w[k] = x[k]; S->t[k] = x[3 + k]; }
    const double theta2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    if (theta2 > std::numeric_limits<double>::epsilon()) {
        const double theta = std::sqrt(theta2);
        velo_sincos(theta, &S->s, &S->c);              // the pinned sin / cos (velo_device_math.h): same bits on host, device and in the oracle
        const double ti = 1.0 / theta;
        S->u[0] = x[0] * ti; S->u[1] = x[1] * ti; S->u[2] = x[2] * ti;
        S->omc = 1.0 - S->c;
        S->small = 0;
    } else {
        S->small = 1;
    }
}

VisualParams visual_params(const velo_params& P) {
    VisualParams V;
    V.w_3d2d = P.weight_3D2D; V.w_2d2d = P.weight_2D2D;
    V.th_3d2d = P.loss_thresh_3D2D; V.th_2d2d = P.loss_thresh_2D2D; V.th_3d3d = P.loss_thresh_3D3D;
    V.outlier_reject = P.outlier_reject; V.enable_2d2d = P.enable_2d2d; V.enable_3d2d = P.enable_3d2d;
    return V;
}
LMParams lm_params(const velo_params& P) {
    LMParams Q;
    Q.max_num_iterations = P.max_num_iterations; Q.max_invalid = P.max_consecutive_invalid_steps;
    Q.function_tolerance = P.function_tolerance; Q.gradient_tolerance = P.gradient_tolerance; Q.parameter_tolerance = P.parameter_tolerance;
    Q.initial_radius = P.initial_trust_region_radius; Q.max_radius = P.max_trust_region_radius; Q.min_radius = P.min_trust_region_radius;
    Q.min_relative_decrease = P.min_relative_decrease; Q.min_diag = P.min_lm_diagonal; Q.max_diag = P.max_lm_diagonal;
    return Q;
}

constexpr int kMaxVisBlocks = 64;
constexpr int kEvalPerThread = 4;      // nominal residuals per thread of the ICP sweep (alone: 4 -> 3.4 us of rows per sweep, 2 -> 2.0 us but twice the partial rows for the step: same 15.2 us per iteration; 8 pairs in flight: 2,553 vs 2,348 pairs/s)
struct EvalPlan { int nb_icp, nb_vis; int total() const { return nb_icp + nb_vis; } };

EvalArgs eval_args(velo_ctx* c, const double* x_override) {
    EvalArgs A;
    std::memset(&A, 0, sizeof(A));
    A.state = c->state.p;
    A.pt = c->eval_pt.p;
    A.x_override = x_override;
    A.cp = c->cp.p; A.cn = c->cn.p; A.cv0 = c->cv0.p;
    if (c->have_corr) q_range(c, &A.q_begin, &A.q_end);
    {   // query-sharded: every rank sweeps a contiguous share of the visual matches too (the gate ran on all of them on every rank)
        const int n = c->vflags_valid ? c->n_matches : 0;
        const int W = std::max(c->shard_world, 1), r = c->shard_rank;
        const int m0 = (int)((int64_t)n * r / W), m1 = (int)((int64_t)n * (r + 1) / W);
        A.vm = c->vm.p + m0; A.vflags = c->vflags.p + (size_t)3 * m0;
        A.n_matches = m1 - m0;
    }
    A.loss_a_3dpd = c->P.loss_thresh_3DPD; A.w_3dpd = c->P.weight_3DPD;
    A.V = visual_params(c->P);
    A.partials = c->partials.p;
    A.trace = c->lm_trace_on ? c->lm_trace.p : nullptr;
    A.trace_eval = 0;
    return A;
}

EvalPlan eval_plan(const EvalArgs& A) {
    EvalPlan E;
    const int nq = A.q_end - A.q_begin;
    static const int per_thread_env = dev_env("VELO_EVAL_PER_THREAD") ? std::max(atoi(dev_env("VELO_EVAL_PER_THREAD")), 1) : 0;
    E.nb_vis = A.n_matches > 0 ? std::min(std::max(cdiv(3 * A.n_matches, kEvalThreads), 1), kMaxVisBlocks) : 0;
    // With visual blocks the sweep's workgroups of a lock-step group of two (2 x (118 + 24) at C3) no longer fit one per CU beside the
    // association workgroups: the launch gets a second wave of workgroups.  The point-to-plane rows then go five (six ...) to a thread until a
    // context's workgroups are at most half the CUs again (C3: 94 + 24; 3,721 -> 3,810 pairs/s, one pair 0.991 -> 0.967 ms).  The partition
    // is a function of the problem's size alone, so every path sums the same rows in the same order.
    int per_thread = per_thread_env > 0 ? per_thread_env : kEvalPerThread;
    if (per_thread_env == 0 && E.nb_vis > 0 && nq > 0)
        while (per_thread < 8 && cdiv(nq, kEvalThreads * per_thread) + E.nb_vis > 128) per_thread++;
    E.nb_icp = nq > 0 ? std::min(std::max(cdiv(nq, kEvalThreads * per_thread), 1), kMaxEvalBlocks) : 0;
    return E;
}

// the evaluation sweep: lean point-to-plane kernel + (only when visual blocks exist) the visual kernel
void launch_eval(velo_ctx* c, EvalArgs A, const EvalPlan& E) {
    if (E.nb_icp > 0) hipLaunchKernelGGL(eval_icp_kernel, dim3(E.nb_icp), dim3(kEvalThreads), 0, c->stream, A);
    if (E.nb_vis > 0) {
        A.vis_row0 = E.nb_icp;
        hipLaunchKernelGGL(eval_visual_kernel, dim3(E.nb_vis), dim3(kEvalThreads), 0, c->stream, A);
    }
}

// warm-start seeds of the tube kernel: cleared (index -1) on the first round after a new source or target
bool direct_round(const velo_ctx* c, int nq, bool partial);
// Does this round take its seeds from the target's direction image (seed_kernel ahead of the search)?  The rounds of the first f2f
// iteration do (wide gate; the pose has just been guessed or moved by a whole solve) and any round without predecessors; the
// rounds of later iterations move the pose by millimetres and keep the previous winners.  Whole query list, tube kernel only.
bool seeds_from_image(const velo_ctx* c, int iter, bool partial) {
    if (!c->dimg_seeds || !c->warm_start || partial || !c->T->dimg_built || c->shard_world != 1) return false;
    if (c->assoc_variant >= 0 && c->assoc_variant != 5) return false;
    if (direct_round(c, c->n_q, partial)) return false;
    return iter == 1 || !c->prev_ready || c->seed_rounds == 0;
}
int attach_seeds(velo_ctx* c, AssocOut* out, bool image_seeds = false, int* had_prev = nullptr) {
    out->prev_a = nullptr; out->prev_b = nullptr; out->prev_r = nullptr;
    if (had_prev) *had_prev = 0;
    if (!c->warm_start) return VELO_OK;
    const size_t nq = (size_t)std::max(c->n_q, 1);
    if (!c->prev_ready) {
        VELO_TRY(c->prev_a.reserve(2 * nq)); VELO_TRY(c->prev_r.reserve(nq));   // both winners' arrays in one allocation: one fill
        if (!image_seeds && !(c->prev_filled && c->prev_filled_nq == c->n_q)) HIP_TRY(hipMemsetAsync(c->prev_a.p, 0xff, sizeof(float4) * 2 * nq, c->stream));   // (the seed kernel writes every entry itself)
        c->prev_filled = false;
        c->prev_ready = true;
        c->seed_rounds = 0;
    } else if (had_prev) *had_prev = 1;
    out->prev_a = c->prev_a.p; out->prev_b = c->prev_a.p + nq; out->prev_r = c->prev_r.p;
    return VELO_OK;
}
void fill_seed_args(const velo_ctx* c, SeedArgs* A, const PoseScalars& S, const PoseRecord* P_dev, const int* chain_fail, int qb, int qe, const AssocOut& out, int had_prev) {
    A->P = S; A->P_dev = P_dev; A->chain_fail = P_dev ? chain_fail : nullptr;
    A->qpts = c->qpts; A->q_begin = qb; A->q_end = qe;
    A->dimg = c->T->dimg.p; A->tgt = c->T->tgt.p; A->ring_of = c->T->tgt_ring_of.p; A->first_point = c->T->tgt_first_point;
    A->prev_a = out.prev_a; A->prev_b = out.prev_b; A->prev_r = out.prev_r; A->has_prev = had_prev;
}

// the asker list of a tube launch on a density-shrunk grid (see assoc_asker_kernel); enable = this launch may defer its askers
int attach_askers(velo_ctx* c, AssocOut* out, bool enable) {
    out->ask_count = nullptr; out->ask_count_next = nullptr; out->ask_list = nullptr; out->ask_keys = nullptr; out->ask_rings = nullptr;
    out->ask_map = c->ask_map;
    if (!enable || !c->asker_queue) return VELO_OK;
    const size_t nq = (size_t)std::max(c->n_q, 1);
    VELO_TRY(c->ask_count.reserve(2)); VELO_TRY(c->ask_list.reserve(nq)); VELO_TRY(c->ask_keys.reserve(2 * nq)); VELO_TRY(c->ask_rings.reserve(nq));
    c->ask_idx ^= 1;
    if (!c->ask_clean[c->ask_idx]) HIP_TRY(hipMemsetAsync(c->ask_count.p + c->ask_idx, 0, sizeof(int), c->stream));
    c->ask_clean[c->ask_idx] = false;
    c->ask_clean[c->ask_idx ^ 1] = true;                      // the launch clears the other counter
    out->ask_count = c->ask_count.p + c->ask_idx; out->ask_count_next = c->ask_count.p + (c->ask_idx ^ 1);
    out->ask_list = c->ask_list.p; out->ask_keys = c->ask_keys.p; out->ask_rings = c->ask_rings.p;
    return VELO_OK;
}

// A round may use the lane kernel when it starts from seeds (a round of this source against this target has run) on the regular
// grid (gate radius of the first iteration <= 5 cells; the density-shrunk grid of a 2M-point map keeps the tube kernel and its
// query-by-query second phase), with the default variant and no diagnostics / placement table / partial records.
bool lane_round(const velo_ctx* c, const Grid* G, bool partial) {
    if (!c->assoc_lane || !c->warm_start || c->assoc_variant >= 0 || c->debug_skip || c->tube_map >= 0 || partial) return false;
    if (!c->prev_ready || c->seed_rounds < 1) return false;
    const int reach_cells = (int)std::ceil(std::sqrt(std::max(gate_of_iter(c->P, 1), 0.0)) / (G->h * 0.999));
    return reach_cells <= 5;
}

// Sparse rounds (the reference's icp_skip = 200: 640 queries, metres apart) search one wave per query (assoc_direct_kernel).
// Measured on the 120k-point pair, us per round, tube / direct (tools/skip_sweep.py): icp_skip 200: 88 / 18, 64: 143 / 31,
// 32: 196 / 51, 24: 201 / 63, 16: 178 / 92, 12: 148 / 115, 8 (15k queries): 101 / 167, 4: 63 / 317, 1: 61 / 1,223 -- the tube
// kernel needs neighbouring queries in a group, the direct kernel costs ~12 ns per query.
bool direct_round(const velo_ctx* c, int nq, bool partial) {
    return c->direct_max > 0 && nq > 0 && nq <= c->direct_max && c->src_skip >= c->direct_skip && c->assoc_variant < 0 && !c->debug_skip && c->tube_map < 0 &&
           !partial;
}

int do_associate(velo_ctx* c, const double x[6], int iter, bool want_aux, bool wait, int* n_valid, bool partial = false, const PoseRecord* P_dev = nullptr) {
    if (!c->have_target || !c->have_source) return fail(VELO_ERR_STATE, "associate needs set_target and set_source first");
    if (iter < 1) return fail(VELO_ERR_INVALID, "iter must be >= 1");
    if (query_list_stale(c)) VELO_TRY(build_query_list(c));
    Grid* G = grid_for_iter(c, iter);
    if (!G) return fail(VELO_ERR_STATE, "the target's search index has not been built");
    int qb, qe;
    q_range(c, &qb, &qe);
    if (partial) { qb = 0; qe = c->n_q; VELO_TRY(c->partials_rec.reserve((size_t)std::max(c->n_q, 1))); }   // every query against the local rings
    VELO_TRY(next_valid_counter(c));
    if (qe > qb) {
        PoseScalars S;
        pose_scalars(x, &S);
        GridView V;
        G->view(&V);
        AssocOut out;
        out.p = c->cp.p; out.n = c->cn.p; out.v0 = c->cv0.p; out.aux0 = c->aux0.p; out.aux1 = c->aux1.p; out.n_valid = c->n_valid.p + c->nv_idx; out.dbg = c->dbg.p; out.wg_times = nullptr;
        out.first_ring = c->T->tgt_first_ring; out.first_point = c->T->tgt_first_point; out.partial = partial ? c->partials_rec.p : nullptr;
        out.n_valid_next = nullptr;
        const bool image_seeds = seeds_from_image(c, iter, partial) && !c->debug_skip && c->tube_map < 0;
        int had_prev = 0;
        VELO_TRY(attach_seeds(c, &out, image_seeds, &had_prev));
        VELO_TRY(attach_askers(c, &out, false));
        const bool direct = direct_round(c, qe - qb, partial);
        const bool lane = !direct && lane_round(c, G, partial);
        if (image_seeds && out.prev_a) {                           // seeds of this round: direction image (+ the previous winners), one thread per query
            SeedArgs SA;
            fill_seed_args(c, &SA, S, P_dev, c->chain_fail.p, qb, qe, out, had_prev);
            VELO_LAUNCH_T(c, "seed_kernel", 132ull * (uint64_t)(qe - qb), seed_kernel, dim3(cdiv(qe - qb, 256)), dim3(256), 0, c->stream, SA);
        }
        if (c->debug_skip & 32) { VELO_TRY(c->wg_times.reserve((size_t)16 * cdiv(qe - qb, 64) + 2)); HIP_TRY(hipMemsetAsync(c->wg_times.p, 0, sizeof(unsigned long long) * ((size_t)16 * cdiv(qe - qb, 64) + 2), c->stream)); out.wg_times = c->wg_times.p; c->wg_times_n = cdiv(qe - qb, 64); }
        std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
        const char* assoc_name = direct ? "assoc_direct_kernel" : "assoc_search_v5_kernel";
        const uint64_t assoc_b = 12ull * (uint64_t)(qe - qb) + 12ull * (uint64_t)c->T->n_tgt + 28ull * (uint64_t)(qe - qb);
        if (assoc_bracket(c, assoc_name, assoc_b)) {
            if (c->assoc_events_used >= 256) c->assoc_events_used = 0;      // standalone velo_associate calls: recycle
            if (c->assoc_events_used >= (int)c->assoc_events.size()) {
                hipEvent_t a, b;
                HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
                c->assoc_events.emplace_back(a, b);
                c->assoc_event_info.emplace_back(nullptr, 0);
            }
            c->assoc_event_info[(size_t)c->assoc_events_used] = {assoc_name, assoc_b};
            ev = &c->assoc_events[c->assoc_events_used++];
        }
        // The tube kernel is launched with hipExtLaunchKernelGGL, which stamps the two events with the KERNEL's own start and
        // stop (what a rocprofv3 kernel trace reports); events recorded around a launch would also count the time the launch waits
        // for the chip while other streams' kernels run.  The A/B variants keep the record-around bracket.
        const int variant_timed = c->assoc_variant >= 0 ? c->assoc_variant : 5;
        const bool ext_timed = direct || lane || variant_timed == 5 || (variant_timed >= 52 && variant_timed <= 59);
        if (ev && !ext_timed) HIP_TRY(hipEventRecord(ev->first, c->stream));
        const int aux = want_aux ? 1 : 0;
        const int groups = cdiv(qe - qb, 64);
        const double gate = gate_of_iter(c->P, iter);
        const unsigned gbits = gate_bits_of(gate);
        const float h_safe = (float)(G->h * 0.999);
        // the clustering radius is a length (VELO_CLUSTER_W is given in cells of the default 0.179 m grid)
        const int cluster_cells = std::max(1, (int)std::lround((double)c->cluster_w * 0.1785 / G->h));
#define VELO_LAUNCH_V3(NW, MINW, DBG)                                                                                              \
        hipLaunchKernelGGL((assoc_search_v3_kernel<NW, MINW, DBG>), dim3(c->xcd_map ? ((groups + 7) / 8) * 8 : groups), dim3(NW * 64), 0, c->stream, S, V, c->src.p, c->q_src.p, qb, qe, \
                           c->T->tgt.p, c->T->tgt_off.p, c->T->tgt_ring_of.p, gbits, c->P.icp_norm_condition, cluster_cells, h_safe, out, aux, c->debug_skip, c->xcd_map)
        // the diagnostic hooks (VELO_DEBUG_SKIP != 0) live in a separate instantiation: compiled in, they spill registers
#ifdef VELO_DIAGNOSTICS
#define VELO_LAUNCH_V2(NW, MINW) do { if (c->debug_skip) VELO_LAUNCH_V3(NW, MINW, true); else VELO_LAUNCH_V3(NW, MINW, false); } while (0)
#else
#define VELO_LAUNCH_V2(NW, MINW) VELO_LAUNCH_V3(NW, MINW, false)
#endif
        // Default = tube kernel (5) with warm start.  120k-pt scans: 69 us per launch averaged over the 6 rounds of a call (box
        // kernel 4: 121 us); 2M-pt map: 535 us vs 1.49 ms -- its cold first round is slower there (density-shrunk grid, gate radius
        // = 15 cells, every query asks for a (2e+1)^2-row box: 1.59 vs 1.45 ms) but the five warm rounds need tiny boxes.
        const int variant = direct ? 7 : lane ? 6 : (c->assoc_variant >= 0 ? c->assoc_variant : 5);
        const int reach_cells = (int)std::ceil(std::sqrt(std::max(gate_of_iter(c->P, 1), 0.0)) / (G->h * 0.999));   // > 5: density-shrunk grid
        switch (variant) {
#ifdef VELO_DIAGNOSTICS   // the A/B kernels (per-lane reference walk, pipelined prepare + persistent search): tools' build only
            case 0: {
                const int reach = (int)std::ceil(std::sqrt(std::max(gate, 0.0)) / (G->h * 0.999)) ;
                hipLaunchKernelGGL(assoc_search_kernel, dim3(cdiv(qe - qb, kAssocThreads)), dim3(kAssocThreads), 0, c->stream,
                                   S, V, c->src.p, c->q_src.p, qb, qe, c->T->tgt.p, c->T->tgt_off.p, c->T->tgt_ring_of.p, gbits, c->P.icp_norm_condition, std::max(reach, 1), out, aux);
                break;
            }
            case 104: case 102: case 108: {   // pipelined: prepare (one item per cluster) + persistent per-cluster search
                const int nqr = qe - qb;
                const int shard_cap = (groups / kQShards + 1) * 64;        // worst case: every lane its own cluster
                VELO_TRY(c->items.reserve((size_t)shard_cap * kQShards));
                VELO_TRY(c->item_counters.reserve((size_t)2 * kQShards * kQStride));
                VELO_TRY(c->qpos.reserve((size_t)c->n_q + 64));
                AssocQueue Q;
                Q.items = c->items.p; Q.shard_cap = shard_cap; Q.counters = c->item_counters.p; Q.qpos = c->qpos.p;
                HIP_TRY(hipMemsetAsync(c->item_counters.p, 0, sizeof(int) * 2 * kQShards * kQStride, c->stream));
                (void)nqr;
                hipLaunchKernelGGL(assoc_prepare_kernel, dim3(groups), dim3(64), 0, c->stream, S, V.d, c->src.p, c->q_src.p, qb, qe, cluster_cells, Q);
                const int wgs = std::min(groups * 4, c->persistent_wgs);
                if (variant == 102)
                    hipLaunchKernelGGL((assoc_cluster_kernel<2, 1>), dim3(wgs), dim3(128), 0, c->stream, V, Q, c->src.p, c->q_src.p, qb, qe,
                                       c->T->tgt.p, c->T->tgt_off.p, c->T->tgt_ring_of.p, gbits, c->P.icp_norm_condition, h_safe, out, aux);
                else if (variant == 108)
                    hipLaunchKernelGGL((assoc_cluster_kernel<8, 6>), dim3(wgs), dim3(512), 0, c->stream, V, Q, c->src.p, c->q_src.p, qb, qe,
                                       c->T->tgt.p, c->T->tgt_off.p, c->T->tgt_ring_of.p, gbits, c->P.icp_norm_condition, h_safe, out, aux);
                else
                    hipLaunchKernelGGL((assoc_cluster_kernel<4, 6>), dim3(wgs), dim3(256), 0, c->stream, V, Q, c->src.p, c->q_src.p, qb, qe,
                                       c->T->tgt.p, c->T->tgt_off.p, c->T->tgt_ring_of.p, gbits, c->P.icp_norm_condition, h_safe, out, aux);
                break;
            }
#endif
            case 7: {   // sparse round: one wave per query
                out.n_valid_next = c->n_valid.p + (c->nv_idx ^ 1);
                c->nv_clean[c->nv_idx ^ 1] = true;
                hipExtLaunchKernelGGL(assoc_direct_kernel, dim3(qe - qb), dim3(64), 0, c->stream, ev ? ev->first : nullptr, ev ? ev->second : nullptr, 0,
                                      S, P_dev, P_dev ? c->chain_fail.p : (int*)nullptr, V, c->qpts, qb, qe, (const float4*)c->T->tgt_pad.p, (const int*)c->T->tgt_off.p,
                                      gbits, c->P.icp_norm_condition, h_safe, out, aux);
                break;
            }
#ifdef VELO_DIAGNOSTICS
            case 6: {   // lane kernel: one lane owns one query (rounds that start from seeds)
                out.n_valid_next = c->n_valid.p + (c->nv_idx ^ 1);
                c->nv_clean[c->nv_idx ^ 1] = true;
                hipExtLaunchKernelGGL(assoc_lane_kernel, dim3(cdiv(groups, 4)), dim3(256), 0, c->stream, ev ? ev->first : nullptr, ev ? ev->second : nullptr, 0,
                                      S, P_dev, P_dev ? c->chain_fail.p : (int*)nullptr, V, c->qpts, qb, qe, (const float4*)c->T->tgt_pad.p, (const int*)c->T->tgt_off.p,
                                      gbits, c->P.icp_norm_condition, out, aux);
                break;
            }
#endif
            case 5: case 55: case 52: case 56: case 57: case 58: case 59: {   // tube variant: per-row intervals, per-query phase 2 (cluster radius only when VELO_CLUSTER_W is given)
                // tubes do not grow with the segment, so the cluster radius only has to bound the row box of pathological groups
                // (a 64-query group straddling a gap in its ring): 96 default cells = 17 m unless VELO_CLUSTER_W says otherwise
                const int cw = c->cluster_w_set ? (c->cluster_w > 0 ? cluster_cells : 2000)
                                                : std::max(1, (int)std::lround(96.0 * 0.1785 / G->h));
                out.n_valid_next = c->n_valid.p + (c->nv_idx ^ 1);      // the tube kernel clears it for the next round
                c->nv_clean[c->nv_idx ^ 1] = true;
                const int asker_rows = c->asker_rows >= 0 ? c->asker_rows : (reach_cells > 5 ? 0 : (1 << 30));
                const int* perm = nullptr;
                if (c->tube_map >= 0) { VELO_TRY(build_group_perm(c, qb, qe, c->tube_map)); perm = c->group_perm.p; }
                else if (c->xcd_chunks) perm = c->xcd_chunks == 2 ? kXcdTiles : kXcdChunks;
                const int grid_groups = perm == kXcdChunks ? 8 * cdiv(groups, 8) : (perm == kXcdTiles ? 64 * cdiv(groups, 64) : groups);
                // density-shrunk grid, default instantiation, COLD round (no seeds yet: half of the queries ask, the heavy ones in clumps): the asking
                // queries go on a list and are searched by a second launch.  Measured per round on the 2M-point map, list vs in place: cold 399 vs
                // 688 us; seeded rounds 237 / 326 / 246 / 183 / 160 vs 230 / 272 / 207 / 138 / 116 us (few askers: the second launch only adds its
                // own ~40 us) -- hence the cold round only.
                const bool queue = asker_rows < (1 << 30) && variant == 5 && !c->debug_skip && c->seed_rounds == 0;
                VELO_TRY(attach_askers(c, &out, queue));
                hipEvent_t ev_stop = ev ? ev->second : nullptr;
                if (out.ask_list && ev) ev_stop = nullptr;              // the bracket closes behind the asker launch
#define VELO_LAUNCH_V5(NW, MINW, DBG, PPT, ASKER)                                                                                         \
                hipExtLaunchKernelGGL((assoc_search_v5_kernel<NW, MINW, DBG, PPT, ASKER>), dim3(grid_groups), dim3(NW * 64), c->assoc_lds_pad, c->stream,                    \
                                      ev ? ev->first : nullptr, ev_stop, 0, S, P_dev, P_dev ? c->chain_fail.p : (int*)nullptr, V, c->qpts, qb, qe,                      \
                                   (const float4*)c->T->tgt_pad.p, (const int*)c->T->tgt_off.p, gbits, c->P.icp_norm_condition, cw, h_safe, out, aux, perm, c->debug_skip ? c->debug_skip : (asker_rows < (1 << 30) ? (c->dense_rows | (c->dense_far << 20)) : 0), asker_rows)
                // default: 5 waves/SIMD (96 VGPRs, no spills, no scratch traffic), 2 candidate pairs per trip.  Measured on C2:
                // 62 us; 6 waves + 2 pairs (5 spilled VGPRs) 65; 7 waves + 2 pairs 64; 5 waves + 4 pairs 66; 6 waves + 4 pairs 71
#ifdef VELO_DIAGNOSTICS
                if (c->debug_skip) VELO_LAUNCH_V5(4, 5, true, 2, true);
                else if (variant == 55) VELO_LAUNCH_V5(4, 5, false, 4, true);
                else if (variant == 52) VELO_LAUNCH_V5(4, 6, false, 2, true);
                else if (variant == 56) VELO_LAUNCH_V5(4, 6, false, 2, false);   // occupancy A/B: 6 / 7 / 8 waves per SIMD
                else if (variant == 57) VELO_LAUNCH_V5(4, 7, false, 2, false);
                else if (variant == 58) VELO_LAUNCH_V5(4, 8, false, 2, false);
                // (workgroups of 2 waves / 1 wave -- every group resident at once -- measured 66 / 134 us per launch against 62: not tail-bound)
                else if (variant == 59) VELO_LAUNCH_V5(4, 5, false, 2, false);   // phase 2 through the row/tile machinery (A/B)
                else
#endif
                if (asker_rows >= (1 << 30)) VELO_LAUNCH_V5(4, 5, false, 2, false);   // regular grid: instantiation without the query-by-query code (no spills)
                else if (out.ask_list) {
                    // through the batch entry (arguments read from one struct): as a kernel with 40 scalar arguments this instantiation
                    // spills 17 SGPRs, which makes the dispatch set up scratch (~11 us per launch)
                    AssocBatch B1;
                    std::memset(&B1, 0, sizeof(B1));
                    AssocArgs& a = B1.item[0];
                    a.P = S; a.P_dev = P_dev; a.chain_fail = P_dev ? c->chain_fail.p : nullptr; a.G = V; a.qpts = c->qpts; a.q_begin = qb; a.q_end = qe;
                    a.tgt_pad = c->T->tgt_pad.p; a.tgt_off = c->T->tgt_off.p; a.gate_bits = gbits; a.norm_cond = c->P.icp_norm_condition; a.cluster_w = cw;
                    a.h_safe = h_safe; a.out = out; a.want_aux = aux; a.group_perm = perm; a.dbg = c->dense_rows | (c->dense_far << 20); a.asker_rows = asker_rows;
                    hipExtLaunchKernelGGL((assoc_search_v5_batch_kernel<4, 5, false, 2, 2>), dim3(grid_groups, 1), dim3(256), c->assoc_lds_pad, c->stream,
                                          ev ? ev->first : nullptr, ev_stop, 0, B1);
                    hipExtLaunchKernelGGL(assoc_asker_kernel, dim3(cdiv(qe - qb, kAskChunk) + 8), dim3(64), 0, c->stream, nullptr, ev ? ev->second : nullptr, 0,
                                          S, P_dev, (const int*)(P_dev ? c->chain_fail.p : nullptr), V, c->qpts, (const float4*)c->T->tgt_pad.p, (const int*)c->T->tgt_off.p,
                                          gbits, c->P.icp_norm_condition, h_safe, out, aux);
                }
                else VELO_LAUNCH_V5(4, 5, false, 2, 1);
#undef VELO_LAUNCH_V5
                break;
            }
#ifdef VELO_DIAGNOSTICS   // box-walk kernel (the second independent implementation the variant tests compare against)
            case 1: VELO_LAUNCH_V2(1, 1); break;
            case 2: VELO_LAUNCH_V2(2, 1); break;
            case 8: VELO_LAUNCH_V2(8, 8); break;
            case 45: VELO_LAUNCH_V2(4, 5); break;
            case 46: VELO_LAUNCH_V2(4, 6); break;
            case 47: VELO_LAUNCH_V2(4, 7); break;
            case 48: VELO_LAUNCH_V2(4, 8); break;
            default: VELO_LAUNCH_V2(4, 6); break;
#else
            default: return fail(VELO_ERR_STATE, "association variant %d exists only in the diagnostics build", variant);
#endif
        }
#undef VELO_LAUNCH_V2
#undef VELO_LAUNCH_V3
        HIP_TRY(hipGetLastError());
        if (out.prev_a && (variant == 7 || variant == 6 || variant == 5 || (variant >= 52 && variant <= 59))) c->seed_rounds++;   // these kernels leave seeds behind
        if (ev && !ext_timed) HIP_TRY(hipEventRecord(ev->second, c->stream));
#ifdef VELO_DIAGNOSTICS
        if ((c->debug_skip & 24) && dev_env("VELO_DEBUG_EACH")) {       // per-launch read-out (default: totals when the context goes)
            unsigned long long h[8];
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(h, c->dbg.p, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[velo dbg launch] groups %d iter %d: %llu %llu %llu %llu %llu %llu %llu %llu\n", groups, iter, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
            HIP_TRY(hipMemset(c->dbg.p, 0, sizeof(h)));
        }
#endif
    }
    if (partial) { c->have_partials = true; c->last_partial_iter = iter; if (wait) HIP_TRY(hipStreamSynchronize(c->stream)); return VELO_OK; }
    c->have_corr = true;
    if (wait) {
        HIP_TRY(hipMemcpyAsync(c->h_int, c->n_valid.p + c->nv_idx, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->last_n_valid = c->h_int[0];
        if (n_valid) *n_valid = c->last_n_valid;
    }
    return VELO_OK;
}

// after a stream synchronisation behind peer traffic: a wait that ran into its time limit has set the error word.  The slabs' sequence
// numbers are out of step from then on: the communicator must be attached again (velo_comm_peer_export + _attach on every rank).
int peer_check(velo_ctx* c) {
    if (!c->peer_on) return VELO_OK;
    HIP_TRY(hipMemcpy(c->h_int, c->peer_err.p, sizeof(int), hipMemcpyDeviceToHost));
    if (c->h_int[0]) return fail(VELO_ERR_COMM, "peer exchange timed out: a rank of the communicator did not arrive (attach the communicator again)");
    return VELO_OK;
}

// timing on: the association launches of the call just finished, summed (HIP events); level 2: every logged launch by kernel name
int read_assoc_timing(velo_ctx* c, velo_summary* S) {
    double ms = 0.0;
    for (int k = 0; k < c->assoc_events_used; k++) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, c->assoc_events[k].first, c->assoc_events[k].second));
        ms += t;
        if (c->timing >= 2 && k < (int)c->assoc_event_info.size()) kacc_add(c, c->assoc_event_info[(size_t)k].first, t, 0, 1, 0);   // (counted when enqueued)
    }
    if (S) S->assoc_kernel_ms = ms;
    int done = 0;
    for (; done < c->klog_used; done++) {
        float t = 0.f;
        const hipError_t e = hipEventElapsedTime(&t, c->klog[(size_t)done].a, c->klog[(size_t)done].b);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }   // launches of the NEXT frame, enqueued behind this call (velo_hint_next_frame): read one call later
        if (e != hipSuccess) return fail(VELO_ERR_HIP, "hipEventElapsedTime: %s", hipGetErrorString(e));
        kacc_add(c, c->klog[(size_t)done].name, t, 0, 1, 0);    // (the launch and its bytes were counted when it was enqueued)
    }
    for (int k = done; k < c->klog_used; k++) std::swap(c->klog[(size_t)(k - done)], c->klog[(size_t)k]);
    c->klog_used -= done;
    return VELO_OK;
}

int do_build_visual(velo_ctx* c, const double* x_host, bool x_on_state, int iter, int* n_blocks) {
    // x: either a host vector (copied to xdev) or the device LM state's x
    const int n = c->n_matches;
    c->vflags_valid = true;
    if (n == 0) { if (n_blocks) *n_blocks = 0; c->h_vflags.clear(); return VELO_OK; }
    const double* xd = nullptr;
    if (x_on_state) xd = c->state.p->x;
    else {
        std::memcpy(c->h_x, x_host, sizeof(double) * 6);
        HIP_TRY(hipMemcpyAsync(c->xdev.p, c->h_x, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
        xd = c->xdev.p;
    }
    hipLaunchKernelGGL(visual_gate_kernel, dim3(cdiv(n, 128)), dim3(128), 0, c->stream, xd, visual_params(c->P), c->vm.p, n, iter, c->vflags.p, (int*)nullptr);
    HIP_TRY(hipGetLastError());
    c->h_vflags.resize((size_t)3 * n);
    HIP_TRY(hipMemcpyAsync(c->h_vflags.data(), c->vflags.p, (size_t)3 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int nb = 0;
    for (unsigned char f : c->h_vflags) nb += f ? 1 : 0;
    if (n_blocks) *n_blocks = nb;
    return VELO_OK;
}

void visual_counts(const velo_ctx* c, int* blocks, int* residuals) {
    int nb = 0, nr = 0;
    for (unsigned char f : c->h_vflags) {
        if (!f) continue;
        nb++;
        const int t = f - 1;
        nr += (t == VELO_RESIDUAL_3D3D) ? 3 : (t == VELO_RESIDUAL_2D2D) ? 1 : 2;
    }
    *blocks = nb; *residuals = nr;
}

// enqueue: eval sweep at the state's current point, (all-reduce), LM transition
int enqueue_lm_iteration(velo_ctx* c, const EvalArgs& A_in, const EvalPlan& E, const LMParams& Q) {
    EvalArgs A = A_in;
    A.trace_eval = c->lm_trace_idx++;
    launch_eval(c, A, E);
    const int nblocks = E.total();
    if (c->peer_on) {
        hipLaunchKernelGGL(lm_step_peer_kernel, dim3(1), dim3(256), 0, c->stream, Q, c->state.p, c->eval_pt.p, (const double*)c->partials.p, nblocks, c->peer, (PoseRecord*)nullptr, (SolveLog*)nullptr);
    } else if (c->comm) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(64), 0, c->stream, (const LMState*)c->state.p, (const double*)c->partials.p, nblocks, c->reduced.p);
        // every rank reaches this call the same number of times: `done` is identical on all ranks, and when it is
        // set the kernels above exit early and the buffer keeps its previous (identical) content
        NCCL_TRY(ncclAllReduce(c->reduced.p, c->reduced.p + kNumAcc, kNumAcc, ncclDouble, ncclSum, c->comm, c->stream));
        hipLaunchKernelGGL(lm_step_kernel, dim3(1), dim3(256), 0, c->stream, Q, c->state.p, c->eval_pt.p, (const double*)(c->reduced.p + kNumAcc), 1, A.trace, A.trace_eval, (PoseRecord*)nullptr, (SolveLog*)nullptr);
    } else {
        hipLaunchKernelGGL(lm_step_kernel, dim3(1), dim3(256), 0, c->stream, Q, c->state.p, c->eval_pt.p, (const double*)c->partials.p, nblocks, A.trace, A.trace_eval, (PoseRecord*)nullptr, (SolveLog*)nullptr);
    }
    HIP_TRY(hipGetLastError());
    return VELO_OK;
}

// K LM iterations + the status read-back as ONE graph launch (the launch-bound inner loop of the solve).
int launch_chunk(velo_ctx* c, const EvalArgs& A, const EvalPlan& E, const LMParams& Q, int iters) {
    const bool graphable = c->use_graphs && !c->comm && !c->peer_on;
    if (!graphable) {
        for (int k = 0; k < iters; k++) VELO_TRY(enqueue_lm_iteration(c, A, E, Q));
        HIP_TRY(hipMemcpyAsync(&c->h_status->s, c->state.p, sizeof(LMState), hipMemcpyDeviceToHost, c->stream));
        return VELO_OK;
    }
    // signature of everything the captured nodes bake in; a mismatch re-captures that slot
    std::vector<unsigned char> sig(sizeof(EvalArgs) + sizeof(LMParams) + sizeof(EvalPlan) + sizeof(void*) * 3 + sizeof(int));
    {
        unsigned char* w = sig.data();
        std::memcpy(w, &A, sizeof(EvalArgs)); w += sizeof(EvalArgs);
        std::memcpy(w, &Q, sizeof(LMParams)); w += sizeof(LMParams);
        std::memcpy(w, &E, sizeof(EvalPlan)); w += sizeof(EvalPlan);
        const void* ptrs[3] = {c->state.p, c->eval_pt.p, c->h_status};
        std::memcpy(w, ptrs, sizeof(ptrs)); w += sizeof(ptrs);
        const int zero = 0;
        std::memcpy(w, &zero, sizeof(int));
    }
    int slot = -1;
    for (int k = 0; k < 2; k++) if (c->chunk_graph[k] && c->chunk_graph_iters[k] == iters && c->chunk_graph_sig[k] == sig) slot = k;
    if (slot < 0) {
        slot = (c->chunk_graph[0] && c->chunk_graph_iters[0] != iters) ? 1 : 0;   // slot 0: first-solve chunk size seen first, slot 1: the other
        if (c->chunk_graph[slot]) { (void)hipGraphExecDestroy(c->chunk_graph[slot]); c->chunk_graph[slot] = nullptr; }
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int st = VELO_OK;
        for (int k = 0; k < iters && st == VELO_OK; k++) st = enqueue_lm_iteration(c, A, E, Q);
        hipError_t e1 = hipMemcpyAsync(&c->h_status->s, c->state.p, sizeof(LMState), hipMemcpyDeviceToHost, c->stream);
        hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        if (st != VELO_OK) { if (g) (void)hipGraphDestroy(g); return st; }
        if (e1 != hipSuccess || e2 != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); return fail(VELO_ERR_HIP, "graph capture of the LM chunk failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
        hipError_t e3 = hipGraphInstantiate(&c->chunk_graph[slot], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e3 != hipSuccess) { c->chunk_graph[slot] = nullptr; return fail(VELO_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e3)); }
        c->chunk_graph_iters[slot] = iters;
        c->chunk_graph_sig[slot] = sig;
    }
    HIP_TRY(hipGraphLaunch(c->chunk_graph[slot], c->stream));
    return VELO_OK;
}

// One ceres::Solve on the device.  x_in: host x to start from, or nullptr to continue from the state's x.
int do_solve(velo_ctx* c, const double* x_in, double x_out[6], velo_solve_summary* S, int* eval_launches, int first_chunk = 6) {
    const LMParams Q = lm_params(c->P);
    const EvalArgs A = eval_args(c, nullptr);
    const EvalPlan E = eval_plan(A);
    const double* xd = nullptr;
    if (x_in) {
        std::memcpy(c->h_x, x_in, sizeof(double) * 6);
        HIP_TRY(hipMemcpyAsync(c->xdev.p, c->h_x, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
        xd = c->xdev.p;
    }
    const int max_iters_all = c->P.max_num_iterations + 1;
    c->lm_trace_idx = 0;
#ifdef VELO_DIAGNOSTICS
    if (c->lm_trace_on) {
        HIP_TRY(hipMemset(c->lm_trace.p, 0, (size_t)kTraceMaxEvals * kTraceStages * kTraceWgs * 8));
    }
#endif
    if (!c->comm && !c->peer_on && !c->use_graphs && c->small_solve && E.total() >= 1 && E.total() <= kSmallRows) {
        // small problem (the reference's icp_skip = 200): the whole solve in one single-workgroup launch, one status copy
        hipLaunchKernelGGL(lm_solve_small_kernel, dim3(1), dim3(kEvalThreads), 0, c->stream, A, Q, c->state.p, xd,
                           (const int*)(c->have_corr ? c->n_valid.p + c->nv_idx : nullptr), E.nb_icp, E.nb_vis, max_iters_all + 2,
                           (PoseRecord*)nullptr, (SolveLog*)nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&c->h_status->s, c->state.p, sizeof(LMState), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (!c->h_status->s.done) return fail(VELO_ERR_STATE, "LM did not terminate after %d sweeps", max_iters_all + 2);
    } else if (!c->comm && !c->peer_on && !c->use_graphs && c->lm_merged && x_in && E.nb_icp > 0 && !c->lm_trace_vis_off) {
        // one launch per LM iteration: every sweep workgroup consumes the previous sweep's partial rows itself (lm_iter_kernel).
        // Launch k reads state / partial rows [k & 1] and writes [(k + 1) & 1]; launch 0 starts the solve.  A solve of n
        // evaluations needs n + 1 launches (the last one only finds the solve done); launches behind that copy the state through.
        const size_t half = (size_t)(kMaxEvalBlocks + kMaxVisBlocks) * kNumAcc;
        const int* nvp = c->have_corr ? c->n_valid.p + c->nv_idx : nullptr;
        int k = 0, chunk = first_chunk + 1;
        const int max_launches = c->P.max_num_iterations + 2;
        for (;;) {
            for (int j = 0; j < chunk; j++, k++) {
                EvalArgs Ak = A;
                Ak.trace_eval = k;
                if (E.nb_vis > 0)       // the visual blocks ride in the same launch (workgroups behind the point-to-plane ones)
                    hipLaunchKernelGGL(lm_iter_vis_kernel, dim3(E.total()), dim3(kEvalThreads), 0, c->stream, Ak, Q, (const LMState*)(c->state.p + (k & 1)), c->state.p + ((k + 1) & 1),
                                       (const double*)(c->partials.p + (size_t)(k & 1) * half), E.total(), c->partials.p + (size_t)((k + 1) & 1) * half, k == 0 ? 1 : 0, xd, nvp,
                                       (PoseRecord*)nullptr, (SolveLog*)nullptr, E.nb_icp, E.nb_vis);
                else
                hipLaunchKernelGGL(lm_iter_kernel, dim3(E.nb_icp), dim3(kEvalThreads), 0, c->stream, Ak, Q, (const LMState*)(c->state.p + (k & 1)), c->state.p + ((k + 1) & 1),
                                   (const double*)(c->partials.p + (size_t)(k & 1) * half), E.nb_icp, c->partials.p + (size_t)((k + 1) & 1) * half, k == 0 ? 1 : 0, xd, nvp, (PoseRecord*)nullptr, (SolveLog*)nullptr);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(&c->h_status->s, c->state.p + (k & 1), sizeof(LMState), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->h_status->s.done) break;
            if (k > max_launches + 16) return fail(VELO_ERR_STATE, "LM did not terminate after %d launches", k);
            chunk = 3;
        }
    } else {
    hipLaunchKernelGGL(lm_begin_kernel, dim3(1), dim3(64), 0, c->stream, c->state.p, c->eval_pt.p, xd, (const int*)(c->have_corr ? c->n_valid.p + c->nv_idx : nullptr),
                       (PoseRecord*)nullptr);
    int launched = 0;
    int chunk = first_chunk;                // LM iterations per host round trip
    const int max_iters = c->P.max_num_iterations + 1;
    for (;;) {
        VELO_TRY(launch_chunk(c, A, E, Q, chunk));
        launched += chunk;
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->h_status->s.done) break;
        if (launched > max_iters + 16) return fail(VELO_ERR_STATE, "LM did not terminate after %d sweeps", launched);
        chunk = 3;
    }
    }
    VELO_TRY(peer_check(c));
    const LMState& s = c->h_status->s;
#ifdef VELO_DIAGNOSTICS
    if (c->lm_trace_on) {
        std::vector<unsigned long long> tr((size_t)kTraceMaxEvals * kTraceStages * kTraceWgs);
        HIP_TRY(hipMemcpy(tr.data(), c->lm_trace.p, tr.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < kTraceWgs; w++) if (tr[(size_t)w]) t0 = std::min(t0, tr[(size_t)w]);
        for (int e = 0; e < s.evals && e < kTraceMaxEvals; e++) {
            fprintf(stderr, "[velo lm trace] eval %2d:", e);
            for (int st = 0; st < 10; st++) {
                unsigned long long a = ~0ull, b = 0ull;
                for (int w = 0; w < kTraceWgs; w++) {
                    const unsigned long long v = tr[((size_t)e * kTraceStages + st) * kTraceWgs + w];
                    if (v) { a = std::min(a, v); b = std::max(b, v); }
                }
                if (a == ~0ull) fprintf(stderr, " -"); else fprintf(stderr, " %.2f/%.2f", (double)(a - t0) * 0.01, (double)(b - t0) * 0.01);
            }
            fprintf(stderr, "\n");
        }
    }
#endif
    for (int k = 0; k < 6; k++) x_out[k] = s.x[k];
    if (S) {
        std::memset(S, 0, sizeof(*S));
        S->termination = s.termination; S->lm_iterations = s.iter; S->evaluations = s.evals;
        c->last_n_valid = s.n_valid;
        S->n_icp_valid = s.n_valid;
        visual_counts(c, &S->n_visual_blocks, &S->n_visual_residuals);
        if (c->shard_rank != 0) { S->n_visual_blocks = 0; S->n_visual_residuals = 0; }
        S->initial_cost = s.initial_cost; S->final_cost = s.cost;
    }
    if (eval_launches) *eval_launches = s.evals;
    return VELO_OK;
}

// the target's direction image (seed_kernel): one fill + one atomicMin per point, on the context's stream
int build_direction_image(velo_ctx* c) {
    c->T->dimg_built = false;
    if (!c->dimg_seeds || !c->warm_start) return VELO_OK;
    VELO_TRY(c->T->dimg.reserve((size_t)kDimgW * kDimgH));
    HIP_TRY(hipMemsetAsync(c->T->dimg.p, 0xff, sizeof(unsigned long long) * (size_t)kDimgW * kDimgH, c->stream));
    const int n = c->T->n_tgt;
    if (n > 0) {
        VELO_LAUNCH_T(c, "dimg_build_kernel", 24ull * (uint64_t)n, dimg_build_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const float4*)c->T->tgt.p, n, c->T->dimg.p);
        HIP_TRY(hipGetLastError());
    }
    c->T->dimg_built = true;
    return VELO_OK;
}

// common tail of every way a target enters the context: ring table, ring ids, bounding box, grid
// target_finalize = target_finalize_begin (everything up to the request for the bounding box, no host wait) + target_finalize_end (the
// one synchronisation of set_target, then the index).  The batch driver begins all contexts of a group before it ends the first,
// so one context's wait is covered by the next one's uploads.
int target_finalize_begin(velo_ctx* c) {
    const int n = c->T->n_tgt, n_rings = c->T->n_tgt_rings;
    c->target_early = false;
    c->prev_ready = false;                                            // seeds refer to points of the old target
    for (int r = 0; r < n_rings; r++) if (c->T->h_tgt_off[r + 1] <= c->T->h_tgt_off[r]) return fail(VELO_ERR_INVALID, "target ring %d is empty", r);
    VELO_TRY(c->T->tgt_off.reserve((size_t)n_rings + 1));
    VELO_TRY(c->T->tgt_ring_of.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->T->tgt_cell_of.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->T->tgt_pad.reserve((size_t)n + 2 * (size_t)n_rings + 2));
    HIP_TRY(hipMemcpyAsync(c->T->tgt_off.p, c->T->h_tgt_off.data(), sizeof(int) * ((size_t)n_rings + 1), hipMemcpyHostToDevice, c->stream));
    // bbox of the finite points -> host (the only sync of set_target; the grid dimensions are sized from it)
    unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if (n == 0) {                                                     // (with points the ring_of launch initialises the keys)
        std::memcpy(c->h_int, init, sizeof(init));
        HIP_TRY(hipMemcpyAsync(c->bbox_keys.p, c->h_int, sizeof(init), hipMemcpyHostToDevice, c->stream));
    }
    if (n > 0) {
        VELO_LAUNCH_T(c, "ring_of_kernel", 4ull * (uint64_t)n, ring_of_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, c->T->tgt_off.p, n_rings, n, c->T->tgt_first_ring, c->T->tgt_ring_of.p, c->bbox_keys.p);
        VELO_LAUNCH_T(c, "pad_rings_kernel", 36ull * (uint64_t)n, pad_rings_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const float4*)c->T->tgt.p, (const int*)c->T->tgt_off.p, (const int*)c->T->tgt_ring_of.p, n,
                      c->T->tgt_first_ring, c->T->tgt_pad.p);
        VELO_LAUNCH_T(c, "bbox_kernel", 16ull * (uint64_t)n, bbox_kernel, dim3(std::min(cdiv(n, 256 * 8), 256)), dim3(256), 0, c->stream, c->T->tgt.p, n, c->bbox_keys.p);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(c->h_int + 8, c->bbox_keys.p, sizeof(init), hipMemcpyDeviceToHost, c->stream));
    VELO_TRY(build_direction_image(c));
    return VELO_OK;
}
// velo_set_target's own way in: the caller's records -> packed cloud, ring ids, padded rings, bounding-box request in ONE copy (ring
// offsets + the box's start keys, through a pinned slot) and ONE launch (target_ingest_kernel), instead of upload_cloud + target_finalize_begin
constexpr int kLbWordsCleared = (1 << 25) / lb_tile(kLbItemsLarge) + 2 > kLbLargeFrom / lb_tile(kLbItemsSmall) + 2 ? (1 << 25) / lb_tile(kLbItemsLarge) + 2 : kLbLargeFrom / lb_tile(kLbItemsSmall) + 2;                 // status words of the largest default table (+ ticket)
int target_ingest(velo_ctx* c, const float* xyz, int64_t stride, int on_device) {
    const int n = c->T->n_tgt, n_rings = c->T->n_tgt_rings;
    c->target_early = false;                                          // (a promotion that knows its box sets it again behind this call)
    c->prev_ready = false;                                            // seeds refer to points of the old target
    VELO_TRY(c->T->tgt.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->T->tgt_off.reserve((size_t)n_rings + 1 + 8));         // the six box keys ride behind the offsets
    VELO_TRY(c->T->tgt_ring_of.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->T->tgt_cell_of.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->T->tgt_pad.reserve((size_t)n + 2 * (size_t)n_rings + 2));
    VELO_TRY(c->lb_status.reserve((size_t)kLbWordsCleared));
    if (c->adv) {                                                     // a promotion inside preload_group (packed cloud in place, box known): collected, launched with the group's
        AdvJob& J = *c->adv;
        J.tgt = c->T->tgt.p; J.tgt_off_dev = c->T->tgt_off.p; J.ring_of = c->T->tgt_ring_of.p; J.pad = c->T->tgt_pad.p;
        J.lb_status = c->lb_status.p; J.lb_words = kLbWordsCleared;
        J.n_t = n; J.n_rings_t = n_rings; J.first_ring = c->T->tgt_first_ring; J.nb_t = cdiv(n, 256 * kIngestPerThread);
        std::memcpy(J.off_t, c->T->h_tgt_off.data(), sizeof(int) * ((size_t)n_rings + 1));
        c->lb_zeroed = kLbWordsCleared;
        return VELO_OK;
    }
    const char* dsrc = (const char*)xyz;
    if (!on_device && n > 0) {
        const size_t bytes = (size_t)(n - 1) * (size_t)stride + 12;
        VELO_TRY(c->staging.reserve(bytes));
        HIP_TRY(hipMemcpyAsync(c->staging.p, xyz, bytes, hipMemcpyHostToDevice, c->stream));
        dsrc = c->staging.p;
    }
    unsigned* keys = reinterpret_cast<unsigned*>(c->T->tgt_off.p + n_rings + 1);
    {
        int* pin = nullptr;
        VELO_TRY(pin_acquire(c, 2, (size_t)n_rings + 1 + 8, &pin));
        std::memcpy(pin, c->T->h_tgt_off.data(), sizeof(int) * ((size_t)n_rings + 1));
        for (int k = 0; k < 6; k++) pin[n_rings + 1 + k] = k < 3 ? -1 : 0;   // min keys all ones, max keys zero
        HIP_TRY(hipMemcpyAsync(c->T->tgt_off.p, pin, sizeof(int) * ((size_t)n_rings + 1 + 6), hipMemcpyHostToDevice, c->stream));
        VELO_TRY(pin_release(c, 2));
    }
    if (n > 0) {
        VELO_LAUNCH_T(c, "target_ingest_kernel", 64ull * (uint64_t)n, target_ingest_kernel, dim3(cdiv(n, 256 * kIngestPerThread)), dim3(256), 0, c->stream, dsrc, stride, n,
                      (const int*)c->T->tgt_off.p, n_rings, c->T->tgt_first_ring, c->T->tgt.p, c->T->tgt_ring_of.p, c->T->tgt_pad.p, keys, c->lb_status.p, kLbWordsCleared);
        HIP_TRY(hipGetLastError());
        c->lb_zeroed = kLbWordsCleared;
    }
    HIP_TRY(hipMemcpyAsync(c->h_int + 8, keys, sizeof(unsigned) * 6, hipMemcpyDeviceToHost, c->stream));
    VELO_TRY(build_direction_image(c));
    return VELO_OK;
}
int target_finalize_end(velo_ctx* c) {
    if (c->target_early) { c->target_early = false; return VELO_OK; }     // promote_begin knew the box: everything is enqueued already
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned keys[6];
    std::memcpy(keys, c->h_int + 8, sizeof(keys));
    if (keys[0] == 0xffffffffu) {   // no finite point at all
        for (int k = 0; k < 6; k++) c->T->bbox[k] = 0.f;
    } else {
        for (int k = 0; k < 6; k++) c->T->bbox[k] = key2f(keys[k]);
    }
    for (Grid& G : c->T->grids) G.built = false;                         // keep the buffers: a new target of the same size rebuilds in place
    VELO_TRY(build_grids(c));
    c->have_target = true;
    return VELO_OK;
}

int target_finalize(velo_ctx* c) {
    VELO_TRY(target_finalize_begin(c));
    return target_finalize_end(c);
}

// velo_set_source's own way in (set_source_begin left the records to be read): ring offsets + query offsets in ONE copy, packed cloud +
// query list + query points in ONE launch (source_ingest_kernel) -- what upload_cloud + source_finalize + build_query_list do in five
int source_ingest(velo_ctx* c) {
    const int R = c->n_src_rings, skip = std::max(c->P.icp_skip, 1);
    c->prev_ready = false;                                            // seeds are indexed by query
    c->h_q_off.assign((size_t)R + 1, 0);
    for (int r = 0; r < R; r++) {
        const int n = c->h_src_off[r + 1] - c->h_src_off[r];
        c->h_q_off[r + 1] = c->h_q_off[r] + (n + skip - 1) / skip;        // smi = 0, skip, 2 skip, ... < n  (velo.h:807)
    }
    c->n_q = c->P.enable_icp ? c->h_q_off[R] : 0;                         // velo.h:806 `* enable_icp`
    c->src_skip = skip;
    const bool patch = want_patch(c);
    const size_t nq = (size_t)std::max(c->n_q, 1);
    VELO_TRY(c->src_off.reserve(2 * ((size_t)R + 1) + 8));                // [ring offsets | query offsets | six bounding-box keys]
    VELO_TRY(c->q_src.reserve(nq));
    const bool own_list = !(skip == 1 && !patch);                         // else q_src[i] == i and the source cloud itself is the list
    if (own_list) VELO_TRY(c->qpts_buf.reserve(nq));
    if (c->adv) {                                                         // collected (preload_group): launched with the group's, the box comes back behind that launch
        AdvJob& J = *c->adv;
        if (c->nf.keys.cap < 16) {
            VELO_TRY(c->nf.keys.reserve(16));
            const unsigned init[16] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
            HIP_TRY(hipMemcpy(c->nf.keys.p, init, sizeof(init), hipMemcpyHostToDevice));
            c->nf.parity = 0;
        }
        J.raw = c->src_raw.dsrc; J.stride = c->src_raw.stride; J.src = c->src.p; J.src_off_dev = c->src_off.p; J.q_src = c->q_src.p;
        J.qpts = own_list ? c->qpts_buf.p : (float4*)nullptr;
        J.keys = c->nf.keys.p + 8 * c->nf.parity; J.keys_next = c->nf.keys.p + 8 * (c->nf.parity ^ 1);
        J.h_keys = reinterpret_cast<unsigned*>(c->h_int + 16);
        c->nf.parity ^= 1;
        J.n_s = c->n_src; J.n_rings_s = R; J.nb_pack = cdiv(c->n_src, 256); J.nb_q = 0;   // (the pack workgroups emit the queries themselves)
        J.skip = skip; J.nq = c->n_q; J.patch = patch ? 1 : 0; J.patch_rings = c->patch_rings; J.patch_len = c->patch_len;
        std::memcpy(J.off_s, c->h_src_off.data(), sizeof(int) * ((size_t)R + 1));
        if (c->warm_start && c->n_q > 0) {                                // the seed arrays of the new queries: "no previous winner", written by the query blocks
            VELO_TRY(c->prev_a.reserve(2 * nq)); VELO_TRY(c->prev_r.reserve(nq));
            J.seed_fill = c->prev_a.p;
            c->prev_filled = true; c->prev_filled_nq = c->n_q;
        } else c->prev_filled = false;
        c->src_bbox_valid = true;                                         // (the keys' copy and its event: advance_launch)
        c->src_raw.on = false;
        c->q_patch = patch;
        c->qpts = own_list ? c->qpts_buf.p : c->src.p;
        VELO_TRY(c->cp.reserve(nq)); VELO_TRY(c->cn.reserve(nq)); VELO_TRY(c->cv0.reserve(nq));
        VELO_TRY(c->aux0.reserve(nq)); VELO_TRY(c->aux1.reserve(nq));
        c->have_corr = false;
        c->have_source = true;
        return VELO_OK;
    }
    {
        int* pin = nullptr;
        VELO_TRY(pin_acquire(c, 0, 2 * ((size_t)R + 1) + 8, &pin));
        std::memcpy(pin, c->h_src_off.data(), sizeof(int) * ((size_t)R + 1));
        std::memcpy(pin + R + 1, c->h_q_off.data(), sizeof(int) * ((size_t)R + 1));
        for (int k = 0; k < 6; k++) pin[2 * (R + 1) + k] = k < 3 ? -1 : 0;   // min keys all ones, max keys zero
        HIP_TRY(hipMemcpyAsync(c->src_off.p, pin, sizeof(int) * (2 * ((size_t)R + 1) + 6), hipMemcpyHostToDevice, c->stream));
        VELO_TRY(pin_release(c, 0));
    }
    const int nb_pack = cdiv(c->n_src, 256), nb_q = c->n_q > 0 ? cdiv(c->n_q, 256) : 0;
    VELO_LAUNCH_T(c, "source_ingest_kernel", 28ull * (uint64_t)c->n_src + 32ull * (uint64_t)c->n_q, source_ingest_kernel, dim3(nb_pack + nb_q), dim3(256), 0, c->stream,
                  c->src_raw.dsrc, c->src_raw.stride, c->n_src, c->src.p, nb_pack, (const int*)c->src_off.p, (const int*)(c->src_off.p + R + 1), R, skip, c->n_q,
                  patch ? 1 : 0, c->patch_rings, c->patch_len, c->q_src.p, own_list ? c->qpts_buf.p : (float4*)nullptr, reinterpret_cast<unsigned*>(c->src_off.p + 2 * (R + 1)));
    HIP_TRY(hipGetLastError());
    // the box keys ride back on the stream; every way out of a call synchronises it, so a LATER call (a promotion) may read them
    HIP_TRY(hipMemcpyAsync(c->h_int + 16, c->src_off.p + 2 * (R + 1), sizeof(unsigned) * 6, hipMemcpyDeviceToHost, c->stream));
    if (!c->src_bbox_ev) HIP_TRY(hipEventCreateWithFlags(&c->src_bbox_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->src_bbox_ev, c->stream));
    c->src_bbox_valid = true;
    c->src_raw.on = false;
    c->q_patch = patch;
    c->qpts = own_list ? c->qpts_buf.p : c->src.p;
    VELO_TRY(c->cp.reserve(nq)); VELO_TRY(c->cn.reserve(nq)); VELO_TRY(c->cv0.reserve(nq));
    VELO_TRY(c->aux0.reserve(nq)); VELO_TRY(c->aux1.reserve(nq));
    c->have_corr = false;
    c->have_source = true;
    return VELO_OK;
}
int source_finalize(velo_ctx* c) {
    if (c->src_raw.on) return source_ingest(c);
    VELO_TRY(c->src_off.reserve((size_t)c->n_src_rings + 1));
    {
        int* pin = nullptr;
        VELO_TRY(pin_acquire(c, 0, (size_t)c->n_src_rings + 1, &pin));
        std::memcpy(pin, c->h_src_off.data(), sizeof(int) * ((size_t)c->n_src_rings + 1));
        HIP_TRY(hipMemcpyAsync(c->src_off.p, pin, sizeof(int) * ((size_t)c->n_src_rings + 1), hipMemcpyHostToDevice, c->stream));
        VELO_TRY(pin_release(c, 0));
    }
    VELO_TRY(build_query_list(c));
    c->have_source = true;
    return VELO_OK;
}

}  // namespace

// =====================================================================================================================
// Resident host threads for the batch entry points: a step of 8 pairs used to create and join 3 (lock-step groups) or 8 (one per
// context) std::threads inside the timed region, every step.  The workers are created on first need and then wait on a condition
// variable between calls; a call hands out task indices 1..n-1 and runs task 0 itself.  One call at a time owns the pool -- a
// second caller that arrives meanwhile (another host thread driving another device) spawns its own threads as before.
class WorkerPool {
public:
    static WorkerPool& instance() { static WorkerPool p; return p; }
    template <typename F>
    void run(int n, F&& fn) {
        if (n <= 1) { if (n == 1) fn(0); return; }
        std::unique_lock<std::mutex> owner(owner_, std::try_to_lock);
        if (!owner.owns_lock()) {                                   // pool busy: plain threads for this call
            std::vector<std::thread> th;
            for (int i = 1; i < n; i++) th.emplace_back([&fn, i]() { fn(i); });
            fn(0);
            for (auto& t : th) t.join();
            return;
        }
        std::function<void(int)> f = [&fn](int i) { fn(i); };
        {
            std::lock_guard<std::mutex> lk(m_);
            while ((int)workers_.size() < n - 1) workers_.emplace_back([this]() { loop(); });
            fn_ = &f; next_ = 1; n_ = n; pending_ = n - 1; generation_++;
        }
        wake_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
    }
private:
    void loop() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            wake_.wait(lk, [&]() { return stop_ || (generation_ != seen && next_ < n_); });
            if (stop_) return;
            while (next_ < n_) {
                const int i = next_++;
                const std::function<void(int)>* f = fn_;
                lk.unlock();
                (*f)(i);
                lk.lock();
                if (--pending_ == 0) done_.notify_all();
            }
            seen = generation_;
        }
    }
    std::mutex owner_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* fn_ = nullptr;
    int next_ = 0, n_ = 0, pending_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
};

static int associate_target_sharded(velo_ctx* c, const double x[6], int iter, bool want_aux);
static int launch_merge(velo_ctx* c, const PartialRec* tables, int world, int stride, int iter, bool want_aux);

extern "C" {

const char* velo_last_error(void) {
    if (g_err.empty()) { std::lock_guard<std::mutex> lk(g_err_mutex); g_err = g_err_shared; }
    return g_err.c_str();
}
const char* velo_version(void) { return "velo_hip 0.1 (gfx950)"; }

int velo_default_params(velo_params* p) {
    if (!p) return fail(VELO_ERR_INVALID, "null params");
    default_params(p);
    return VELO_OK;
}

int velo_create(velo_ctx** out, int device) {
    if (!out) return fail(VELO_ERR_INVALID, "null out");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(VELO_ERR_NODEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= count) return fail(VELO_ERR_INVALID, "device %d out of range (0..%d)", device, count - 1);
    HIP_TRY(hipSetDevice(device));
    if (getenv("VELO_SPIN")) (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
    (void)hipGetLastError();
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VELO_ERR_NODEVICE, "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
    velo_ctx* c = new velo_ctx();
    c->device = device;
    // everything that can fail after the allocation runs inside init, so that a failure releases what was already created
    // (stream, pinned buffers, events, device buffers) instead of leaking it behind a NULL *out
    auto init = [&]() -> int {
        default_params(&c->P);
        for (int k = 0; k < VELO_MAX_SOLVES; k++) { c->pred_evals[k] = (k == 0) ? 12 : 5; c->eval_hist_n[k] = 0; }
        if (const char* e = dev_env("VELO_ASSOC_VARIANT")) c->assoc_variant = atoi(e);
        if (const char* e = dev_env("VELO_CLUSTER_W")) { c->cluster_w = std::max(atoi(e), 0); c->cluster_w_set = true; }
        if (const char* e = dev_env("VELO_TRI_VARIANT")) c->tri_variant = atoi(e);
#ifdef VELO_DIAGNOSTICS
        if (dev_env("VELO_LM_TRACE") && atoi(dev_env("VELO_LM_TRACE"))) {
            c->lm_trace_on = true;
            VELO_TRY(c->lm_trace.reserve((size_t)kTraceMaxEvals * kTraceStages * kTraceWgs));
        }
        // the diagnostic instantiations (cycle stamps, counters, sections switched off -- some bits give WRONG results on purpose) exist
        // only in the tools' build of this file (build.py: libvelo_hip_diag.so); the product library ignores the variable
        if (const char* e = dev_env("VELO_DEBUG_SKIP")) c->debug_skip = atoi(e);
#endif
        if (const char* e = dev_env("VELO_GRAPHS")) c->use_graphs = atoi(e) != 0;
        if (const char* e = dev_env("VELO_XCD_MAP")) c->xcd_map = atoi(e);
        if (const char* e = dev_env("VELO_TUBE_MAP")) c->tube_map = atoi(e);
        if (const char* e = dev_env("VELO_WARM_START")) c->warm_start = atoi(e);
        if (const char* e = dev_env("VELO_DIMG_SEEDS")) c->dimg_seeds = atoi(e);
        if (const char* e = dev_env("VELO_XCD_CHUNKS")) c->xcd_chunks = atoi(e);
        if (const char* e = dev_env("VELO_CU_MASK")) c->cu_mask_mode = atoi(e);
        if (const char* e = dev_env("VELO_SMALL_SOLVE")) c->small_solve = atoi(e);
        if (const char* e = dev_env("VELO_LM_MERGED")) c->lm_merged = atoi(e);
        if (const char* e = dev_env("VELO_LM_FUSED")) c->lm_fused = atoi(e);
        if (const char* e = dev_env("VELO_LM_ITER")) c->lm_iter = atoi(e);
        if (const char* e = dev_env("VELO_LM_VIS_MERGED")) c->lm_vis_merged = atoi(e);
        if (const char* e = dev_env("VELO_ASSOC_LANE")) c->assoc_lane = atoi(e);
        if (const char* e = dev_env("VELO_LM_MERGED_VIS")) c->lm_trace_vis_off = atoi(e) == 0;
        if (const char* e = dev_env("VELO_ASKER_QUEUE")) c->asker_queue = atoi(e);
        if (const char* e = dev_env("VELO_DENSE_ROWS")) c->dense_rows = std::min(std::max(atoi(e), 0), 0xfffff);
        if (const char* e = dev_env("VELO_ASK_MAP")) c->ask_map = atoi(e) != 0 ? 1 : 0;
        if (const char* e = dev_env("VELO_DENSE_FAR")) c->dense_far = std::min(std::max(atoi(e), 0), 64);
        if (const char* e = dev_env("VELO_DENSE_BATCH")) c->dense_batch = atoi(e);
        if (const char* e = dev_env("VELO_ASSOC_DIRECT_MAX")) c->direct_max = std::max(atoi(e), 0);
        if (const char* e = dev_env("VELO_PATCH_ORDER")) c->patch_order = atoi(e);
        if (const char* e = dev_env("VELO_PATCH_SHAPE")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 1 && b >= 1) { c->patch_rings = a; c->patch_len = b; } }
        if (const char* e = dev_env("VELO_ASSOC_DIRECT_SKIP")) c->direct_skip = std::max(atoi(e), 1);
        if (const char* e = getenv("VELO_CHAIN")) c->chain = atoi(e);
        if (const char* e = getenv("VELO_CHAIN_MARGIN")) { c->chain_margin = std::max(atoi(e), 0); c->chain_margin_fixed = true; }
        if (const char* e = dev_env("VELO_ASSOC_LDS_PAD")) { c->assoc_lds_pad = std::max(atoi(e), 0); c->assoc_lds_pad_fixed = true; }
        if (const char* e = dev_env("VELO_LM_LEAN")) c->lm_lean = atoi(e);
        if (const char* e = dev_env("VELO_LM_PERSIST")) c->lm_persist = atoi(e);
        if (const char* e = dev_env("VELO_LM_PERSIST_WGS")) c->lm_persist_wgs = std::max(atoi(e), 0);
        if (const char* e = dev_env("VELO_ASKER_ROWS")) c->asker_rows = atoi(e);
        if (const char* e = dev_env("VELO_PERSISTENT_WGS")) c->persistent_wgs = std::max(atoi(e), 1);
        if (const char* e = getenv("VELO_BATCH_LOCKSTEP")) c->batch_lockstep = atoi(e);
        if (c->cu_mask_mode > 0) {
            static std::atomic<int> seq{0};
            const int q = (seq.fetch_add(1) / 2) % 4;
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 256; i++) {
                const bool mine = c->cu_mask_mode == 1 ? (i / 64 == q) : ((i % 8) / 2 == q);
                if (mine) mask[i / 32] |= 1u << (i % 32);
            }
            HIP_TRY(hipExtStreamCreateWithCUMask(&c->stream, 8, mask));
        } else
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_TRY(hipHostMalloc((void**)&c->h_status, sizeof(HostStatus), hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&c->h_x, sizeof(double) * 64, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&c->h_int, sizeof(int) * 32, hipHostMallocDefault));   // [0] counters, [8..13] target box keys, [16..21] source box keys
        VELO_TRY(c->state.reserve(2));                       // [1]: the other half of the one-launch iteration's double buffer
        VELO_TRY(c->eval_pt.reserve(1));
        VELO_TRY(c->pose_rec.reserve(1)); VELO_TRY(c->solve_log.reserve(VELO_MAX_SOLVES)); VELO_TRY(c->chain_fail.reserve(1));
        HIP_TRY(hipMemsetAsync(c->chain_fail.p, 0, sizeof(int), c->stream));
        HIP_TRY(hipMemsetAsync(c->pose_rec.p, 0, sizeof(PoseRecord), c->stream));
        HIP_TRY(hipHostMalloc((void**)&c->h_log, sizeof(SolveLog) * VELO_MAX_SOLVES + 64, hipHostMallocDefault));
        VELO_TRY(c->partials.reserve((size_t)2 * (kMaxEvalBlocks + kMaxVisBlocks) * kNumAcc));   // two halves, same reason
        VELO_TRY(c->reduced.reserve(2 * kNumAcc));
        VELO_TRY(c->xdev.reserve(8));
        VELO_TRY(c->ticket.reserve(1));
        HIP_TRY(hipMemsetAsync(c->ticket.p, 0, sizeof(int), c->stream));
        VELO_TRY(c->bbox_keys.reserve(6));
        VELO_TRY(c->n_valid.reserve(2));
        VELO_TRY(c->dbg.reserve(8));
        HIP_TRY(hipMemsetAsync(c->dbg.p, 0, 64, c->stream));
        HIP_TRY(hipMemsetAsync(c->state.p, 0, 2 * sizeof(LMState), c->stream));
        HIP_TRY(hipEventCreate(&c->ev0));
        HIP_TRY(hipEventCreate(&c->ev1));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return VELO_OK;
    };
    const int st = init();
    if (st != VELO_OK) { const std::string keep = g_err; velo_destroy(c); g_err = keep; return st; }
    *out = c;
    return VELO_OK;
}

int velo_destroy(velo_ctx* c) {
    if (!c) return VELO_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if ((c->debug_skip & 32) && c->wg_times.p && c->wg_times_n > 0) {
        std::vector<unsigned long long> h((size_t)16 * c->wg_times_n);
        if (hipMemcpy(h.data(), c->wg_times.p, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long t0 = ~0ull, t1 = 0; std::vector<double> dur, st;
            for (int i = 0; i < c->wg_times_n; i++) { t0 = std::min(t0, h[2 * i]); t1 = std::max(t1, h[2 * i + 1]); }
            for (int i = 0; i < c->wg_times_n; i++) { dur.push_back((h[2 * i + 1] - h[2 * i]) * 0.01); st.push_back((h[2 * i] - t0) * 0.01); }
            std::sort(dur.begin(), dur.end()); std::sort(st.begin(), st.end());
            auto pc = [&](std::vector<double>& v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
            double mean = 0; for (double d : dur) mean += d; mean /= dur.size();
            {   // the five slowest groups: which queries are they?
                std::vector<std::pair<double, int>> slow;
                for (int i = 0; i < c->wg_times_n; i++) slow.emplace_back((h[2 * i + 1] - h[2 * i]) * 0.01, i);
                std::sort(slow.begin(), slow.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first > y.first; });
                for (int k = 0; k < 5 && k < (int)slow.size(); k++) fprintf(stderr, "[velo dbg]   slow group %d: %.1f us (queries %d..%d of the list)\n", slow[(size_t)k].second, slow[(size_t)k].first, slow[(size_t)k].second * 64, slow[(size_t)k].second * 64 + 63),
                    [&](const unsigned long long* g) {
                        fprintf(stderr, "[velo dbg]     clusters %llu chunks %llu candidates %llu askers %llu asker-candidates %llu asker-time %.1f us | wave-0 kcycles: setup %.1f boxes+rows %.1f runlist %.1f stage %.1f sweep %.1f sweepbar %.1f merge+askers %.1f finish %.1f\n",
                                g[0], g[1], g[2], g[3], g[4], g[5] * 0.01, g[6] * 1e-3, g[7] * 1e-3, g[8] * 1e-3, g[9] * 1e-3, g[10] * 1e-3, g[11] * 1e-3, g[12] * 1e-3, g[13] * 1e-3);
                    }(h.data() + 2 * (size_t)c->wg_times_n + 14 * (size_t)slow[(size_t)k].second);
            }
            fprintf(stderr, "[velo dbg] last assoc launch: %d WGs, span %.1f us | WG duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | WG start us: p50 %.1f p90 %.1f p99 %.1f max %.1f\n",
                    c->wg_times_n, (t1 - t0) * 0.01, mean, pc(dur, .5), pc(dur, .9), pc(dur, .99), dur.back(), pc(st, .5), pc(st, .9), pc(st, .99), st.back());
        }
    }
    if ((c->debug_skip & 24) && c->dbg.p) {
        unsigned long long h[8];
        if (hipMemcpy(h, c->dbg.p, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[velo dbg] wave-0 cycles: setup %llu cluster %llu runlist %llu stage %llu sweep %llu sweepbar %llu merge %llu finish %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        }
    }
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    for (int r = 0; r < kMaxPeers; r++) if (c->peer_mapped[r]) { (void)hipIpcCloseMemHandle(c->peer_mapped[r]); c->peer_mapped[r] = nullptr; }
    for (int r = 0; r < kMaxPeers; r++) if (c->peer_area_mapped[r]) { (void)hipIpcCloseMemHandle(c->peer_area_mapped[r]); c->peer_area_mapped[r] = nullptr; }
    if (c->peer_slab) { (void)hipFree(c->peer_slab); c->peer_slab = nullptr; }
    for (void* p : c->peer_retired) (void)hipFree(p);
    c->peer_retired.clear();
    if (c->h_agree) { (void)hipHostFree(c->h_agree); c->h_agree = nullptr; }
    if (c->peer_area) { (void)hipFree(c->peer_area); c->peer_area = nullptr; }
    c->T.reset();                                            // the target goes with its last holder
    c->vis_counts.release(); c->lb_status.release(); c->scan_tiles.release(); c->cursor.release(); c->scan_total.release(); c->bbox_keys.release();
    c->src.release(); c->src_off.release(); c->q_off.release(); c->q_src.release(); c->staging.release();
    c->seg_flag.release(); c->seg_excl.release(); c->seg_ring.release(); c->seg_off.release();
    c->cp.release(); c->cn.release(); c->cv0.release(); c->aux0.release(); c->aux1.release(); c->n_valid.release(); c->dbg.release(); c->wg_times.release(); c->items.release(); c->item_counters.release(); c->qpos.release(); c->partials_rec.release(); c->partials_all.release();
    c->vm.release(); c->vflags.release();
    for (int k = 0; k < 2; k++) if (c->chunk_graph[k]) (void)hipGraphExecDestroy(c->chunk_graph[k]);
    c->state.release(); c->partials.release(); c->reduced.release(); c->xdev.release(); c->ticket.release();
    c->row_off_vis.release(); c->row_off_icp.release(); c->rows_r.release(); c->rows_J.release();
    if (c->h_batch) (void)hipHostFree(c->h_batch);
    c->batch_items.release(); c->batch_states.release(); c->batch_x.release();
    c->ask_count.release(); c->ask_list.release(); c->ask_keys.release(); c->ask_rings.release();
    c->solve_ctl.release(); c->ag_ctl.release();
    c->pf.land[0].release(); c->pf.land[1].release(); c->nf.undo_cloud.release();
    for (int k = 0; k < 2; k++) if (c->pf.pin[k]) { (void)hipHostFree(c->pf.pin[k]); c->pf.pin[k] = nullptr; c->pf.pin_cap[k] = 0; }
    if (c->nf.call_done) { (void)hipEventDestroy(c->nf.call_done); c->nf.call_done = nullptr; }
    if (c->pf.stream) { (void)hipStreamSynchronize(c->pf.stream); (void)hipStreamDestroy(c->pf.stream); c->pf.stream = nullptr; }
    if (c->pf.ev) { (void)hipEventDestroy(c->pf.ev); c->pf.ev = nullptr; }
    c->batch_tickets.release(); c->batch_pose.release(); c->batch_logs.release(); c->batch_fail.release(); c->pose_rec.release(); c->solve_log.release(); c->chain_fail.release();
    if (c->h_log) (void)hipHostFree(c->h_log);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_x) (void)hipHostFree(c->h_x);
    if (c->h_int) (void)hipHostFree(c->h_int);
    for (auto& ps : c->pin) { if (ps.ev) (void)hipEventDestroy(ps.ev); if (ps.p) (void)hipHostFree(ps.p); }
    if (c->src_bbox_ev) (void)hipEventDestroy(c->src_bbox_ev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto& e : c->assoc_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : c->klog) { if (e.a) (void)hipEventDestroy(e.a); if (e.b) (void)hipEventDestroy(e.b); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return VELO_OK;
}

int velo_set_params(velo_ctx* c, const velo_params* p) {
    if (!c || !p) return fail(VELO_ERR_INVALID, "null argument");
    if (p->icp_skip < 1 || p->f2f_iterations < 0 || p->icp_iterations < 0 || p->max_num_iterations < 0)
        return fail(VELO_ERR_INVALID, "icp_skip must be >= 1 and iteration counts >= 0");
    if (p->f2f_iterations * std::max(p->icp_iterations, 1) > VELO_MAX_SOLVES)
        return fail(VELO_ERR_INVALID, "more than %d solves per call", VELO_MAX_SOLVES);
    HIP_TRY(hipSetDevice(c->device));
    const bool gates_changed = p->correspondence_thresh_icp != c->P.correspondence_thresh_icp || p->f2f_iterations != c->P.f2f_iterations;
    const bool queries_changed = p->icp_skip != c->P.icp_skip || p->enable_icp != c->P.enable_icp;
    if (gates_changed && c->have_target && c->T.use_count() > 1)
        return fail(VELO_ERR_STATE, "the target is shared with other contexts: its index cannot be rebuilt for new gates here; load the target again");
    c->P = *p;
    if (gates_changed && c->have_target) VELO_TRY(build_grids(c));
    if (queries_changed && c->have_source) VELO_TRY(build_query_list(c));
    return VELO_OK;
}

int velo_get_params(const velo_ctx* c, velo_params* p) {
    if (!c || !p) return fail(VELO_ERR_INVALID, "null argument");
    *p = c->P;
    return VELO_OK;
}

int velo_set_timing(velo_ctx* c, int enable) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    c->timing = enable < 0 ? 0 : (enable > 3 ? 3 : enable);
    return VELO_OK;
}

int velo_set_residual_stats(velo_ctx* c, int enable) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    c->want_stats = enable != 0;
    return VELO_OK;
}

int velo_set_target(velo_ctx* c, const float* xyz, int64_t stride, const int32_t* off, int32_t n_rings, int on_device) {
    return velo_set_target_part(c, xyz, stride, off, n_rings, 0, 0, on_device);
}

static int set_target_begin(velo_ctx* c, const float* xyz, int64_t stride, const int32_t* off, int32_t n_rings, int32_t first_ring, int32_t first_point, int on_device) {
    if (!c || !off || n_rings < 0 || first_ring < 0 || first_point < 0) return fail(VELO_ERR_INVALID, "null/negative argument");
    if (stride < 12) return fail(VELO_ERR_INVALID, "stride_bytes must be >= 12");
    if (off[0] != 0) return fail(VELO_ERR_INVALID, "ring_offsets[0] must be 0");
    for (int r = 0; r < n_rings; r++) {
        // an empty ring makes pcl::KdTreeFLANN::setInputCloud fail in the reference; reject it loudly
        if (off[r + 1] <= off[r]) return fail(VELO_ERR_INVALID, "target ring %d is empty or offsets are not increasing", r);
    }
    const int n = n_rings > 0 ? off[n_rings] : 0;
    if (n > 0 && !xyz) return fail(VELO_ERR_INVALID, "null xyz");
    HIP_TRY(hipSetDevice(c->device));
    own_target(c);
    c->have_target = false; c->have_corr = false; c->have_partials = false;
    c->T->n_tgt = n; c->T->n_tgt_rings = n_rings;
    c->T->tgt_first_ring = first_ring; c->T->tgt_first_point = first_point;
    c->T->h_tgt_off.assign(off, off + n_rings + 1);
    return target_ingest(c, xyz, stride, on_device);
}
int velo_set_target_part(velo_ctx* c, const float* xyz, int64_t stride, const int32_t* off, int32_t n_rings, int32_t first_ring, int32_t first_point, int on_device) {
    VELO_TRY(set_target_begin(c, xyz, stride, off, n_rings, first_ring, first_point, on_device));
    return target_finalize_end(c);
}

static int set_source_begin(velo_ctx* c, const float* xyz, int64_t stride, const int32_t* off, int32_t n_rings, int on_device) {
    if (c) { c->src_raw.on = false; c->src_bbox_valid = false; }
    if (!c || !off || n_rings < 0) return fail(VELO_ERR_INVALID, "null/negative argument");
    if (stride < 12) return fail(VELO_ERR_INVALID, "stride_bytes must be >= 12");
    if (off[0] != 0) return fail(VELO_ERR_INVALID, "ring_offsets[0] must be 0");
    for (int r = 0; r < n_rings; r++) if (off[r + 1] < off[r]) return fail(VELO_ERR_INVALID, "source ring offsets decrease at ring %d", r);
    const int n = n_rings > 0 ? off[n_rings] : 0;
    if (n > 0 && !xyz) return fail(VELO_ERR_INVALID, "null xyz");
    HIP_TRY(hipSetDevice(c->device));
    c->have_source = false; c->have_corr = false;
    c->n_src = n; c->n_src_rings = n_rings;
    c->h_src_off.assign(off, off + n_rings + 1);
    // (the packed copy is written by the launch that also lays out the query list: source_finalize -> source_ingest)
    VELO_TRY(c->src.reserve((size_t)std::max(n, 1)));
    const char* dsrc = (const char*)xyz;
    if (!on_device && n > 0) {
        const size_t bytes = (size_t)(n - 1) * (size_t)stride + 12;
        if (c->pf.ready && c->pf.host == (const void*)xyz && c->pf.bytes == bytes) {
            // this cloud was announced one call ago (velo_hint_next_source) and is on the device already: the ingest waits for its copy's event
            if (c->pf.in_pin) dsrc = c->pf.pin[c->pf.buf];                // (page-locked host memory: the ingest launch reads it over the bus)
            else { HIP_TRY(hipStreamWaitEvent(c->stream, c->pf.ev, 0)); dsrc = c->pf.land[c->pf.buf].p; }
        } else {
            VELO_TRY(c->staging.reserve(bytes));
            HIP_TRY(hipMemcpyAsync(c->staging.p, xyz, bytes, hipMemcpyHostToDevice, c->stream));
            dsrc = c->staging.p;
        }
    }
    c->pf.ready = false;                                                  // (a hint is good for the very next source only)
    c->src_raw.dsrc = dsrc; c->src_raw.stride = stride; c->src_raw.on = n > 0;
    return VELO_OK;
}
int velo_set_source(velo_ctx* c, const float* xyz, int64_t stride, const int32_t* off, int32_t n_rings, int on_device) {
    VELO_TRY(set_source_begin(c, xyz, stride, off, n_rings, on_device));
    { const int st = source_finalize(c); c->src_raw.on = false; if (st != VELO_OK) return st; }
    HIP_TRY(hipStreamSynchronize(c->stream));                             // the caller's buffer has been read when the call returns
    return VELO_OK;
}

// kitti.h:121-185 on the device ("next" row 1 of SURVEY.md 8(f)): raw Velodyne records (x, y, z, reflectance; any stride
// >= 12) in file order -> camera-0-frame rings, loaded straight into this context as its source or target.
int velo_set_scan_velodyne(velo_ctx* c, int32_t as_target, const float* xyzr, int64_t stride, int32_t n, const float velo_to_cam[16], int on_device) {
    if (!c || n < 0 || (n > 0 && !xyzr) || !velo_to_cam) return fail(VELO_ERR_INVALID, "null/negative argument");
    if (stride < 12) return fail(VELO_ERR_INVALID, "stride_bytes must be >= 12");
    HIP_TRY(hipSetDevice(c->device));
    if (as_target) own_target(c);
    DevBuf<float4>& dst = as_target ? c->T->tgt : c->src;
    std::vector<int>& h_off = as_target ? c->T->h_tgt_off : c->h_src_off;
    if (as_target) { c->have_target = false; c->have_partials = false; c->T->tgt_first_ring = 0; c->T->tgt_first_point = 0; } else { c->have_source = false; c->src_bbox_valid = false; }
    c->have_corr = false;
    VELO_TRY(dst.reserve((size_t)std::max(n, 1)));
    int n_rings = 0;
    h_off.assign(1, 0);
    if (n > 0) {
        const char* rec = (const char*)xyzr;
        if (!on_device) {
            const size_t bytes = (size_t)(n - 1) * (size_t)stride + 12;
            VELO_TRY(c->staging.reserve(bytes));
            HIP_TRY(hipMemcpyAsync(c->staging.p, xyzr, bytes, hipMemcpyHostToDevice, c->stream));
            rec = c->staging.p;
        }
        VELO_TRY(c->seg_flag.reserve((size_t)n + 2)); VELO_TRY(c->seg_excl.reserve((size_t)n + 2));
        VELO_TRY(c->seg_ring.reserve((size_t)n + 2)); VELO_TRY(c->seg_off.reserve((size_t)n + 2));
        VELO_TRY(c->cursor.reserve((size_t)n + 2));
        const int n_tiles = cdiv(n, kScanTile);
        VELO_TRY(c->scan_tiles.reserve((size_t)n_tiles + 1));
        VELO_TRY(c->scan_total.reserve(1));
        hipLaunchKernelGGL(ring_break_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, rec, stride, n, c->seg_flag.p);
        HIP_TRY(hipMemcpyAsync(c->seg_excl.p, c->seg_flag.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(scan_tiles_kernel, dim3(n_tiles), dim3(kScanThreads), 0, c->stream, c->seg_excl.p, n, c->scan_tiles.p);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanThreads), 0, c->stream, c->scan_tiles.p, n_tiles, c->scan_total.p);
        hipLaunchKernelGGL(scan_add_kernel, dim3(cdiv(n + 1, 256)), dim3(256), 0, c->stream, c->seg_excl.p, n, c->scan_tiles.p, c->scan_total.p, c->cursor.p);
        hipLaunchKernelGGL(ring_offsets_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const int*)c->seg_excl.p, (const int*)c->seg_flag.p, n,
                           c->seg_ring.p, c->seg_off.p, c->scan_total.p);
        Mat34f M;
        for (int k = 0; k < 12; k++) M.m[k] = velo_to_cam[k];       // rows 0..2 of the row-major 4x4
        hipLaunchKernelGGL(ring_reorder_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, rec, stride, n, (const int*)c->seg_ring.p, (const int*)c->seg_off.p, M, dst.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->h_int, c->scan_total.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        n_rings = c->h_int[0];
        h_off.resize((size_t)n_rings + 1);
        HIP_TRY(hipMemcpy(h_off.data(), c->seg_off.p, sizeof(int) * ((size_t)n_rings + 1), hipMemcpyDeviceToHost));
    }
    if (as_target) { c->T->n_tgt = n; c->T->n_tgt_rings = n_rings; return target_finalize(c); }
    c->n_src = n; c->n_src_rings = n_rings;
    c->src_raw.on = false; return source_finalize(c);          // (the cloud is packed already)
}

int velo_share_target(velo_ctx* dst, velo_ctx* src) {
    if (!dst || !src) return fail(VELO_ERR_INVALID, "null ctx");
    if (dst == src) return VELO_OK;
    if (dst->device != src->device) return fail(VELO_ERR_INVALID, "contexts on different devices (%d, %d)", dst->device, src->device);
    if (!src->have_target) return fail(VELO_ERR_STATE, "the source context holds no target");
    double g1 = gate_of_iter(dst->P, 1), g2 = gate_of_iter(src->P, 1);
    for (int it = 2; it <= dst->P.f2f_iterations; it++) g1 = std::min(g1, gate_of_iter(dst->P, it));
    for (int it = 2; it <= src->P.f2f_iterations; it++) g2 = std::min(g2, gate_of_iter(src->P, it));
    if (g1 != g2) return fail(VELO_ERR_INVALID, "the contexts work with different gates: the index of one does not serve the other");
    HIP_TRY(hipSetDevice(src->device));
    HIP_TRY(hipStreamSynchronize(src->stream));                // the index is complete before another stream reads it
    HIP_TRY(hipStreamSynchronize(dst->stream));                // nothing of dst still reads what it is about to drop
    dst->T = src->T;
    dst->have_target = true; dst->have_corr = false; dst->have_partials = false;
    dst->prev_ready = false;
    return VELO_OK;
}

// the promotion in two halves (like a target load): swap + the fused ingest launch on the packed records, IN PLACE (record i -> tgt[i],
// every thread reads its own record before it writes it; pack of a packed record is the identity), then the index once the box is known
static int promote_begin(velo_ctx* c) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (!c->have_source) return fail(VELO_ERR_STATE, "no source cloud to promote");
    HIP_TRY(hipSetDevice(c->device));
    own_target(c);
    std::swap(c->T->tgt.p, c->src.p); std::swap(c->T->tgt.cap, c->src.cap);
    c->T->h_tgt_off = c->h_src_off;
    c->T->n_tgt = c->n_src; c->T->n_tgt_rings = c->n_src_rings;
    c->T->tgt_first_ring = 0; c->T->tgt_first_point = 0;
    c->have_source = false; c->have_target = false; c->have_corr = false; c->have_partials = false;
    c->n_src = 0; c->n_src_rings = 0; c->n_q = 0; c->h_src_off.assign(1, 0); c->h_q_off.assign(1, 0);
    for (int r = 0; r < c->T->n_tgt_rings; r++) if (c->T->h_tgt_off[r + 1] <= c->T->h_tgt_off[r]) return fail(VELO_ERR_INVALID, "target ring %d is empty", r);
    // The scan's bounding box came back with the call that loaded it as source (source_ingest_kernel takes it: the same keys
    // target_ingest_kernel computes for the same points), so the grid can be sized and the index build enqueued right behind the ingest
    // launch -- no host wait in the load of a drive's frame.
#ifdef VELO_NO_EARLY_PROMOTE                                           // A/B build: the promotion waits for its own bounding box, as before round 4
    const bool box_known = false;
#else
    const bool box_known = c->src_bbox_valid && c->T->n_tgt > 0;
#endif
    unsigned keys[6];
    if (box_known) {
        HIP_TRY(hipEventSynchronize(c->src_bbox_ev));                  // passed long ago unless the loading call ended on an error before its synchronisation
        std::memcpy(keys, c->h_int + 16, sizeof(keys));
    }
    c->src_bbox_valid = false;
    VELO_TRY(target_ingest(c, reinterpret_cast<const float*>(c->T->tgt.p), (int64_t)sizeof(float4), 1));
    if (box_known) {
        if (keys[0] == 0xffffffffu) { for (int k = 0; k < 6; k++) c->T->bbox[k] = 0.f; }      // no finite point at all
        else for (int k = 0; k < 6; k++) c->T->bbox[k] = key2f(keys[k]);
        for (Grid& G : c->T->grids) G.built = false;
        VELO_TRY(build_grids(c));
        c->have_target = true;
        c->target_early = true;
    }
    return VELO_OK;
}
int velo_source_to_target(velo_ctx* c) {
    VELO_TRY(promote_begin(c));
    return target_finalize_end(c);
}

// ---- device-resident scan cache (lru.h:31-61) -------------------------------------------------------------------------------
struct CachedScan {
    int frame = 0;
    int n = 0, n_rings = 0;
    DevBuf<float4> cloud;
    std::vector<int> h_off;
    // target side only: what target_finalize builds
    bool has_index = false;
    DevBuf<int> ring_of;
    float bbox[6] = {0, 0, 0, 0, 0, 0};
    Grid grid;
};
struct velo_scan_cache {
    int device = 0;
    int capacity = 50;                                   // lru.h:33
    std::list<CachedScan> times;                         // front = most recently used (lru.h:34)
    std::unordered_map<int, std::list<CachedScan>::iterator> exists;   // lru.h:35
};

int velo_cache_create(velo_scan_cache** out, int32_t device, int32_t capacity) {
    if (!out || capacity < 1) return fail(VELO_ERR_INVALID, "bad cache arguments");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(VELO_ERR_NODEVICE, "no HIP device visible: the scan cache lives in device memory");
    if (device < 0 || device >= n_dev) return fail(VELO_ERR_INVALID, "device %d out of range (%d visible)", device, n_dev);
    velo_scan_cache* k = new velo_scan_cache();
    k->device = device; k->capacity = capacity;
    *out = k;
    return VELO_OK;
}

int velo_cache_destroy(velo_scan_cache* k) {
    if (!k) return VELO_OK;
    (void)hipSetDevice(k->device);
    delete k;                                            // DevBuf members release their memory
    return VELO_OK;
}

int velo_cache_contains(const velo_scan_cache* k, int32_t frame) { return (k && k->exists.count(frame)) ? 1 : 0; }

int velo_cache_frames(const velo_scan_cache* k, int32_t* frames_out, int32_t capacity) {
    if (!k) return 0;
    int i = 0;
    for (const CachedScan& e : k->times) { if (frames_out && i < capacity) frames_out[i] = e.frame; i++; }
    return i;
}

int velo_cache_store(velo_scan_cache* k, int32_t frame, velo_ctx* c, int32_t of_target) {
    if (!k || !c) return fail(VELO_ERR_INVALID, "null argument");
    if (c->device != k->device) return fail(VELO_ERR_INVALID, "context on device %d, cache on device %d", c->device, k->device);
    if (of_target ? !c->have_target : !c->have_source) return fail(VELO_ERR_STATE, "the context holds no %s scan", of_target ? "target" : "source");
    if (of_target && (c->T->tgt_first_ring != 0 || c->T->tgt_first_point != 0)) return fail(VELO_ERR_STATE, "a target shard is not a whole scan");
    HIP_TRY(hipSetDevice(k->device));
    // The node that takes the scan: the frame's own older copy (replaced, not duplicated), else -- when the cache is full -- the
    // least recently used one (lru.h:52-57: it would be dropped anyway), else a new one.  A recycled node keeps its device
    // buffers, so a cache in steady state stores without allocating.
    auto it = k->exists.find(frame);
    if (it != k->exists.end()) {
        k->times.splice(k->times.begin(), k->times, it->second);
    } else if ((int)k->times.size() >= k->capacity) {
        k->exists.erase(k->times.back().frame);
        k->times.splice(k->times.begin(), k->times, std::prev(k->times.end()));
    } else {
        k->times.emplace_front();
    }
    CachedScan& e = k->times.front();
    k->exists[frame] = k->times.begin();
    e.frame = frame;
    e.has_index = false;
    e.n = of_target ? c->T->n_tgt : c->n_src;
    e.n_rings = of_target ? c->T->n_tgt_rings : c->n_src_rings;
    e.h_off = of_target ? c->T->h_tgt_off : c->h_src_off;
    int st = e.cloud.reserve((size_t)std::max(e.n, 1));
    hipError_t he = hipSuccess;
    if (st == VELO_OK && e.n > 0) he = hipMemcpyAsync(e.cloud.p, of_target ? c->T->tgt.p : c->src.p, sizeof(float4) * (size_t)e.n, hipMemcpyDeviceToDevice, c->stream);
    Grid* G = of_target ? grid_for_iter(c, 1) : nullptr;
    if (st == VELO_OK && he == hipSuccess && G) {
        e.has_index = true;
        std::memcpy(e.bbox, c->T->bbox, sizeof(e.bbox));
        e.grid.d = G->d; e.grid.gate = G->gate; e.grid.h = G->h; e.grid.built = true;
        e.grid.wpr = G->wpr; e.grid.n_points_cap = G->n_points_cap;
        const size_t nc = G->table_len(), ns = (size_t)e.n + kGridPad, nw = G->wpr > 0 ? G->n_words() + 1 : 0;
        if ((st = e.ring_of.reserve((size_t)std::max(e.n, 1))) == VELO_OK && (st = e.grid.cell_start.reserve(nc + 7)) == VELO_OK &&
            (st = e.grid.sorted.reserve(ns)) == VELO_OK && (st = e.grid.sring.reserve(ns)) == VELO_OK &&
            (nw == 0 || ((st = e.grid.wmask.reserve(nw)) == VELO_OK && (st = e.grid.wprefix.reserve(nw + 3)) == VELO_OK))) {
            if (nw > 0) he = hipMemcpyAsync(e.grid.wmask.p, G->wmask.p, sizeof(unsigned long long) * nw, hipMemcpyDeviceToDevice, c->stream);
            if (nw > 0 && he == hipSuccess) he = hipMemcpyAsync(e.grid.wprefix.p, G->wprefix.p, sizeof(int) * nw, hipMemcpyDeviceToDevice, c->stream);
            if (e.n > 0 && he == hipSuccess) he = hipMemcpyAsync(e.ring_of.p, c->T->tgt_ring_of.p, sizeof(int) * (size_t)e.n, hipMemcpyDeviceToDevice, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(e.grid.table(), G->table(), sizeof(int) * nc, hipMemcpyDeviceToDevice, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(e.grid.sorted.p, G->sorted.p, sizeof(float4) * ns, hipMemcpyDeviceToDevice, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(e.grid.sring.p, G->sring.p, sizeof(int) * ns, hipMemcpyDeviceToDevice, c->stream);
        }
    }
    if (st == VELO_OK && he == hipSuccess) he = hipStreamSynchronize(c->stream);           // the entry is complete when the call returns
    if (st != VELO_OK || he != hipSuccess) {
        k->exists.erase(frame); k->times.pop_front();
        return st != VELO_OK ? st : fail(VELO_ERR_HIP, "scan cache copy failed: %s", hipGetErrorString(he));
    }
    return VELO_OK;
}

int velo_cache_load(velo_scan_cache* k, int32_t frame, velo_ctx* c, int32_t as_target) {
    if (!k || !c) return fail(VELO_ERR_INVALID, "null argument");
    if (c->device != k->device) return fail(VELO_ERR_INVALID, "context on device %d, cache on device %d", c->device, k->device);
    auto it = k->exists.find(frame);
    if (it == k->exists.end()) return fail(VELO_ERR_STATE, "frame %d is not in the scan cache", frame);
    k->times.splice(k->times.begin(), k->times, it->second);                               // most recently used (lru.h:42-47)
    const CachedScan& e = k->times.front();
    HIP_TRY(hipSetDevice(k->device));
    if (!as_target) {
        VELO_TRY(c->src.reserve((size_t)std::max(e.n, 1)));
        if (e.n > 0) HIP_TRY(hipMemcpyAsync(c->src.p, e.cloud.p, sizeof(float4) * (size_t)e.n, hipMemcpyDeviceToDevice, c->stream));
        c->h_src_off = e.h_off;
        c->n_src = e.n; c->n_src_rings = e.n_rings;
        c->have_source = false; c->src_bbox_valid = false;
        c->src_raw.on = false; return source_finalize(c);          // (the cloud is packed already)
    }
    for (int r = 0; r < e.n_rings; r++) if (e.h_off[(size_t)r + 1] <= e.h_off[(size_t)r]) return fail(VELO_ERR_INVALID, "target ring %d is empty", r);
    own_target(c);
    VELO_TRY(c->T->tgt.reserve((size_t)std::max(e.n, 1)));
    if (e.n > 0) HIP_TRY(hipMemcpyAsync(c->T->tgt.p, e.cloud.p, sizeof(float4) * (size_t)e.n, hipMemcpyDeviceToDevice, c->stream));
    c->T->h_tgt_off = e.h_off;
    c->T->n_tgt = e.n; c->T->n_tgt_rings = e.n_rings;
    c->T->tgt_first_ring = 0; c->T->tgt_first_point = 0;
    c->have_target = false; c->have_corr = false; c->have_partials = false;
    // the cached index serves when it was built for the gates this context works with (same cell size rule, same cloud)
    double gmin = gate_of_iter(c->P, 1);
    for (int iter = 2; iter <= c->P.f2f_iterations; iter++) gmin = std::min(gmin, gate_of_iter(c->P, iter));
    if (const char* env = dev_env("VELO_GRID_GATE")) gmin = atof(env);
    if (!e.has_index || e.grid.gate != gmin) return target_finalize(c);
    c->prev_ready = false;
    VELO_TRY(c->T->tgt_off.reserve((size_t)e.n_rings + 1));
    VELO_TRY(c->T->tgt_ring_of.reserve((size_t)std::max(e.n, 1)));
    VELO_TRY(c->T->tgt_cell_of.reserve((size_t)std::max(e.n, 1)));                            // scratch of a later rebuild (velo_set_params)
    if (c->T->grids.empty()) c->T->grids.resize(1);
    Grid& G = c->T->grids[0];
    const size_t nc = e.grid.table_len(), ns = (size_t)e.n + kGridPad, nw = e.grid.wpr > 0 ? e.grid.n_words() + 1 : 0;
    VELO_TRY(G.cell_start.reserve(nc + 7)); VELO_TRY(G.sorted.reserve(ns)); VELO_TRY(G.sring.reserve(ns));
    if (nw > 0) {
        VELO_TRY(G.wmask.reserve(nw)); VELO_TRY(G.wprefix.reserve(nw + 3));
        HIP_TRY(hipMemcpyAsync(G.wmask.p, e.grid.wmask.p, sizeof(unsigned long long) * nw, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(G.wprefix.p, e.grid.wprefix.p, sizeof(int) * nw, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(c->T->tgt_off.p, c->T->h_tgt_off.data(), sizeof(int) * ((size_t)e.n_rings + 1), hipMemcpyHostToDevice, c->stream));
    if (e.n > 0) HIP_TRY(hipMemcpyAsync(c->T->tgt_ring_of.p, e.ring_of.p, sizeof(int) * (size_t)e.n, hipMemcpyDeviceToDevice, c->stream));
    VELO_TRY(c->T->tgt_pad.reserve((size_t)e.n + 2 * (size_t)e.n_rings + 2));
    if (e.n > 0) {
        hipLaunchKernelGGL(pad_rings_kernel, dim3(cdiv(e.n, 256)), dim3(256), 0, c->stream, (const float4*)c->T->tgt.p, (const int*)c->T->tgt_off.p, (const int*)c->T->tgt_ring_of.p, e.n, 0, c->T->tgt_pad.p);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(G.table(), e.grid.table(), sizeof(int) * nc, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(G.sorted.p, e.grid.sorted.p, sizeof(float4) * ns, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(G.sring.p, e.grid.sring.p, sizeof(int) * ns, hipMemcpyDeviceToDevice, c->stream));
    G.d = e.grid.d; G.gate = e.grid.gate; G.h = e.grid.h; G.built = true;
    G.wpr = e.grid.wpr; G.n_points_cap = e.grid.n_points_cap;
    std::memcpy(c->T->bbox, e.bbox, sizeof(c->T->bbox));
    VELO_TRY(build_direction_image(c));
    HIP_TRY(hipStreamSynchronize(c->stream));                                              // h_tgt_off (pageable) has been read; the entry may be evicted
    c->have_target = true;
    return VELO_OK;
}

int velo_get_ring_offsets(velo_ctx* c, int32_t of_target, int32_t* out, int32_t capacity, int32_t* n_rings) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (of_target ? !(c->have_target && c->T) : !c->have_source) { if (n_rings) *n_rings = 0; return fail(VELO_ERR_STATE, "the context holds no %s scan", of_target ? "target" : "source"); }
    const std::vector<int>& h = of_target ? c->T->h_tgt_off : c->h_src_off;
    const int nr = h.empty() ? 0 : (int)h.size() - 1;
    if (n_rings) *n_rings = nr;
    if (out) for (int i = 0; i <= nr && i < capacity; i++) out[i] = h[i];
    return VELO_OK;
}

// copies the context's camera-frame cloud back (tests): n points, 3 floats each
int velo_get_cloud(velo_ctx* c, int32_t of_target, float* xyz_out, int32_t capacity_points, int32_t* n_points) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (of_target ? !(c->have_target && c->T) : !c->have_source) { if (n_points) *n_points = 0; return fail(VELO_ERR_STATE, "the context holds no %s scan", of_target ? "target" : "source"); }
    const int n = of_target ? c->T->n_tgt : c->n_src;
    if (n_points) *n_points = n;
    if (!xyz_out || capacity_points <= 0 || n == 0) return VELO_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<float4> h((size_t)n);
    HIP_TRY(hipMemcpy(h.data(), of_target ? c->T->tgt.p : c->src.p, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < capacity_points; i++) { xyz_out[3 * i] = h[i].x; xyz_out[3 * i + 1] = h[i].y; xyz_out[3 * i + 2] = h[i].z; }
    return VELO_OK;
}

static int set_visual_impl(velo_ctx* c, const velo_match* m, int32_t n, bool wait, hipStream_t on = nullptr);
int velo_set_visual(velo_ctx* c, const velo_match* m, int32_t n) { return set_visual_impl(c, m, n, true); }
// wait = false: the copy stays queued on the context's stream (the records were copied into the context first), for callers that
// order the stream against their launches themselves (velo_register_batch_visual: the group driver synchronises the contexts' streams)
// on: the stream the copies are queued on instead of the context's own (the lock-step group's: a context's own stream shares a hardware queue
// with some OTHER group's chain of launches, and a host thread was seen to spend 8.5 ms in this function once in a hundred steps)
static int set_visual_impl(velo_ctx* c, const velo_match* m, int32_t n, bool wait, hipStream_t on) {
    if (!c || n < 0 || (n > 0 && !m)) return fail(VELO_ERR_INVALID, "bad visual arguments");
    const hipStream_t st = on ? on : c->stream;
    static_assert(sizeof(VisualMatch) == sizeof(velo_match), "device/host match layout");
    static const bool slow_trace = dev_env("VELO_SLOW_TRACE") != nullptr;     // dev aid: which host call of this function takes milliseconds once in a hundred steps?
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!slow_trace) return;
        const auto t = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t - t_last).count();
        if (us > 500.0) fprintf(stderr, "[velo slow] set_visual: %s took %.0f us\n", what, us);
        t_last = t;
    };
    HIP_TRY(hipSetDevice(c->device));
    lap("hipSetDevice");
    c->n_matches = n;
    c->h_matches.assign(m, m + n);
    c->vflags_valid = false;
    c->h_vflags.clear();
    lap("host copy of the records");
    if (n > 0) {
        VELO_TRY(c->vm.reserve((size_t)n));
        VELO_TRY(c->vflags.reserve((size_t)3 * n));
        lap("reserve");
        if (wait) {
            HIP_TRY(hipMemcpyAsync(c->vm.p, c->h_matches.data(), sizeof(velo_match) * (size_t)n, hipMemcpyHostToDevice, st));
        } else {                                                      // through a pinned slot: the copy is really asynchronous
            int* pin = nullptr;
            VELO_TRY(pin_acquire(c, 3, (sizeof(velo_match) * (size_t)n + sizeof(int) - 1) / sizeof(int), &pin));
            lap("pin_acquire");
            std::memcpy(pin, m, sizeof(velo_match) * (size_t)n);
            lap("memcpy into the pinned slot");
            static_assert(sizeof(velo_match) % sizeof(int) == 0, "records are copied word by word");
            const int n_words = (int)(sizeof(velo_match) / sizeof(int)) * n;
            hipLaunchKernelGGL(upload_words_kernel, dim3(cdiv(n_words, 256)), dim3(256), 0, st, (const int*)pin, reinterpret_cast<int*>(c->vm.p), n_words);
            HIP_TRY(hipGetLastError());
            lap("upload launch");
            HIP_TRY(hipEventRecord(c->pin[3].ev, st));
            c->pin[3].pending = true;
            lap("hipEventRecord");
        }
        HIP_TRY(hipMemsetAsync(c->vflags.p, 0, (size_t)3 * n, st));
        lap("hipMemsetAsync");
        if (wait) HIP_TRY(hipStreamSynchronize(st));
    }
    return VELO_OK;
}

int velo_associate(velo_ctx* c, const double x[6], int32_t iter, int32_t* n_valid) {
    if (!c || !x) return fail(VELO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if ((c->comm || c->peer_on) && c->target_sharded) {
        VELO_TRY(associate_target_sharded(c, x, iter, true));
        HIP_TRY(hipMemcpyAsync(c->h_int, c->n_valid.p + c->nv_idx, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->last_n_valid = c->h_int[0];
        if (n_valid) *n_valid = c->last_n_valid;
        return peer_check(c);                                  // a timed-out record exchange merged stale areas
    }
    return do_associate(c, x, iter, true, true, n_valid);
}

// merge `world` device-resident partial tables (table w at tables + w * stride, records of queries [qb, qe) in order)
static int launch_merge(velo_ctx* c, const PartialRec* tables, int world, int stride, int iter, bool want_aux) {
    int qb, qe;
    q_range(c, &qb, &qe);
    VELO_TRY(next_valid_counter(c));
    if (qe > qb) {
        AssocOut out;
        out.p = c->cp.p; out.n = c->cn.p; out.v0 = c->cv0.p; out.aux0 = c->aux0.p; out.aux1 = c->aux1.p; out.n_valid = c->n_valid.p + c->nv_idx;
        out.dbg = c->dbg.p; out.wg_times = nullptr; out.first_ring = c->T->tgt_first_ring; out.first_point = c->T->tgt_first_point; out.partial = nullptr; out.prev_a = nullptr; out.prev_b = nullptr; out.prev_r = nullptr; out.n_valid_next = nullptr;
        out.ask_count = nullptr; out.ask_count_next = nullptr; out.ask_list = nullptr; out.ask_keys = nullptr; out.ask_rings = nullptr;
        const unsigned long long key_inf = ((unsigned long long)gate_bits_of(gate_of_iter(c->P, iter)) + 1ull) << 32;
        hipLaunchKernelGGL(merge_partials_kernel, dim3(cdiv(qe - qb, 256)), dim3(256), 0, c->stream, tables, world, stride, qb, qe,
                           (const float4*)c->src.p, (const int*)c->q_src.p, key_inf, c->P.icp_norm_condition, out, want_aux ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    c->have_corr = true;
    return VELO_OK;
}

// target-sharded association with a communicator: partial search over all queries, all-to-all of the record slices
// (rank r receives, from everybody, the records of ITS query share), merge on the owner
static int associate_target_sharded(velo_ctx* c, const double x[6], int iter, bool want_aux) {
    static_assert(sizeof(PartialRec) == sizeof(velo_partial), "partial record layout");
    VELO_TRY(do_associate(c, x, iter, false, false, nullptr, true));
    const int W = c->shard_world;
    int qb, qe;
    q_range(c, &qb, &qe);
    int max_share = 0;
    for (int r = 0; r < W; r++) max_share = std::max(max_share, (int)((int64_t)c->n_q * (r + 1) / W - (int64_t)c->n_q * r / W));
    if (c->peer_on) {
        if (!c->peer_recs_on) return fail(VELO_ERR_STATE, "target-sharded mode over peers needs velo_comm_peer_attach_records");
        if (c->n_q > c->peer_area_queries) return fail(VELO_ERR_INVALID, "%d queries, the peers' record areas were sized for %d", c->n_q, c->peer_area_queries);
        const unsigned long long seq = ++c->peer_xseq;
        PeerRecs R = c->peer_recs;
        R.max_share = max_share;
        if ((size_t)W * max_share > R.parity_stride) return fail(VELO_ERR_STATE, "record area too small");
        if (c->n_q > 0) hipLaunchKernelGGL(peer_scatter_records_kernel, dim3(cdiv(c->n_q, 256)), dim3(256), 0, c->stream, (const PartialRec*)c->partials_rec.p, c->n_q, R, (int)(seq & 1ull));
        hipLaunchKernelGGL(peer_exchange_sync_kernel, dim3(1), dim3(64), 0, c->stream, c->peer, seq);
        HIP_TRY(hipGetLastError());
        return launch_merge(c, c->peer_area + (size_t)(seq & 1ull) * R.parity_stride, W, max_share, iter, want_aux);
    }
    VELO_TRY(c->partials_all.reserve((size_t)W * std::max(max_share, 1)));
    NCCL_TRY(ncclGroupStart());
    ncclResult_t gr = ncclSuccess;                                      // an error inside the group must still close it
    for (int r = 0; r < W && gr == ncclSuccess; r++) {
        const int rb = (int)((int64_t)c->n_q * r / W), re = (int)((int64_t)c->n_q * (r + 1) / W);
        if (re > rb) gr = ncclSend(c->partials_rec.p + rb, (size_t)(re - rb) * sizeof(PartialRec), ncclChar, r, c->comm, c->stream);
        if (gr == ncclSuccess && qe > qb) gr = ncclRecv(c->partials_all.p + (size_t)r * max_share, (size_t)(qe - qb) * sizeof(PartialRec), ncclChar, r, c->comm, c->stream);
    }
    const ncclResult_t ge = ncclGroupEnd();
    if (gr != ncclSuccess) return fail(VELO_ERR_COMM, "record exchange failed: %s", ncclGetErrorString(gr));
    if (ge != ncclSuccess) return fail(VELO_ERR_COMM, "ncclGroupEnd failed: %s", ncclGetErrorString(ge));
    return launch_merge(c, c->partials_all.p, W, max_share, iter, want_aux);
}

int velo_associate_partial(velo_ctx* c, const double x[6], int32_t iter) {
    if (!c || !x) return fail(VELO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    c->ring_order_forced = true;                                      // the records are an exchange format: one order for every rank
    return do_associate(c, x, iter, false, true, nullptr, true);
}

int velo_get_partials(velo_ctx* c, velo_partial* out, int32_t capacity, int32_t* n_queries) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (n_queries) *n_queries = c->have_partials ? c->n_q : 0;
    if (!out || capacity <= 0) return VELO_OK;
    if (!c->have_partials) return fail(VELO_ERR_STATE, "no partial association has run yet");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->partials_rec.p, sizeof(velo_partial) * (size_t)std::min(capacity, c->n_q), hipMemcpyDeviceToHost));
    return VELO_OK;
}

int velo_merge_partials(velo_ctx* c, const velo_partial* const* tables, int32_t world, int32_t* n_valid) {
    if (!c || !tables || world < 1) return fail(VELO_ERR_INVALID, "bad merge arguments");
    if (!c->have_source) return fail(VELO_ERR_STATE, "merge needs set_source first");
    HIP_TRY(hipSetDevice(c->device));
    c->ring_order_forced = true;
    if (query_list_stale(c)) VELO_TRY(build_query_list(c));
    int qb, qe;
    q_range(c, &qb, &qe);
    const int share = std::max(qe - qb, 1);
    VELO_TRY(c->partials_all.reserve((size_t)world * share));
    for (int w = 0; w < world; w++) {
        if (!tables[w]) return fail(VELO_ERR_INVALID, "null table %d", w);
        if (qe > qb) HIP_TRY(hipMemcpyAsync(c->partials_all.p + (size_t)w * share, tables[w] + qb, sizeof(velo_partial) * (size_t)(qe - qb), hipMemcpyHostToDevice, c->stream));
    }
    VELO_TRY(launch_merge(c, c->partials_all.p, world, share, c->last_partial_iter, true));
    HIP_TRY(hipMemcpyAsync(c->h_int, c->n_valid.p + c->nv_idx, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->last_n_valid = c->h_int[0];
    if (n_valid) *n_valid = c->last_n_valid;
    return VELO_OK;
}

int velo_get_correspondences(velo_ctx* c, velo_corr* out, int32_t capacity, int32_t* n_queries) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    int qb = 0, qe = 0;
    if (c->have_corr) q_range(c, &qb, &qe);
    const int n = qe - qb;
    if (n_queries) *n_queries = n;
    if (!out || capacity <= 0 || n == 0) return VELO_OK;
    if (!c->have_corr) return fail(VELO_ERR_STATE, "no association has run yet");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float4> p(n), nn(n), v0(n), a1(n);
    std::vector<int4> a0(n);
    HIP_TRY(hipMemcpy(p.data(), c->cp.p + qb, sizeof(float4) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nn.data(), c->cn.p + qb, sizeof(float4) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(v0.data(), c->cv0.p + qb, sizeof(float4) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(a0.data(), c->aux0.p + qb, sizeof(int4) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(a1.data(), c->aux1.p + qb, sizeof(float4) * n, hipMemcpyDeviceToHost));
    const int skip = c->src_skip;
    int ring = 0;
    for (int i = 0; i < std::min(n, capacity); i++) {            // record i = query i of the reference's order (ring by ring)
        const int qi = qb + i;
        while (ring + 1 < c->n_src_rings && c->h_q_off[ring + 1] <= qi) ring++;
        while (c->h_q_off[ring + 1] <= qi && ring + 1 < c->n_src_rings) ring++;
        // where the list keeps it (patch order is unsharded only, so qb = 0 there)
        const int t = c->q_patch ? patch_position(c->h_q_off.data(), c->n_src_rings, ring, qi - c->h_q_off[ring], c->patch_rings, c->patch_len) : i;
        velo_corr& o = out[i];
        std::memset(&o, 0, sizeof(o));
        int valid; std::memcpy(&valid, &p[t].w, 4);
        int idx_k; std::memcpy(&idx_k, &a1[t].x, 4);
        o.valid = valid; o.ring_i = a0[t].x; o.idx_i = a0[t].y; o.ring_j = a0[t].z; o.idx_j = a0[t].w; o.idx_k = idx_k;
        o.src_ring = ring; o.src_idx = (qi - c->h_q_off[ring]) * skip;
        o.dist_i = a1[t].y; o.dist_j = a1[t].z;
        o.p[0] = p[t].x; o.p[1] = p[t].y; o.p[2] = p[t].z;
        o.n[0] = nn[t].x; o.n[1] = nn[t].y; o.n[2] = nn[t].z;
        o.v0[0] = v0[t].x; o.v0[1] = v0[t].y; o.v0[2] = v0[t].z;
    }
    return VELO_OK;
}

int velo_build_visual(velo_ctx* c, const double x[6], int32_t iter, int32_t* n_blocks) {
    if (!c || !x) return fail(VELO_ERR_INVALID, "null argument");
    if (iter < 1) return fail(VELO_ERR_INVALID, "iter must be >= 1");
    HIP_TRY(hipSetDevice(c->device));
    return do_build_visual(c, x, false, iter, n_blocks);
}

int velo_get_good_matches(velo_ctx* c, velo_good_match* out, int32_t capacity, int32_t* n) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    int count = 0;
    // emission order of the reference: per match 3D3D|2D2D, 3D2D, 2D3D (velo.h:662-789); matches are cam-major
    for (int i = 0; i < c->n_matches && (size_t)(3 * i + 2) < c->h_vflags.size(); i++) {
        for (int s = 0; s < 3; s++) {
            const unsigned char f = c->h_vflags[3 * i + s];
            if (!f) continue;
            if (out && count < capacity) {
                out[count].cam = c->h_matches[i].cam; out[count].point1 = c->h_matches[i].point1;
                out[count].point2 = c->h_matches[i].point2; out[count].residual_type = f - 1;
            }
            count++;
        }
    }
    if (n) *n = count;
    return VELO_OK;
}

int velo_evaluate(velo_ctx* c, const double x[6], double* cost, double JtJ[36], double Jtr[6]) {
    if (!c || !x) return fail(VELO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    std::memcpy(c->h_x, x, sizeof(double) * 6);
    HIP_TRY(hipMemcpyAsync(c->xdev.p, c->h_x, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
    const EvalArgs A = eval_args(c, c->xdev.p);
    const EvalPlan plan = eval_plan(A);
    const int nblocks = plan.total();
    launch_eval(c, A, plan);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(64), 0, c->stream, (const LMState*)nullptr, (const double*)c->partials.p, nblocks, c->reduced.p);
    HIP_TRY(hipGetLastError());
    double* res = c->reduced.p;
    if (c->peer_on) {
        hipLaunchKernelGGL(peer_reduce_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->partials.p, nblocks, c->peer, c->reduced.p + kNumAcc);
        HIP_TRY(hipGetLastError());
        res = c->reduced.p + kNumAcc;
    } else if (c->comm) {
        NCCL_TRY(ncclAllReduce(c->reduced.p, c->reduced.p + kNumAcc, kNumAcc, ncclDouble, ncclSum, c->comm, c->stream));
        res = c->reduced.p + kNumAcc;
    }
    HIP_TRY(hipMemcpyAsync(c->h_x + 8, res, sizeof(double) * kNumAcc, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    VELO_TRY(peer_check(c));                                   // a timed-out all-reduce summed stale slab contents
    const double* E = c->h_x + 8;
    if (cost) *cost = E[27];
    if (JtJ) {
        int k = 0;
        for (int i = 0; i < 6; i++) for (int j = i; j < 6; j++) { JtJ[i * 6 + j] = E[k]; JtJ[j * 6 + i] = E[k]; k++; }
    }
    if (Jtr) for (int i = 0; i < 6; i++) Jtr[i] = E[21 + i];
    return VELO_OK;
}

int velo_evaluate_rows(velo_ctx* c, const double x[6], double* residuals, double* jacobian, int32_t capacity_rows, int32_t* n_rows) {
    if (!c || !x) return fail(VELO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    // row layout: visual blocks first (match order, slot order), then valid ICP blocks in query order
    int qb = 0, qe = 0;
    if (c->have_corr) q_range(c, &qb, &qe);
    const int nq = qe - qb;
    // this rank's share of the visual matches (the same split as eval_args): rows are numbered within the share
    const int nvm = c->vflags_valid ? c->n_matches : 0, Wv = std::max(c->shard_world, 1);
    const int m0 = (int)((int64_t)nvm * c->shard_rank / Wv), m1 = (int)((int64_t)nvm * (c->shard_rank + 1) / Wv);
    const bool vis = m1 > m0 && c->h_vflags.size() >= (size_t)3 * m1;
    std::vector<int> h_vis((size_t)3 * (vis ? m1 - m0 : 0), -1);
    int rows = 0;
    if (vis) {
        for (size_t s = 0; s < h_vis.size(); s++) {
            const unsigned char f = c->h_vflags[(size_t)3 * m0 + s];
            if (!f) continue;
            h_vis[s] = rows;
            const int t = f - 1;
            rows += (t == VELO_RESIDUAL_3D3D) ? 3 : (t == VELO_RESIDUAL_2D2D) ? 1 : 2;
        }
    }
    std::vector<int> h_icp((size_t)std::max(c->n_q, 1), -1);
    if (nq > 0) {
        std::vector<float4> p(nq);
        HIP_TRY(hipMemcpy(p.data(), c->cp.p + qb, sizeof(float4) * nq, hipMemcpyDeviceToHost));
        int ring = 0;
        for (int i = 0; i < nq; i++) {                              // rows in the reference's query order; the table may be in patch order
            const int qi = qb + i;
            while (ring + 1 < c->n_src_rings && c->h_q_off[ring + 1] <= qi) ring++;
            const int t = c->q_patch ? patch_position(c->h_q_off.data(), c->n_src_rings, ring, qi - c->h_q_off[ring], c->patch_rings, c->patch_len) : qi;
            int valid; std::memcpy(&valid, &p[t - qb].w, 4);
            if (valid) h_icp[t] = rows++;
        }
    }
    if (n_rows) *n_rows = rows;
    if (!residuals || !jacobian) return VELO_OK;
    if (capacity_rows < rows) return fail(VELO_ERR_INVALID, "row capacity %d < %d", capacity_rows, rows);
    if (rows == 0) return VELO_OK;
    VELO_TRY(c->row_off_vis.reserve(std::max(h_vis.size(), (size_t)1)));
    VELO_TRY(c->row_off_icp.reserve(h_icp.size()));
    VELO_TRY(c->rows_r.reserve((size_t)rows));
    VELO_TRY(c->rows_J.reserve((size_t)rows * 6));
    if (!h_vis.empty()) HIP_TRY(hipMemcpy(c->row_off_vis.p, h_vis.data(), sizeof(int) * h_vis.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->row_off_icp.p, h_icp.data(), sizeof(int) * h_icp.size(), hipMemcpyHostToDevice));
    std::memcpy(c->h_x, x, sizeof(double) * 6);
    HIP_TRY(hipMemcpyAsync(c->xdev.p, c->h_x, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
    EvalArgs A = eval_args(c, c->xdev.p);
    A.rows_r = c->rows_r.p; A.rows_J = c->rows_J.p; A.row_offset_vis = c->row_off_vis.p; A.row_offset_icp = c->row_off_icp.p;
    launch_eval(c, A, eval_plan(A));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(residuals, c->rows_r.p, sizeof(double) * rows, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(jacobian, c->rows_J.p, sizeof(double) * (size_t)rows * 6, hipMemcpyDeviceToHost));
    return VELO_OK;
}

int velo_evaluate_functors(velo_ctx* c, const velo_functor* f, int32_t n, const double x[6], double* residuals, double* jacobians) {
    if (!c || !x || n < 0 || (n > 0 && (!f || !residuals))) return fail(VELO_ERR_INVALID, "bad functor batch arguments");
    static_assert(sizeof(FunctorRec) == sizeof(velo_functor), "device/host functor layout");
    for (int i = 0; i < n; i++) if (f[i].kind < 0 || f[i].kind > VELO_FUNCTOR_3DPD) return fail(VELO_ERR_INVALID, "functor %d: kind %d", i, f[i].kind);
    if (n == 0) return VELO_OK;
    HIP_TRY(hipSetDevice(c->device));
    VELO_TRY(c->fn_in.reserve((size_t)n));
    VELO_TRY(c->rows_r.reserve((size_t)n * 3));
    if (jacobians) VELO_TRY(c->rows_J.reserve((size_t)n * 18));
    std::memcpy(c->h_x, x, sizeof(double) * 6);
    HIP_TRY(hipMemcpyAsync(c->xdev.p, c->h_x, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->fn_in.p, f, sizeof(velo_functor) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(functor_batch_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const FunctorRec*)c->fn_in.p, n, (const double*)c->xdev.p,
                       c->rows_r.p, jacobians ? c->rows_J.p : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));              // also: the pageable functor array has been read
    HIP_TRY(hipMemcpy(residuals, c->rows_r.p, sizeof(double) * (size_t)n * 3, hipMemcpyDeviceToHost));
    if (jacobians) HIP_TRY(hipMemcpy(jacobians, c->rows_J.p, sizeof(double) * (size_t)n * 18, hipMemcpyDeviceToHost));
    return VELO_OK;
}

int velo_residual_stats_at(velo_ctx* c, const double x[6], velo_residual_stats* out) {
    if (!c || !x || !out) return fail(VELO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    std::memset(out, 0, sizeof(*out));
    if (!c->vflags_valid) VELO_TRY(do_build_visual(c, x, false, 1, nullptr));
    std::memcpy(c->h_x, x, sizeof(double) * 6);
    HIP_TRY(hipMemcpyAsync(c->xdev.p, c->h_x, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
    EvalArgs A = eval_args(c, c->xdev.p);
    const int n = 3 * A.n_matches + (A.q_end - A.q_begin);
    int blocks = 0, residuals = 0;
    visual_counts(c, &blocks, &residuals);
    if (c->shard_rank != 0) { blocks = 0; residuals = 0; }
    if (n > 0) {
        const int nb = cdiv(n, 256);
        VELO_TRY(c->stat_vals.reserve((size_t)n)); VELO_TRY(c->stat_types.reserve((size_t)n)); VELO_TRY(c->stat_part.reserve((size_t)nb * (kStatTypes + 1)));
        const bool fresh = c->stat_hist.cap == 0;
        VELO_TRY(c->stat_hist.reserve((size_t)kStatTypes * kStatBins)); VELO_TRY(c->stat_work.reserve(1)); VELO_TRY(c->stat_out.reserve(1));
        if (fresh) HIP_TRY(hipMemsetAsync(c->stat_hist.p, 0, sizeof(int) * (size_t)kStatTypes * kStatBins, c->stream));   // the pick kernel leaves it cleared
        hipLaunchKernelGGL(residual_norms_kernel, dim3(nb), dim3(256), 0, c->stream, (const double*)c->xdev.p, A, c->stat_vals.p, c->stat_types.p, c->stat_part.p);
        for (int pass = 0; pass < 4; pass++) {
            hipLaunchKernelGGL(stats_hist_kernel, dim3(nb), dim3(256), 0, c->stream, (const double*)c->stat_vals.p, (const signed char*)c->stat_types.p, n, pass, (const StatWork*)c->stat_work.p, c->stat_hist.p);
            hipLaunchKernelGGL(stats_pick_kernel, dim3(kStatTypes), dim3(256), 0, c->stream, pass, c->stat_work.p, c->stat_hist.p);
        }
        hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(64), 0, c->stream, (const double*)c->stat_part.p, nb, c->stat_work.p, c->stat_out.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, c->stat_out.p, sizeof(*out), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    out->n_blocks = blocks + (int)out->type[VELO_FUNCTOR_3DPD].count;
    out->n_residuals = residuals + (int)out->type[VELO_FUNCTOR_3DPD].count;
    return VELO_OK;
}

int velo_solve(velo_ctx* c, double x[6], velo_solve_summary* summary) {
    if (!c || !x) return fail(VELO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    double xo[6];
    VELO_TRY(do_solve(c, x, xo, summary, nullptr));
    for (int k = 0; k < 6; k++) x[k] = xo[k];
    return VELO_OK;
}

}  // extern "C"

// ---- chain mode ---------------------------------------------------------------------------------------------------------------------
// The whole call as ONE chain of launches with ONE host synchronisation at its end.  What the host needed between rounds -- the pose
// scalars of the next association, the solve summary -- stays on the device: the LM launch that finishes a solve writes a
// PoseRecord (pose_scalars_compute: the pinned sin/cos, bit-identical to the host's) and a SolveLog; the next round's tube kernel
// reads the record.  The host cannot see when a solve ends, so it enqueues as many LM launches per solve as the same solve of the
// previous call needed plus a margin (launches behind the end of a solve copy the state through, ~3 us each); a solve that needs
// more raises the chain's failure flag in the next association (its record is not ready), everything behind it drains, and the
// call is repeated by the host-driven path below -- same kernels, same arithmetic, so the result does not depend on which path ran.
// Launch-count prediction of solve k = what the same solve of the previous call needed + a margin.  The margin follows how far that
// count has moved over the last four calls (1 + spread, between 1 and 3); until four calls have been seen, and after a miss, it is
// the default 2.  A launch behind the end of a solve costs ~3.5 us, a miss a whole repeated call.
static int preload_group(velo_ctx** ctxs, int n, hipStream_t bs, bool* any_loaded);      // velo_hint_next_frame: defined with the batch driver below
static int undo_preload(velo_ctx* c);
static void note_evals(velo_ctx* c, int k, int evals) {
    if (k < 0 || k >= VELO_MAX_SOLVES) return;
    c->pred_evals[k] = evals;
    c->eval_hist[k][c->eval_hist_n[k] & 3] = evals;
    c->eval_hist_n[k]++;
}
static int margin_for(const velo_ctx* c, int k) {
    k = std::min(std::max(k, 0), VELO_MAX_SOLVES - 1);
    if (c->chain_margin_fixed || c->eval_hist_n[k] < 4) return c->chain_margin;
    int mn = c->eval_hist[k][0], mx = mn;
    for (int i = 1; i < 4; i++) { mn = std::min(mn, c->eval_hist[k][i]); mx = std::max(mx, c->eval_hist[k][i]); }
    static const int base = dev_env("VELO_MARGIN_BASE") ? atoi(dev_env("VELO_MARGIN_BASE")) : 1;      // A/B (diagnostics build)
    return std::min(std::max(base + (mx - mn), 1), 3);
}
static void note_miss(velo_ctx* c) {
    for (int k = 0; k < VELO_MAX_SOLVES; k++) { c->pred_evals[k] += 2; c->eval_hist_n[k] = 0; }    // the host-driven repeat records the real counts
}

static bool chain_eligible(velo_ctx* c) {
    if (!c->chain || c->want_stats || c->comm || c->use_graphs || !c->lm_merged || !c->P.enable_icp) return false;
    if (c->assoc_variant >= 0 && c->assoc_variant != 5) return false;
    if (c->debug_skip || c->lm_trace_on || c->tube_map >= 0) return false;
    // several ranks: only the query-sharded mode over peer slabs (every rank holds the same state, so every rank computes the same
    // record and the same launch counts; the all-reduce lives inside the step kernel, no host in between)
    if (c->shard_world != 1 && !(c->peer_on && !c->target_sharded)) return false;
    if (c->peer_on && c->target_sharded) return false;
    if (c->peer_on && c->n_q < 64 * c->shard_world) return false;    // (a rank-uniform test: every rank must take the same path, and n_q is the global count)
    if (c->P.f2f_iterations * c->P.icp_iterations < 1) return false;
    return true;
}

static int frame_to_frame_chain(velo_ctx* c, double xc[6], velo_summary* S, bool* completed) {
    *completed = false;
    if (query_list_stale(c)) VELO_TRY(build_query_list(c));
    int qb, qe;
    q_range(c, &qb, &qe);
    if (qe <= qb) return VELO_OK;
    bool small = false;                                              // problems of a few workgroups: a whole solve is ONE launch (lm_solve_small_kernel)
    {
        EvalArgs A0;
        std::memset(&A0, 0, sizeof(A0));
        A0.q_begin = qb; A0.q_end = qe;
        const EvalPlan E0 = eval_plan(A0);
        if (E0.nb_icp <= 0) return VELO_OK;
        small = c->small_solve && E0.total() <= kSmallRows;
    }
    const LMParams Q = lm_params(c->P);
    const size_t half = (size_t)(kMaxEvalBlocks + kMaxVisBlocks) * kNumAcc;
    const int max_launches = c->P.max_num_iterations + 2;
    std::memcpy(c->h_x, xc, sizeof(double) * 6);
    HIP_TRY(hipMemcpyAsync(c->xdev.p, c->h_x, sizeof(double) * 6, hipMemcpyHostToDevice, c->stream));
    // Visual blocks: the residual-type choice + outlier gate of every f2f iteration (velo.h:622-792) runs on the device at the pose
    // the device holds (iteration 1: the initial guess, later: the state's x); block / residual counts per iteration and the
    // last iteration's flags come back with everything else at the end.  The solves then take sweep + visual sweep + step launches.
    const bool visual = c->n_matches > 0;
    if (visual) {
        VELO_TRY(c->vis_counts.reserve(2 * VELO_MAX_STATS));
        HIP_TRY(hipMemsetAsync(c->vis_counts.p, 0, sizeof(int) * 2 * VELO_MAX_STATS, c->stream));
        if (c->P.f2f_iterations > VELO_MAX_STATS) return VELO_OK;
        c->vflags_valid = true;
    } else {
        VELO_TRY(do_build_visual(c, xc, false, 1, nullptr));         // no measurements: only clears the host flags
    }
    c->have_corr = false;
    c->chain_calls++;
    velo_ctx::TimingMark tmark;
    c->timing_mark(&tmark);
    int j = 0, r = 0;                                                // launch counter (its parity selects the double-buffer halves), round
    const int rounds = c->P.f2f_iterations * c->P.icp_iterations;
    // Over peers every rank must enqueue the same number of LM launches per solve (the all-reduce sits inside the step kernel): the
    // ranks agree on the maximum of their predictions before anything else is enqueued -- one tiny launch and one synchronisation.
    int k_agreed[VELO_MAX_SOLVES];
    if (c->peer_on) {
        static_assert(VELO_MAX_SOLVES <= 64, "AgreeCounts holds 64 counts (one lane each)");
        AgreeCounts mine;
        std::memset(&mine, 0, sizeof(mine));
        for (int k = 0; k < VELO_MAX_SOLVES; k++) mine.v[k] = std::min(std::max(c->pred_evals[k], 1) + margin_for(c, k), max_launches);
        hipLaunchKernelGGL(peer_agree_kernel, dim3(1), dim3(64), 0, c->stream, c->peer, mine, (int)VELO_MAX_SOLVES, c->h_agree);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        VELO_TRY(peer_check(c));
        for (int k = 0; k < VELO_MAX_SOLVES; k++) k_agreed[k] = std::min(std::max(c->h_agree[k], 1), max_launches);
    }
    for (int iter = 1; iter <= c->P.f2f_iterations; iter++) {
        if (visual) {
            // (x of a later iteration: the state buffer the last launch wrote -- index j & 1 with one-launch iterations, else buffer 0)
            hipLaunchKernelGGL(visual_gate_kernel, dim3(cdiv(c->n_matches, 128)), dim3(128), 0, c->stream, (const double*)(iter == 1 ? c->xdev.p : (c->state.p + (j & 1))->x),
                               visual_params(c->P), c->vm.p, c->n_matches, iter, c->vflags.p, c->vis_counts.p + 2 * (iter - 1));
            HIP_TRY(hipGetLastError());
        }
        for (int icp_iter = 0; icp_iter < c->P.icp_iterations; icp_iter++, r++) {
            int nv = 0;
            VELO_TRY(do_associate(c, xc, iter, false, false, &nv, false, r == 0 ? nullptr : c->pose_rec.p));
            const EvalArgs A = eval_args(c, nullptr);
            const EvalPlan E = eval_plan(A);
            if (E.total() <= 0) return fail(VELO_ERR_STATE, "chain mode: unexpected evaluation plan");
            const int* nvp = c->n_valid.p + c->nv_idx;
            SolveLog* logp = c->solve_log.p + std::min(r, VELO_MAX_SOLVES - 1);
            const bool peer = c->peer_on;
            if (small && !peer && (!visual || E.total() <= kSmallRows)) {     // no prediction needed: the launch runs the solve to its end
                c->lm_kernel_name = E.nb_vis > 0 ? "lm_solve_small_kernel" : "lm_solve_small_icp_kernel";
                c->lm_round_name[std::min(r, VELO_MAX_SOLVES - 1)] = c->lm_kernel_name;
                if (E.nb_vis > 0)
                    VELO_LAUNCH_T(c, c->lm_kernel_name, 0, lm_solve_small_kernel, dim3(1), dim3(kEvalThreads), 0, c->stream, A, Q, c->state.p, (const double*)(r == 0 ? c->xdev.p : nullptr),
                                  nvp, E.nb_icp, E.nb_vis, c->P.max_num_iterations + 3, c->pose_rec.p, logp);
                else                                                  // no visual blocks: the instantiation without their code
                    VELO_LAUNCH_T(c, c->lm_kernel_name, 0, lm_solve_small_icp_kernel, dim3(1), dim3(kEvalThreads), 0, c->stream, A, Q, c->state.p, (const double*)(r == 0 ? c->xdev.p : nullptr),
                                  nvp, E.nb_icp, c->P.max_num_iterations + 3, c->pose_rec.p, logp);
                HIP_TRY(hipGetLastError());
                continue;
            }
            if (peer || (visual && c->lm_trace_vis_off)) {           // sweep (+ visual sweep) + step per LM iteration, state single-buffered
                c->lm_round_name[std::min(r, VELO_MAX_SOLVES - 1)] = nullptr;      // (separate sweep / step launches: no bytes booked on a name)
                const int Kv = peer ? k_agreed[std::min(r, VELO_MAX_SOLVES - 1)]
                                    : std::min(std::max(c->pred_evals[std::min(r, VELO_MAX_SOLVES - 1)], 1) + margin_for(c, r), max_launches);
                hipLaunchKernelGGL(lm_begin_kernel, dim3(1), dim3(64), 0, c->stream, c->state.p, c->eval_pt.p, (const double*)(r == 0 ? c->xdev.p : nullptr), nvp, c->pose_rec.p);
                for (int k = 0; k < Kv; k++) {
                    launch_eval(c, A, E);
                    if (peer) hipLaunchKernelGGL(lm_step_peer_kernel, dim3(1), dim3(256), 0, c->stream, Q, c->state.p, c->eval_pt.p, (const double*)c->partials.p, E.total(),
                                                 c->peer, c->pose_rec.p, logp);
                    else hipLaunchKernelGGL(lm_step_kernel, dim3(1), dim3(256), 0, c->stream, Q, c->state.p, c->eval_pt.p, (const double*)c->partials.p, E.total(),
                                            (unsigned long long*)nullptr, 0, c->pose_rec.p, logp);
                }
                HIP_TRY(hipGetLastError());
                continue;
            }
            const int K = std::min(std::max(c->pred_evals[std::min(r, VELO_MAX_SOLVES - 1)], 1) + 1 + margin_for(c, r), max_launches);
            c->lm_kernel_name = visual ? "lm_iter_vis_kernel" : "lm_iter_kernel";
            c->lm_round_name[std::min(r, VELO_MAX_SOLVES - 1)] = c->lm_kernel_name;
            for (int k = 0; k < K; k++, j++) {
                if (visual)
                    VELO_LAUNCH_T(c, c->lm_kernel_name, 0, lm_iter_vis_kernel, dim3(E.total()), dim3(kEvalThreads), 0, c->stream, A, Q, (const LMState*)(c->state.p + (j & 1)), c->state.p + ((j + 1) & 1),
                                  (const double*)(c->partials.p + (size_t)(j & 1) * half), E.total(), c->partials.p + (size_t)((j + 1) & 1) * half, k == 0 ? 1 : 0,
                                  (const double*)((r == 0 && k == 0) ? c->xdev.p : nullptr), nvp, c->pose_rec.p, logp, E.nb_icp, E.nb_vis);
                else
                    VELO_LAUNCH_T(c, c->lm_kernel_name, 0, lm_iter_kernel, dim3(E.nb_icp), dim3(kEvalThreads), 0, c->stream, A, Q, (const LMState*)(c->state.p + (j & 1)), c->state.p + ((j + 1) & 1),
                                  (const double*)(c->partials.p + (size_t)(j & 1) * half), E.nb_icp, c->partials.p + (size_t)((j + 1) & 1) * half, k == 0 ? 1 : 0,
                                  (const double*)((r == 0 && k == 0) ? c->xdev.p : nullptr), nvp, c->pose_rec.p, c->solve_log.p + std::min(r, VELO_MAX_SOLVES - 1));
            }
            HIP_TRY(hipGetLastError());
        }
    }
    // the last solve has no association behind it that would notice an unfinished solve: the final state says so itself
    int* h_fail = reinterpret_cast<int*>(c->h_log + VELO_MAX_SOLVES);
    HIP_TRY(hipMemcpyAsync(c->h_log, c->solve_log.p, sizeof(SolveLog) * (size_t)std::min(rounds, VELO_MAX_SOLVES), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_fail, c->chain_fail.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&c->h_status->s, c->state.p + (j & 1), sizeof(LMState), hipMemcpyDeviceToHost, c->stream));
    int* h_vis_counts = h_fail + 1;                                  // 2 x VELO_MAX_STATS ints behind the failure flag (the pinned block has 64 spare bytes)
    int* pin_flags = nullptr;
    if (visual) {                                                    // (through page-locked memory: a pageable destination makes the copy a staged, host-blocking one)
        VELO_TRY(pin_acquire(c, 3, ((size_t)3 * c->n_matches + sizeof(int) - 1) / sizeof(int), &pin_flags));
        HIP_TRY(hipMemcpyAsync(pin_flags, c->vflags.p, (size_t)3 * c->n_matches, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h_vis_counts, c->vis_counts.p, sizeof(int) * 2 * VELO_MAX_STATS, hipMemcpyDeviceToHost, c->stream));
    }
    // what the summary says about THIS call's scans, before a frame loaded ahead replaces them
    const int nq_call = c->n_q, nt_call = c->T->n_tgt;
    const bool ahead = c->nf.hint_valid && !c->peer_on && !c->comm;   // (sharded registrations load their slices together: nothing ahead)
    if (ahead) {
        if (!c->nf.call_done) HIP_TRY(hipEventCreateWithFlags(&c->nf.call_done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->nf.call_done, c->stream));
    }
    VELO_TRY(prefetch_issue(c));                                     // the next frame's upload runs under this chain (velo_hint_next_source)
    bool preloaded = false;
    if (ahead) VELO_TRY(preload_group(&c, 1, c->stream, &preloaded));  // ... and its promotion, ingest and index build behind it (velo_hint_next_frame)
    if (preloaded) HIP_TRY(hipEventSynchronize(c->nf.call_done));    // the results are in; the next frame's loads are still running
    else HIP_TRY(hipStreamSynchronize(c->stream));
    if (visual) { const unsigned char* pf = reinterpret_cast<const unsigned char*>(pin_flags); c->h_vflags.assign(pf, pf + (size_t)3 * c->n_matches); }
    VELO_TRY(peer_check(c));
    if (*h_fail || !c->h_status->s.done) {
        if (preloaded) { HIP_TRY(hipStreamSynchronize(c->stream)); VELO_TRY(undo_preload(c)); }     // the repeat runs on the pair this call registered
        HIP_TRY(hipMemsetAsync(c->chain_fail.p, 0, sizeof(int), c->stream));
        note_miss(c);
        c->nv_clean[0] = c->nv_clean[1] = false;                            // drained association launches did not clear the next round's counter
        c->ask_clean[0] = c->ask_clean[1] = false;
        c->chain_misses++;
        c->timing_rewind(tmark);
        return VELO_OK;
    }
    const uint64_t nq = (uint64_t)nq_call;
    for (int k = 0; k < rounds; k++) {
        const SolveLog& L = c->h_log[std::min(k, VELO_MAX_SOLVES - 1)];
        S->n_assoc_rounds++;
        S->n_queries = nq_call;
        const uint64_t b_assoc = 12ull * nq + 12ull * (uint64_t)nt_call + 28ull * nq;
        S->assoc_bytes += b_assoc; S->algorithmic_bytes += b_assoc;
        S->assoc_kernel_launches++;
        velo_solve_summary ss;
        std::memset(&ss, 0, sizeof(ss));
        ss.termination = L.termination; ss.lm_iterations = L.iter; ss.evaluations = L.evals; ss.n_icp_valid = L.n_valid;
        ss.initial_cost = L.initial_cost; ss.final_cost = L.final_cost;
        if (visual && c->shard_rank == 0) {                          // the blocks of the f2f iteration this solve belongs to (rank 0 reports them)
            const int it0 = std::min(k / std::max(c->P.icp_iterations, 1), VELO_MAX_STATS - 1);
            ss.n_visual_blocks = h_vis_counts[2 * it0]; ss.n_visual_residuals = h_vis_counts[2 * it0 + 1];
        }
        note_evals(c, k, L.evals);
        S->eval_kernel_launches += L.evals;
        S->algorithmic_bytes += (uint64_t)L.evals * (36ull * (uint64_t)L.n_valid + 32ull * (uint64_t)ss.n_visual_blocks + 224ull);
        if (c->timing >= 2) kacc_add(c, c->lm_round_name[std::min(k, VELO_MAX_SOLVES - 1)], 0.0, 0, 0, (uint64_t)L.evals * (36ull * (uint64_t)L.n_valid + 32ull * (uint64_t)ss.n_visual_blocks + 224ull));
        if (S->n_solves < VELO_MAX_SOLVES) S->solves[S->n_solves] = ss;
        S->n_solves++;
    }
    c->last_n_valid = c->h_status->s.n_valid;
    for (int k = 0; k < 6; k++) xc[k] = c->h_status->s.x[k];
    *completed = true;
    return VELO_OK;
}

extern "C" {

int velo_get_kernel_times(velo_ctx* c, velo_kernel_time* out, int32_t capacity, int32_t* n, int32_t reset) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (n) *n = (int32_t)c->kacc.size();
    for (int i = 0; out && i < capacity && i < (int)c->kacc.size(); i++) {
        std::memset(&out[i], 0, sizeof(out[i]));
        std::snprintf(out[i].name, sizeof(out[i].name), "%s", c->kacc[(size_t)i].name);
        const velo_ctx::KernelAcc& a = c->kacc[(size_t)i];
        out[i].sampled = a.sampled; out[i].launches = a.launches; out[i].algorithmic_bytes = a.bytes;
        out[i].ms = a.sampled > 0 ? a.ms * (double)a.launches / (double)a.sampled : 0.0;
    }
    if (reset) c->kacc.clear();
    return VELO_OK;
}

int velo_chain_stats(const velo_ctx* c, int32_t* calls, int32_t* misses) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (calls) *calls = c->chain_calls;
    if (misses) *misses = c->chain_misses;
    return VELO_OK;
}

int velo_frame_to_frame(velo_ctx* c, double x[6], double T[16], velo_summary* summary) {
    if (!c || !x) return fail(VELO_ERR_INVALID, "null argument");
    if (!c->have_target || !c->have_source) return fail(VELO_ERR_STATE, "frame_to_frame needs set_target and set_source first");
    if (c->nf.state == velo_ctx::NextFrame::LOADED) return fail(VELO_ERR_STATE, "the context holds a frame loaded ahead (velo_hint_next_frame): the job that brings it must come first");
    if (c->nf.state == velo_ctx::NextFrame::CONSUMED) c->nf.state = velo_ctx::NextFrame::NONE;
    struct HintEnd { velo_ctx* c; ~HintEnd() { c->nf.hint_valid = false; } } hint_end{c};
    HIP_TRY(hipSetDevice(c->device));
    velo_summary local;
    velo_summary* S = summary ? summary : &local;
    std::memset(S, 0, sizeof(*S));
    S->n_target = c->T->n_tgt;
    c->assoc_events_used = 0;
    double xc[6];
    for (int k = 0; k < 6; k++) xc[k] = x[k];
    if (chain_eligible(c)) {
        bool completed = false;
        VELO_TRY(frame_to_frame_chain(c, xc, S, &completed));
        if (completed) {
            if (c->timing) VELO_TRY(read_assoc_timing(c, S));
            for (int k = 0; k < 6; k++) x[k] = xc[k];
            if (T) velo_pose_vec_to_mat(x, T);
            return VELO_OK;
        }
        std::memset(S, 0, sizeof(*S));
        S->n_target = c->T->n_tgt;
        c->assoc_events_used = 0;
        for (int k = 0; k < 6; k++) xc[k] = x[k];
    }
    for (int iter = 1; iter <= c->P.f2f_iterations; iter++) {                       // velo.h:616
        VELO_TRY(do_build_visual(c, xc, false, iter, nullptr));                      // velo.h:622-792
        c->have_corr = false;
        c->last_n_valid = 0;
        for (int icp_iter = 0; icp_iter < c->P.icp_iterations; icp_iter++) {        // velo.h:800
            int nv = 0;
            if ((c->comm || c->peer_on) && c->target_sharded) VELO_TRY(associate_target_sharded(c, xc, iter, false));
            else VELO_TRY(do_associate(c, xc, iter, false, false, &nv));             // velo.h:806-894 (no host sync: the count rides on the LM status)
            int qb, qe;
            q_range(c, &qb, &qe);
            S->n_assoc_rounds++;
            S->n_queries = c->n_q;
            const uint64_t nq = (uint64_t)c->n_q;
            const uint64_t b_assoc = 12ull * nq + 12ull * (uint64_t)c->T->n_tgt + 28ull * nq;
            S->assoc_bytes += b_assoc; S->algorithmic_bytes += b_assoc;
            if (qe > qb) S->assoc_kernel_launches++;
            velo_solve_summary ss;
            int evals = 0;
            const int solve_idx = std::min(S->n_solves, VELO_MAX_SOLVES - 1);
            // consecutive frames behave alike: size the first chunk to the evaluations this solve needed last time (+1)
            VELO_TRY(do_solve(c, xc, xc, &ss, &evals, std::min(std::max(c->pred_evals[solve_idx] + 1, 2), c->P.max_num_iterations + 1)));   // velo.h:897-902
            note_evals(c, solve_idx, ss.evaluations);
            S->eval_kernel_launches += evals;
            S->algorithmic_bytes += (uint64_t)ss.evaluations * (36ull * (uint64_t)ss.n_icp_valid + 32ull * (uint64_t)ss.n_visual_blocks + 224ull);
            if (S->n_solves < VELO_MAX_SOLVES) S->solves[S->n_solves] = ss;
            S->n_solves++;
        }
        if (c->want_stats && iter <= VELO_MAX_STATS) {                              // velo.h:909
            VELO_TRY(velo_residual_stats_at(c, xc, &S->residual_stats[iter - 1]));
            S->n_residual_stats = iter;
        }
    }
    if (c->timing) VELO_TRY(read_assoc_timing(c, S));
    for (int k = 0; k < 6; k++) x[k] = xc[k];
    if (T) velo_pose_vec_to_mat(x, T);
    return VELO_OK;
}

// Lock-step batch: every context does what velo_frame_to_frame does, in the same order and with the same kernels' arithmetic,
// but the n contexts advance together on ONE stream and share launches in the LM phase: one sweep launch covers the
// point-to-plane residuals of all contexts (blockIdx.y = context), one launch steps all n LM states, one copy brings all n
// states back per chunk.  With a host thread per context (the fallback) a step of 8 pairs is 8 x 84 small LM launches that each
// fill half the chip and stall behind the other contexts' association kernels; here it is 84 launches that fill it.
// Conditions: same device, same parameters, no communicator, default kernels -- anything else falls back.
static bool batch_can_lockstep(velo_ctx** ctxs, int n, bool targets_follow = false, bool sources_follow = false) {
    if (n < 2 || !ctxs[0] || !ctxs[0]->batch_lockstep) return false;
    for (int i = 0; i < n; i++) {
        const velo_ctx* c = ctxs[i];
        if (!c || c->device != ctxs[0]->device || c->comm || c->peer_on || c->use_graphs || c->want_stats) return false;
        if ((!c->have_target && !targets_follow) || (!c->have_source && !sources_follow) || c->shard_world != 1) return false;
        if (std::memcmp(&c->P, &ctxs[0]->P, sizeof(velo_params)) != 0) return false;
        for (int j = 0; j < i; j++) if (ctxs[j] == c) return false;
    }
    return true;
}

// ---- the same association round of several contexts in ONE launch (lock-step batch driver) ---------------------------------------
// Host-side preparation of one context's round for the tube kernel, exactly what do_associate does before its launch.
// *groups = 0 when the context has no queries.
static int prepare_assoc_v5(velo_ctx* c, const double x[6], int iter, AssocArgs* A, int* groups, bool* asker, bool* lane, bool* direct, SeedArgs* SA = nullptr, bool* seeded = nullptr) {
    if (!c->have_target || !c->have_source) return fail(VELO_ERR_STATE, "associate needs set_target and set_source first");
    if (query_list_stale(c)) VELO_TRY(build_query_list(c));
    Grid* G = grid_for_iter(c, iter);
    if (!G) return fail(VELO_ERR_STATE, "the target's search index has not been built");
    int qb, qe;
    q_range(c, &qb, &qe);
    VELO_TRY(next_valid_counter(c));
    *groups = 0;
    if (qe <= qb) return VELO_OK;
    pose_scalars(x, &A->P);
    A->P_dev = nullptr; A->chain_fail = nullptr;
    G->view(&A->G);
    A->qpts = c->qpts; A->q_begin = qb; A->q_end = qe;
    A->tgt_pad = c->T->tgt_pad.p; A->tgt_off = c->T->tgt_off.p;
    const double gate = gate_of_iter(c->P, iter);
    A->gate_bits = gate_bits_of(gate);
    A->norm_cond = c->P.icp_norm_condition;
    const int cluster_cells = std::max(1, (int)std::lround((double)c->cluster_w * 0.1785 / G->h));
    A->cluster_w = c->cluster_w_set ? (c->cluster_w > 0 ? cluster_cells : 2000) : std::max(1, (int)std::lround(96.0 * 0.1785 / G->h));
    A->h_safe = (float)(G->h * 0.999);
    AssocOut& out = A->out;
    out.p = c->cp.p; out.n = c->cn.p; out.v0 = c->cv0.p; out.aux0 = c->aux0.p; out.aux1 = c->aux1.p; out.n_valid = c->n_valid.p + c->nv_idx; out.dbg = c->dbg.p; out.wg_times = nullptr;
    out.first_ring = c->T->tgt_first_ring; out.first_point = c->T->tgt_first_point; out.partial = nullptr;
    const bool image_seeds = SA != nullptr && seeds_from_image(c, iter, false);
    int had_prev = 0;
    VELO_TRY(attach_seeds(c, &out, image_seeds, &had_prev));
    if (seeded) *seeded = image_seeds && out.prev_a != nullptr;
    if (image_seeds && out.prev_a) fill_seed_args(c, SA, A->P, nullptr, nullptr, qb, qe, out, had_prev);
    *direct = direct_round(c, qe - qb, false);
    *lane = !*direct && lane_round(c, G, false);
    const bool cold = c->seed_rounds == 0;
    if (out.prev_a) c->seed_rounds++;
    {
        const int reach0 = (int)std::ceil(std::sqrt(std::max(gate_of_iter(c->P, 1), 0.0)) / (G->h * 0.999));
        const int ar = c->asker_rows >= 0 ? c->asker_rows : (reach0 > 5 ? 0 : (1 << 30));
        // The list pays off for ONE pair in flight (a launch's tail is idle chip); with several groups in flight other streams' kernels fill the
        // tail anyway and the second launch only costs (8 pairs on the 2M-point map: 938 vs 957 pairs/s).  VELO_ASKER_QUEUE=2 forces it here too.
        VELO_TRY(attach_askers(c, &out, ar < (1 << 30) && !*lane && !*direct && cold && c->asker_queue >= 2));
    }
    out.n_valid_next = c->n_valid.p + (c->nv_idx ^ 1);
    c->nv_clean[c->nv_idx ^ 1] = true;
    A->want_aux = 0; A->group_perm = c->xcd_chunks ? (c->xcd_chunks == 2 ? kXcdTiles : kXcdChunks) : nullptr; A->dbg = 0;
    const int reach_cells = (int)std::ceil(std::sqrt(std::max(gate_of_iter(c->P, 1), 0.0)) / (G->h * 0.999));
    A->asker_rows = c->asker_rows >= 0 ? c->asker_rows : (reach_cells > 5 ? 0 : (1 << 30));
    *asker = A->asker_rows < (1 << 30);
    if (*asker && c->dense_batch) A->dbg = c->dense_rows | (c->dense_far << 20);                       // (see assoc_search_v5_body: groups that go query by query as a whole)
    *groups = cdiv(qe - qb, 64);
    return VELO_OK;
}

// contexts whose round may share a launch: default tube kernel, no diagnostics, no placement table, whole (unsharded) query list
static bool assoc_batchable(const velo_ctx* c) {
    static const bool on = dev_env("VELO_ASSOC_BATCH") ? atoi(dev_env("VELO_ASSOC_BATCH")) != 0 : true;
    return on && (c->assoc_variant < 0 || c->assoc_variant == 5) && !c->debug_skip && c->tube_map < 0 && !c->comm && !c->peer_on;
}

// All contexts share one stream here (the lock-step driver swapped it in).  launched[i] = 1 for the context that carries the timing
// events of its launch, 0 for the others of the same launch.
static int do_associate_group(velo_ctx** ctxs, int n, const std::vector<std::array<double, 6>>& xs, int iter, std::vector<int>& launched,
                              const PoseRecord* pose_dev = nullptr, int* fail_dev = nullptr) {
    // pose_dev / fail_dev (chain mode): per-context device pose records [n] the kernels read instead of xs, and failure flags [n]
    launched.assign((size_t)n, 0);
    bool all = true;
    for (int i = 0; i < n; i++) all = all && assoc_batchable(ctxs[i]);
    if (!all || n < 2) {
        for (int i = 0; i < n; i++) {
            int nv = 0;
            if (pose_dev) return fail(VELO_ERR_STATE, "chain mode needs contexts whose rounds share a launch");
            VELO_TRY(do_associate(ctxs[i], xs[(size_t)i].data(), iter, false, false, &nv));
            int qb, qe; q_range(ctxs[i], &qb, &qe);
            launched[(size_t)i] = qe > qb ? 1 : 0;
        }
        return VELO_OK;
    }
    for (int b = 0; b < n; b += kAssocBatchMax) {
        const int m = std::min(kAssocBatchMax, n - b);
        AssocBatch B;
        std::memset(&B, 0, sizeof(B));
        SeedBatch SB;
        std::memset(&SB, 0, sizeof(SB));
        int gmax = 0, k = 0, first = -1, n_seeded = 0;
        bool any_asker = false, all_lane = true, all_direct = true;
        int nq_max = 0;
        for (int i = b; i < b + m; i++) {
            int groups = 0; bool asker = false, lane = false, direct = false, seeded = false;
            VELO_TRY(prepare_assoc_v5(ctxs[i], xs[(size_t)i].data(), iter, &B.item[k], &groups, &asker, &lane, &direct, &SB.item[n_seeded], &seeded));
            if (groups > 0) { all_lane = all_lane && lane; all_direct = all_direct && direct; nq_max = std::max(nq_max, B.item[k].q_end - B.item[k].q_begin); }
            ctxs[i]->have_corr = true;
            if (pose_dev) { B.item[k].P_dev = pose_dev + i; B.item[k].chain_fail = fail_dev + i; }
            if (seeded && groups > 0) {
                if (pose_dev) { SB.item[n_seeded].P_dev = pose_dev + i; SB.item[n_seeded].chain_fail = fail_dev + i; }
                n_seeded++;
            }
            if (groups == 0) continue;                                  // no queries: nothing to launch for it
            if (first < 0) first = i;
            gmax = std::max(gmax, groups); any_asker = any_asker || asker; k++;
        }
        if (k == 0) continue;
        if (ctxs[first]->xcd_chunks) gmax = ctxs[first]->xcd_chunks == 2 ? 64 * cdiv(gmax, 64) : 8 * cdiv(gmax, 8);
        velo_ctx* c = ctxs[first];
        std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
        uint64_t bytes = 0;                                           // B_assoc of every context this launch serves
        for (int j = 0; j < k; j++) bytes += 40ull * (uint64_t)(B.item[j].q_end - B.item[j].q_begin);
        for (int i = b; i < b + m; i++) { int q0, q1; q_range(ctxs[i], &q0, &q1); if (q1 > q0) bytes += 12ull * (uint64_t)ctxs[i]->T->n_tgt; }
        const char* assoc_name = all_direct ? "assoc_direct_batch_kernel" : "assoc_search_v5_batch_kernel";
        if (assoc_bracket(c, assoc_name, bytes)) {
            if (c->assoc_events_used >= 256) c->assoc_events_used = 0;
            if (c->assoc_events_used >= (int)c->assoc_events.size()) {
                hipEvent_t e0, e1;
                HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
                c->assoc_events.emplace_back(e0, e1);
                c->assoc_event_info.emplace_back(nullptr, 0);
            }
            c->assoc_event_info[(size_t)c->assoc_events_used] = {assoc_name, bytes};
            ev = &c->assoc_events[c->assoc_events_used++];
        }
        launched[(size_t)first] = 1;
        if (n_seeded > 0) {                                          // this round's seeds for the contexts that take them from their target's direction image
            int nq_seed = 0;
            uint64_t sbytes = 0;
            for (int j = 0; j < n_seeded; j++) { nq_seed = std::max(nq_seed, SB.item[j].q_end - SB.item[j].q_begin); sbytes += 132ull * (uint64_t)(SB.item[j].q_end - SB.item[j].q_begin); }
            VELO_LAUNCH_T(c, "seed_batch_kernel", sbytes, seed_batch_kernel, dim3(cdiv(nq_seed, 256), n_seeded), dim3(256), 0, c->stream, SB);
        }
        bool all_queue = true;                                       // deferral is compiled in or out: all contexts of the launch or none
        for (int j = 0; j < k; j++) all_queue = all_queue && B.item[j].out.ask_list != nullptr;
        if (!all_queue) {
            for (int j = 0; j < k; j++) {
                AssocOut& o = B.item[j].out;
                o.ask_count = nullptr; o.ask_count_next = nullptr; o.ask_list = nullptr; o.ask_keys = nullptr; o.ask_rings = nullptr;
            }
            for (int i = b; i < b + m; i++) ctxs[i]->ask_clean[0] = ctxs[i]->ask_clean[1] = false;   // no launch clears a counter this round
        }
        if (all_direct) hipExtLaunchKernelGGL(assoc_direct_batch_kernel, dim3(nq_max, k), dim3(64), 0, c->stream, ev ? ev->first : nullptr, ev ? ev->second : nullptr, 0, B);
#ifdef VELO_DIAGNOSTICS
        else if (all_lane) hipExtLaunchKernelGGL(assoc_lane_batch_kernel, dim3(cdiv(gmax, 4), k), dim3(256), 0, c->stream, ev ? ev->first : nullptr, ev ? ev->second : nullptr, 0, B);
#endif
        else if (any_asker && all_queue) {
            hipExtLaunchKernelGGL((assoc_search_v5_batch_kernel<4, 5, false, 2, 2>), dim3(gmax, k), dim3(256), c->assoc_lds_pad, c->stream, ev ? ev->first : nullptr, nullptr, 0, B);
            hipExtLaunchKernelGGL(assoc_asker_batch_kernel, dim3(cdiv(gmax * 64, kAskChunk) + 8, k), dim3(64), 0, c->stream, nullptr, ev ? ev->second : nullptr, 0, B);
        }
        else if (any_asker) hipExtLaunchKernelGGL((assoc_search_v5_batch_kernel<4, 5, false, 2, true>), dim3(gmax, k), dim3(256), c->assoc_lds_pad, c->stream,
                                             ev ? ev->first : nullptr, ev ? ev->second : nullptr, 0, B);
        else hipExtLaunchKernelGGL((assoc_search_v5_batch_kernel<4, 5, false, 2, false>), dim3(gmax, k), dim3(256), c->assoc_lds_pad, c->stream,
                                   ev ? ev->first : nullptr, ev ? ev->second : nullptr, 0, B);
        HIP_TRY(hipGetLastError());
    }
    return VELO_OK;
}

// Several lock-step groups share the chip (shared_chip): a group's LM launches arrive while other groups' association kernels fill
// every CU.  The tube kernel's workgroup is 22 KB of LDS and 72 VGPRs per lane: seven fit a CU and leave 8 of 512 VGPRs per SIMD, so
// an LM workgroup (240 VGPRs, 33 KB) had to wait until FOUR of them had drained -- 21 us per LM launch alone, ~35 us in the mix, the
// largest single item of a step.  There the association workgroups get kAssocPadShared bytes of unused dynamic LDS (27.9 KB each: five
// per CU, 360 VGPRs per SIMD) and the LM launches use the lean instantiation (<= 152 VGPRs, 19 KB), which always fits beside them.
constexpr int kAssocPadShared = 5632;
// ---- velo_hint_next_frame: the next frame's loads behind the current chain ---------------------------------------------------------------
// Called by the thread that has just enqueued a chained call on c->stream and is about to wait for it.  Everything here is enqueued on that
// same stream, i.e. it runs when the chain has finished reading the old target and source.
static int preload_next_frame(velo_ctx* c, AdvJob* job) {
    if (!c->nf.hint_valid || c->nf.state == velo_ctx::NextFrame::LOADED || !c->have_source || !c->have_target || !c->src_bbox_valid) return VELO_OK;   // (without the source's box the promotion would wait for the chain)
    if (c->T.use_count() != 1) return VELO_OK;                         // a target other contexts hold cannot be given back after a repeat
    for (int r = 0; r < c->n_src_rings; r++) if (c->h_src_off[(size_t)r + 1] <= c->h_src_off[(size_t)r]) return VELO_OK;   // (a scan that cannot be promoted: the NEXT call says so, not this one)
    {
        const velo_scan_ref& h = c->nf.hint;                           // (an announcement the loaders would refuse is left to the call that brings it, too)
        if (h.n_rings <= 0 || !h.ring_offsets || h.ring_offsets[0] != 0) return VELO_OK;
        for (int r = 0; r < h.n_rings; r++) if (h.ring_offsets[r + 1] < h.ring_offsets[r]) return VELO_OK;
    }
    c->nf.hint_valid = false;
    c->nf.ref = c->nf.hint;
    const velo_scan_ref& r = c->nf.ref;
    // the loads of ALL contexts of the group in three launches (AdvJob), when the ring tables fit the kernel arguments and the general
    // loaders would build exactly this index (dense table, no direction image); else through the general loaders, launch by launch
    const int n_next = r.ring_offsets[r.n_rings];
    static const bool compress = dev_env("VELO_GRID_COMPRESS") && atoi(dev_env("VELO_GRID_COMPRESS")) != 0;
    static const bool no_batch = dev_env("VELO_ADV_BATCH") && atoi(dev_env("VELO_ADV_BATCH")) == 0;      // A/B (diagnostics build)
    const bool batched = job && !no_batch && !compress && !(c->dimg_seeds && c->warm_start) && c->n_src > 0 && c->n_src_rings <= kAdvRings && r.n_rings <= kAdvRings && n_next > 0;
    struct AdvScope { velo_ctx* c; ~AdvScope() { c->adv = nullptr; } } scope{c};
    if (batched) { std::memset(job, 0, sizeof(*job)); c->adv = job; }
    // the old target's cloud stays until this call is known to be good (a repeat needs the pair back) -- its BUFFER: the promotion swaps the
    // clouds' buffers (the old target's becomes the source's), and the source side then takes the spare one instead
    c->nf.undo_n = c->T->n_tgt; c->nf.undo_rings = c->T->n_tgt_rings; c->nf.undo_off = c->T->h_tgt_off;
    VELO_TRY(promote_begin(c));
    std::swap(c->src.p, c->nf.undo_cloud.p); std::swap(c->src.cap, c->nf.undo_cloud.cap);
    VELO_TRY(set_source_begin(c, r.xyz, r.stride_bytes, r.ring_offsets, r.n_rings, r.on_device & 1));
    VELO_TRY(target_finalize_end(c));
    VELO_TRY(source_finalize(c));
    c->nf.state = velo_ctx::NextFrame::LOADED;
    return batched ? 1 : VELO_OK;
}
// the collected loads of a group's contexts (jobs[i] valid where used[i]): three launches on the group's stream, then every box's way back
static int advance_launch(velo_ctx** ctxs, int n, const AdvJob* jobs, const std::vector<char>& used, hipStream_t bs) {
    velo_ctx* c0 = ctxs[0];
    for (int b = 0; b < n;) {
        AdvBatch B;
        std::memset(&B, 0, sizeof(B));
        int m = 0, gx_a = 0, gx_sc = 0, tiles[2] = {0, 0};
        uint64_t by_a = 0, by_s = 0, by_c = 0;
        velo_ctx* owner[kAdvJobs];
        for (; b < n && m < kAdvJobs; b++) {
            if (!used[(size_t)b]) continue;
            const AdvJob& J = jobs[b];
            B.job[m] = J; owner[m] = ctxs[b]; m++;
            gx_a = std::max(gx_a, J.nb_t + J.nb_pack + J.nb_q); gx_sc = std::max(gx_sc, J.nb_sc);
            tiles[J.nc >= kLbLargeFrom ? 1 : 0] = std::max(tiles[J.nc >= kLbLargeFrom ? 1 : 0], J.n_tiles);
            by_a += 40ull * (uint64_t)J.n_t + 28ull * (uint64_t)J.n_s + 32ull * (uint64_t)J.nq; by_s += 8ull * (uint64_t)J.nc; by_c += 44ull * (uint64_t)J.n_t;
        }
        if (m == 0) break;
        hipLaunchKernelGGL(advance_clear_kernel, dim3(256, m), dim3(256), 0, bs, B);
        VELO_LAUNCH_T(c0, "advance_ingest_kernel", by_a, advance_ingest_kernel, dim3(gx_a, m), dim3(256), 0, bs, B);
        for (int large = 0; large < 2; large++) {                      // (a group's tables are of one kind in practice: one launch)
            if (tiles[large] == 0) continue;
            AdvBatch S = B;
            for (int k = 0; k < m; k++) if ((S.job[k].nc >= kLbLargeFrom ? 1 : 0) != large) S.job[k].n_tiles = 0;
            if (large) VELO_LAUNCH_T(c0, "advance_scan_kernel", by_s, advance_scan_kernel<kLbItemsLarge>, dim3(tiles[1], m), dim3(kScanThreads), 0, bs, S);
            else VELO_LAUNCH_T(c0, "advance_scan_kernel", by_s, advance_scan_kernel<kLbItemsSmall>, dim3(tiles[0], m), dim3(kScanThreads), 0, bs, S);
        }
        VELO_LAUNCH_T(c0, "advance_scatter_kernel", by_c, advance_scatter_kernel, dim3(gx_sc, m), dim3(256), 0, bs, B);
        HIP_TRY(hipGetLastError());
        for (int k = 0; k < m; k++) {                                  // the boxes ride back on the stream (a LATER call, the next promotion, reads them)
            velo_ctx* c = owner[k];
            // (advance_scatter_kernel wrote the keys into the page-locked words itself)
            if (!c->src_bbox_ev) HIP_TRY(hipEventCreateWithFlags(&c->src_bbox_ev, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(c->src_bbox_ev, bs));
        }
    }
    return VELO_OK;
}
// every context of a group that announced its next frame: loads collected, then launched together
static int preload_group(velo_ctx** ctxs, int n, hipStream_t bs, bool* any_loaded) {
    std::vector<AdvJob> jobs((size_t)n);
    std::vector<char> used((size_t)n, 0);
    bool any = false;
    *any_loaded = false;
    for (int i = 0; i < n; i++) {
        const int st = preload_next_frame(ctxs[i], &jobs[(size_t)i]);
        if (st < 0) return st;
        used[(size_t)i] = st == 1; any = any || st == 1;
        *any_loaded = *any_loaded || ctxs[i]->nf.state == velo_ctx::NextFrame::LOADED;
    }
    if (any) VELO_TRY(advance_launch(ctxs, n, jobs.data(), used, bs));
    return VELO_OK;
}
// A preloaded context whose call has to be repeated: the pair it registered comes back -- the frame that was promoted (now the target's cloud)
// as source again, the kept cloud of the old target as target -- through the ordinary loaders.  The stream has been synchronised.
static int undo_preload(velo_ctx* c) {
    if (c->nf.state != velo_ctx::NextFrame::LOADED) return VELO_OK;
    c->nf.state = velo_ctx::NextFrame::NONE;
    const std::vector<int> src_off = c->T->h_tgt_off;                    // the promoted frame's rings (copied: the loaders rewrite the tables)
    const int src_rings = c->T->n_tgt_rings;
    VELO_TRY(set_source_begin(c, reinterpret_cast<const float*>(c->T->tgt.p), (int64_t)sizeof(float4), src_off.data(), src_rings, 1));
    { const int st = source_finalize(c); c->src_raw.on = false; if (st != VELO_OK) return st; }
    const std::vector<int> tgt_off = c->nf.undo_off;
    VELO_TRY(set_target_begin(c, reinterpret_cast<const float*>(c->nf.undo_cloud.p), (int64_t)sizeof(float4), tgt_off.data(), c->nf.undo_rings, 0, 0, 1));
    VELO_TRY(target_finalize_end(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VELO_OK;
}

// stagger_slot: this group's place among the lock-step groups of the call (0 = the first).  Groups that start a step together stay
// together: their association launches overlap (each at a fraction of the chip), then all of them are in their LM phases at once and no
// association kernel runs at all -- a quarter of the wall time with four groups.  Group k therefore enqueues its chain k x (its own
// previous chain's duration / kStaggerDiv) late: the phases stay apart for the whole step, and the chains shorten by more than the last
// group's delay (C2: +2-3 %; free-running groups, which drift apart by themselves, +6 %).  Results do not depend on it.
constexpr double kStaggerDiv = 27.0;             // a C2 chain of 1.9 ms: 70 us per slot (measured best among 40 / 70 / 100 / 130)
constexpr int kStaggerMinQueries = 16384;        // 256 association workgroups per context: a launch that takes the whole chip
static int f2f_batch_lockstep(velo_ctx** ctxs, int n, double* x, double* T, velo_summary* summaries, bool shared_chip = false, int stagger_slot = 0) {
    velo_ctx* c0 = ctxs[0];
    HIP_TRY(hipSetDevice(c0->device));
    for (int i = 0; i < n; i++) {
        if (ctxs[i]->nf.state == velo_ctx::NextFrame::LOADED) return fail(VELO_ERR_STATE, "context %d holds a frame loaded ahead (velo_hint_next_frame): the job that brings it must come first", i);
        if (ctxs[i]->nf.state == velo_ctx::NextFrame::CONSUMED) ctxs[i]->nf.state = velo_ctx::NextFrame::NONE;
    }
    struct HintEnd { velo_ctx** c; int n; ~HintEnd() { for (int i = 0; i < n; i++) c[i]->nf.hint_valid = false; } } hint_end{ctxs, n};   // an announcement is good for ONE call
    const bool lean = c0->lm_lean >= 0 ? c0->lm_lean != 0 : shared_chip;
    struct PadRestore { velo_ctx** c; int n; std::vector<int> old; ~PadRestore() { for (int i = 0; i < n; i++) c[i]->assoc_lds_pad = old[(size_t)i]; } } pad_restore{ctxs, n, {}};
    // A group with visual blocks: its LM launches carry the visual sweep.  With at most one block slot per thread (3 n_matches <= 64 x 256:
    // 5,461 matches) the LEAN kernel takes them (visual_sweep_one: 126 VGPRs like the plain lean kernel), pad as usual; with more, and
    // lean wanted, the visual sweep stays a launch of its own ahead of the lean one (its kernel carries the slot loop: 256 VGPRs).
    bool group_visual = false, one_slot = true;
    for (int i = 0; i < n; i++) {
        group_visual = group_visual || ctxs[i]->n_matches > 0;
        one_slot = one_slot && (int64_t)3 * ctxs[i]->n_matches <= (int64_t)kMaxVisBlocks * kEvalThreads;
    }
    const bool vis_in_launch = group_visual && c0->lm_vis_merged && c0->lm_fused && (!lean || one_slot);
    for (int i = 0; i < n; i++) {
        pad_restore.old.push_back(ctxs[i]->assoc_lds_pad);
        if (!ctxs[i]->assoc_lds_pad_fixed) ctxs[i]->assoc_lds_pad = (lean && shared_chip) ? kAssocPadShared : 0;
    }
    const velo_params P = c0->P;
    const LMParams Q = lm_params(P);
    // everything queued on the contexts' own streams (set_target / set_source) must be done before the shared stream uses it
    static const bool turn_trace = dev_env("VELO_TURN_TRACE") != nullptr;     // dev aid: the step boundary as the group's host thread sees it
    static thread_local std::chrono::steady_clock::time_point t_results;
    static thread_local bool t_results_valid = false;
    const auto t_entry = std::chrono::steady_clock::now();
    // (what is queued on the group's OWN stream -- a frame loaded ahead, velo_hint_next_frame -- is ordered before this call's launches by the stream itself)
    for (int i = 0; i < n; i++) if (ctxs[i]->stream != c0->stream) HIP_TRY(hipStreamSynchronize(ctxs[i]->stream));
    if (turn_trace && t_results_valid)
        fprintf(stderr, "[velo turn] results in -> next call's registration entered %.0f us; then waited %.0f us for the loads enqueued ahead\n",
                std::chrono::duration<double, std::micro>(t_entry - t_results).count(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_entry).count());
    std::vector<hipStream_t> own((size_t)n);
    for (int i = 0; i < n; i++) { own[(size_t)i] = ctxs[i]->stream; ctxs[i]->stream = c0->stream; }
    struct Restore { velo_ctx** c; std::vector<hipStream_t>& s; int n; ~Restore() { for (int i = 0; i < n; i++) c[i]->stream = s[(size_t)i]; } } restore{ctxs, own, n};
    hipStream_t bs = c0->stream;
    // pinned + device scratch
    const int rounds = P.f2f_iterations * P.icp_iterations;
    const size_t n_item_slots = (size_t)n * (size_t)std::max(rounds, 1);                // chain mode: one item array per round
    size_t vis_bytes = 0;                                            // chain mode with visual blocks: flags and block counts come back through page-locked memory too
    for (int i = 0; i < n; i++) vis_bytes += (((size_t)3 * (size_t)std::max(ctxs[i]->n_matches, 0) + 15) & ~(size_t)15) + sizeof(int) * 2 * VELO_MAX_STATS;
    const size_t need = n_item_slots * sizeof(LMBatchItem) + (size_t)n * (sizeof(LMState) + 8 * sizeof(double) + sizeof(SolveLog) * VELO_MAX_SOLVES + 8) + 16 + vis_bytes;
    if (c0->h_batch_bytes < need) {
        static const bool alloc_trace = dev_env("VELO_ALLOC_TRACE") != nullptr;
        if (alloc_trace) fprintf(stderr, "[velo alloc] pinned batch block: %zu -> %zu bytes\n", c0->h_batch_bytes, need);
        if (c0->h_batch) (void)hipHostFree(c0->h_batch);
        c0->h_batch = nullptr; c0->h_batch_bytes = 0;
        HIP_TRY(hipHostMalloc(&c0->h_batch, need, hipHostMallocDefault));
        c0->h_batch_bytes = need;
    }
    LMBatchItem* h_items = (LMBatchItem*)c0->h_batch;
    LMState* h_states = (LMState*)(h_items + n_item_slots);
    double* h_x = (double*)(h_states + n);
    SolveLog* h_logs = (SolveLog*)(h_x + 8 * (size_t)n);
    int* h_fail = (int*)(h_logs + (size_t)n * VELO_MAX_SOLVES);
    unsigned char* h_vis_pin = (unsigned char*)(((uintptr_t)(h_fail + 2 * n) + 15) & ~(uintptr_t)15);
    VELO_TRY(c0->batch_items.reserve(n_item_slots)); VELO_TRY(c0->batch_states.reserve((size_t)n)); VELO_TRY(c0->batch_x.reserve((size_t)8 * n));
    if (c0->batch_tickets.cap < (size_t)n) { VELO_TRY(c0->batch_tickets.reserve((size_t)n)); HIP_TRY(hipMemsetAsync(c0->batch_tickets.p, 0, sizeof(int) * c0->batch_tickets.cap, bs)); }

    std::vector<velo_summary> local((size_t)n);
    std::vector<velo_summary*> S((size_t)n);
    std::vector<std::array<double, 6>> xc((size_t)n);
    for (int i = 0; i < n; i++) {
        S[(size_t)i] = summaries ? summaries + i : &local[(size_t)i];
        std::memset(S[(size_t)i], 0, sizeof(velo_summary));
        S[(size_t)i]->n_target = ctxs[i]->T->n_tgt;
        ctxs[i]->assoc_events_used = 0;
        for (int k = 0; k < 6; k++) xc[(size_t)i][(size_t)k] = x[6 * (size_t)i + k];
    }
    const int max_iters = P.max_num_iterations + 1;
    std::vector<int> assoc_launched;

    // ---- chain mode (see frame_to_frame_chain): all rounds of the group as one chain of launches, one synchronisation ---------------
    bool chain = c0->chain && c0->lm_merged < 2 && rounds >= 1 && rounds <= VELO_MAX_SOLVES;
    for (int i = 0; i < n && chain; i++) {
        velo_ctx* c = ctxs[i];
        // (visual blocks: their gate runs on the device at the pose the device holds, like frame_to_frame_chain's; needs the fused sweep + step)
        chain = assoc_batchable(c) && c->P.enable_icp && !c->lm_trace_on && (c->n_matches == 0 || (c->lm_fused && !c->lm_persist && P.f2f_iterations <= VELO_MAX_STATS));
        if (chain) {
            if (query_list_stale(c)) VELO_TRY(build_query_list(c));
            int qb, qe; q_range(c, &qb, &qe);
            chain = qe > qb;
        }
    }
    if (chain) {
        const auto t_chain0 = std::chrono::steady_clock::now();
        {
            static const double div_env = dev_env("VELO_STAGGER_DIV") ? atof(dev_env("VELO_STAGGER_DIV")) : -1.0;      // A/B (diagnostics build): 0 = no stagger
            const double div = div_env >= 0.0 ? div_env : kStaggerDiv;
            int nq_min = 1 << 30;
            for (int i = 0; i < n; i++) nq_min = std::min(nq_min, ctxs[i]->n_q);
            // (only where a group's association launch fills the chip by itself: the sparse rounds of the reference's own constants -- 640 queries,
            //  a launch of 40 us on a tenth of the chip -- have nothing to keep apart, and a late start is all they get: C1 12.8 k against 13.2 k)
            if (shared_chip && stagger_slot > 0 && div > 0.0 && c0->last_chain_us > 0.0 && nq_min >= kStaggerMinQueries) {
                double mult = (double)stagger_slot;
                if (const char* pat = dev_env("VELO_STAGGER_PATTERN")) {     // A/B (diagnostics build): the slots' multipliers, e.g. "0,1,1,2"
                    double m[8] = {0, 1, 2, 3, 4, 5, 6, 7};
                    std::sscanf(pat, "%lf,%lf,%lf,%lf", &m[0], &m[1], &m[2], &m[3]);
                    mult = m[std::min(stagger_slot, 7)];
                }
                const auto until = t_chain0 + std::chrono::nanoseconds((long long)(1e3 * std::min(c0->last_chain_us / div, 400.0) * mult));
                while (std::chrono::steady_clock::now() < until) { }
            }
        }
        const auto t_chain1 = std::chrono::steady_clock::now();
        const bool fresh = c0->batch_pose.cap < (size_t)n;
        VELO_TRY(c0->batch_pose.reserve((size_t)n)); VELO_TRY(c0->batch_logs.reserve((size_t)n * VELO_MAX_SOLVES)); VELO_TRY(c0->batch_fail.reserve((size_t)n));
        if (fresh) { HIP_TRY(hipMemsetAsync(c0->batch_fail.p, 0, sizeof(int) * (size_t)n, bs)); HIP_TRY(hipMemsetAsync(c0->batch_pose.p, 0, sizeof(PoseRecord) * (size_t)n, bs)); }
        bool any_matches = false;
        std::vector<velo_ctx::TimingMark> tmarks((size_t)n);
        for (int i = 0; i < n; i++) {
            velo_ctx* c = ctxs[i];
            c->chain_calls++;
            c->timing_mark(&tmarks[(size_t)i]);
            if (c->n_matches > 0) {                                  // block / residual counts per f2f iteration come back at the end
                any_matches = true;
                VELO_TRY(c->vis_counts.reserve(2 * VELO_MAX_STATS));
                HIP_TRY(hipMemsetAsync(c->vis_counts.p, 0, sizeof(int) * 2 * VELO_MAX_STATS, bs));
                c->vflags_valid = true;
            } else VELO_TRY(do_build_visual(c, xc[(size_t)i].data(), false, 1, nullptr));      // no measurements: clears the host flags
            c->have_corr = false; c->last_n_valid = 0;
            for (int k = 0; k < 6; k++) h_x[8 * (size_t)i + k] = xc[(size_t)i][(size_t)k];
        }
        // (the start poses stay in the page-locked block: the first LM launch's workgroups -- and the visual gate -- read their six doubles from there,
        //  one queue operation less; nothing writes the block before the call's results are in)
        const double* d_x0 = h_x;
        // (A/B, measured: the LM launches on a high-priority stream of their own halve the throughput -- 1,530 vs 3,020 pairs/s: more than four
        //  active hardware queues are time-sliced, the same effect as GPU_MAX_HW_QUEUES=8)
        int r = 0;
        const bool iter_mode = c0->lm_iter && n <= 4 && !any_matches && c0->lm_fused && !c0->lm_persist;
        int iter_launches = 0;                                       // iter_mode: parity of every context's state / partial-row double buffer
        for (int iter = 1; iter <= P.f2f_iterations; iter++) {
            if (any_matches && n <= kGateJobs) {                     // residual-type choice + outlier gate of this iteration (velo.h:622-792), on the device:
                GateBatch Gb;                                        // the group's contexts in ONE launch (the contexts share their parameters: batch_can_lockstep)
                std::memset(&Gb, 0, sizeof(Gb));
                int k = 0, gx = 0;
                for (int i = 0; i < n; i++) {
                    velo_ctx* c = ctxs[i];
                    if (c->n_matches <= 0) continue;
                    Gb.x[k] = iter == 1 ? d_x0 + 8 * (size_t)i : c->state.p->x; Gb.m[k] = c->vm.p; Gb.flags[k] = c->vflags.p;
                    Gb.counts[k] = c->vis_counts.p + 2 * (iter - 1); Gb.n[k] = c->n_matches;
                    gx = std::max(gx, cdiv(c->n_matches, 128)); k++;
                }
                Gb.V = visual_params(c0->P); Gb.iter = iter;
                if (k > 0) hipLaunchKernelGGL(visual_gate_batch_kernel, dim3(gx, k), dim3(128), 0, bs, Gb);
            } else
            for (int i = 0; i < n && any_matches; i++) {
                velo_ctx* c = ctxs[i];
                if (c->n_matches <= 0) continue;
                hipLaunchKernelGGL(visual_gate_kernel, dim3(cdiv(c->n_matches, 128)), dim3(128), 0, bs, (const double*)(iter == 1 ? d_x0 + 8 * (size_t)i : c->state.p->x),
                                   visual_params(c->P), c->vm.p, c->n_matches, iter, c->vflags.p, c->vis_counts.p + 2 * (iter - 1));
            }
            for (int icp_iter = 0; icp_iter < P.icp_iterations; icp_iter++, r++) {
                VELO_TRY(do_associate_group(ctxs, n, xc, iter, assoc_launched, r == 0 ? nullptr : c0->batch_pose.p, c0->batch_fail.p));
                LMBatchItem* items_r = h_items + (size_t)r * n;
                int nb_max = 0, nbv_max = 0, K = 1;
                for (int i = 0; i < n; i++) {
                    velo_ctx* c = ctxs[i];
                    LMBatchItem& it = items_r[i];
                    it.A = eval_args(c, nullptr);
                    const EvalPlan E = eval_plan(it.A);
                    it.S = c->state.p; it.xd = r == 0 ? d_x0 + 8 * (size_t)i : nullptr;
                    it.n_valid = c->n_valid.p + c->nv_idx;
                    it.nb_icp = E.nb_icp; it.nb_vis = E.nb_vis; it.n_rows = E.total();
                    it.A.vis_row0 = E.nb_icp;
                    it.pose_out = c0->batch_pose.p + i; it.log = c0->batch_logs.p + (size_t)i * VELO_MAX_SOLVES + r;
                    nb_max = std::max(nb_max, E.nb_icp); nbv_max = std::max(nbv_max, E.nb_vis);
                    K = std::max(K, std::min(std::max(c->pred_evals[r], 1) + margin_for(c, r), max_iters));
                    S[(size_t)i]->assoc_kernel_launches += assoc_launched[(size_t)i];
                }
                const LMBatchItem* d_items = c0->batch_items.p + (size_t)r * n;
                // groups of up to four contexts with the fused sweep + step: the items ride in the kernel arguments; device copies are needed by
                // the kernels that take a pointer (the visual sweep, the two-launch path, the one-launch solve, the final state gather)
                // (round 5: the lean launch that carries the visual blocks takes them by value too -- six copies and their queue hand-overs less per call)
                const bool by_value_vis = n <= 4 && c0->lm_fused && !c0->lm_persist && nbv_max > 0 && vis_in_launch && lean;
                const bool by_value = (n <= 4 && c0->lm_fused && (!c0->lm_persist || c0->lm_persist == 2) && nbv_max == 0) || by_value_vis;
                // (the first round's copy used to serve the final state gather as well: chain_finish_kernel takes its pointers by value)
                if (!by_value || (r == 0 && n > kFinishJobs)) HIP_TRY(hipMemcpyAsync(c0->batch_items.p + (size_t)r * n, items_r, sizeof(LMBatchItem) * (size_t)n, hipMemcpyHostToDevice, bs));
                bool small = c0->small_solve != 0 && !iter_mode;        // every solve of the group is ONE single-workgroup launch (lm_solve_small_kernel's body)
                for (int i = 0; i < n; i++) small = small && items_r[i].n_rows >= 1 && items_r[i].n_rows <= kSmallRows;
                if (iter_mode) {                                        // K + 1 launches: the last one only advances the state over the K-th sweep's rows
                    LMBatchPackV pk;
                    std::memset(&pk, 0, sizeof(pk));
                    for (int i = 0; i < n; i++) pk.item[i] = items_r[i];
                    c0->lm_kernel_name = "lm_iter_batch_lean_kernel";
                    c0->lm_round_name[std::min(r, VELO_MAX_SOLVES - 1)] = c0->lm_kernel_name;
                    const size_t half = (size_t)(kMaxEvalBlocks + kMaxVisBlocks) * kNumAcc;
                    for (int k = 0; k <= K; k++, iter_launches++)
                        VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, lm_iter_batch_lean_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, Q, pk, iter_launches & 1, k == 0 ? 1 : 0, half);
                    HIP_TRY(hipGetLastError());
                    continue;
                }
                if (small) {
                    for (int b0 = 0; b0 < n; b0 += kItemsByValue) {     // (the items copied above serve lm_gather_states_kernel)
                        LMBatchPack pack;
                        std::memset(&pack, 0, sizeof(pack));
                        const int m = std::min(kItemsByValue, n - b0);
                        for (int i = 0; i < m; i++) pack.item[i] = items_r[b0 + i];
                        c0->lm_kernel_name = nbv_max > 0 ? "lm_solve_small_batch_kernel" : "lm_solve_small_icp_batch_kernel";
                        c0->lm_round_name[std::min(r, VELO_MAX_SOLVES - 1)] = c0->lm_kernel_name;
                        if (nbv_max > 0) VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, lm_solve_small_batch_kernel, dim3(m), dim3(kEvalThreads), 0, bs, Q, pack, max_iters + 1);
                        else VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, lm_solve_small_icp_batch_kernel, dim3(m), dim3(kEvalThreads), 0, bs, Q, pack, max_iters + 1);
                    }
                    HIP_TRY(hipGetLastError());
                    continue;
                }
                if (c0->lm_persist == 2 && c0->lm_fused && n <= 4 && nbv_max == 0) {   // all-gather form: one launch per solve, every workgroup steps itself
                    if (c0->ag_ctl.cap < (size_t)n) {
                        VELO_TRY(c0->ag_ctl.reserve((size_t)n));
                        HIP_TRY(hipMemsetAsync(c0->ag_ctl.p, 0, sizeof(AgCtl) * c0->ag_ctl.cap, bs));
                    }
                    LMBatchPackV pk;
                    std::memset(&pk, 0, sizeof(pk));
                    for (int i = 0; i < n; i++) pk.item[i] = items_r[i];
                    c0->lm_kernel_name = "lm_solve_ag_batch_kernel";
                    c0->lm_round_name[std::min(r, VELO_MAX_SOLVES - 1)] = c0->lm_kernel_name;
                    const size_t half = (size_t)(kMaxEvalBlocks + kMaxVisBlocks) * kNumAcc;
                    {
                        velo_ctx::TimedLaunch* tl = klog_slot(c0, c0->lm_kernel_name, 0);
                        const int le = velo_launch_lm_solve_ag(nb_max, n, (void*)bs, &Q, sizeof(Q), &pk, sizeof(pk), c0->ag_ctl.p, max_iters + 2, half, tl ? (void*)tl->a : nullptr, tl ? (void*)tl->b : nullptr);
                        if (le != 0) return fail(VELO_ERR_HIP, "lm_solve_ag launch: %s", hipGetErrorString((hipError_t)le));
                    }
                    HIP_TRY(hipGetLastError());
                    continue;
                }
                if (c0->lm_persist && c0->lm_fused) {                  // the whole solve of every context of the group: one launch, no prediction
                    if (c0->solve_ctl.cap < (size_t)n) {
                        VELO_TRY(c0->solve_ctl.reserve((size_t)n));
                        HIP_TRY(hipMemsetAsync(c0->solve_ctl.p, 0, sizeof(SolveCtl) * c0->solve_ctl.cap, bs));
                    }
                    c0->lm_kernel_name = "lm_solve_persist_batch_kernel";
                    c0->lm_round_name[std::min(r, VELO_MAX_SOLVES - 1)] = c0->lm_kernel_name;
                    const int wgs = c0->lm_persist_wgs > 0 ? std::min(c0->lm_persist_wgs, nb_max) : nb_max;
                    VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, lm_solve_persist_batch_kernel, dim3(wgs, n), dim3(kEvalThreads), 0, bs, Q, d_items, c0->solve_ctl.p, max_iters + 2);
                    HIP_TRY(hipGetLastError());
                    continue;
                }
                // (fused sweep + step: the first launch of a solve starts it as well -- no begin launch; with visual blocks their sweep runs as a
                //  launch of its own ahead of every fused one and reads the eval point from memory, so the begin launch stays)
                const bool vis_merged = nbv_max > 0 && vis_in_launch;
                const int first_fused = (nbv_max == 0 || vis_merged) ? 1 : 0;
                if (!c0->lm_fused || (nbv_max > 0 && !vis_merged)) VELO_LAUNCH_T(c0, "lm_begin_batch_kernel", 0, lm_begin_batch_kernel, dim3(n), dim3(64), 0, bs, d_items);
                if (c0->lm_fused) c0->lm_kernel_name = lean ? "eval_step_batch_lean_kernel" : "eval_step_batch_kernel";
                LMBatchPackV packv;
                if (by_value) {
                    std::memset(&packv, 0, sizeof(packv));
                    for (int i = 0; i < n; i++) packv.item[i] = items_r[i];
                    c0->lm_kernel_name = lean ? "eval_step_batch_lean_v_kernel" : "eval_step_batch_v_kernel";
                }
                for (int k = 0; k < K; k++) {
                    if (by_value_vis) {
                        int nb_all = 0;
                        for (int i = 0; i < n; i++) nb_all = std::max(nb_all, items_r[i].nb_icp + items_r[i].nb_vis);
                        c0->lm_kernel_name = "eval_step_batch_lean_vis_kernel";          // (one name for both argument forms: the same body)
                        VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_lean_vis_v_kernel, dim3(nb_all, n), dim3(kEvalThreads), 0, bs, Q, packv, c0->batch_tickets.p, k == 0 ? 1 : 0);
                        continue;
                    }
                    if (by_value) {
                        if (lean) VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_lean_v_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, Q, packv, c0->batch_tickets.p, k == 0 ? 1 : 0);
                        else VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_v_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, Q, packv, c0->batch_tickets.p, k == 0 ? 1 : 0);
                        continue;
                    }
                    if (vis_merged) {
                        int nb_all = 0;
                        for (int i = 0; i < n; i++) nb_all = std::max(nb_all, items_r[i].nb_icp + items_r[i].nb_vis);
                        c0->lm_kernel_name = lean ? "eval_step_batch_lean_vis_kernel" : "eval_step_batch_vis_kernel";
                        if (lean) VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_lean_vis_kernel, dim3(nb_all, n), dim3(kEvalThreads), 0, bs, Q, d_items, c0->batch_tickets.p, k == 0 ? 1 : 0);
                        else VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_vis_kernel, dim3(nb_all, n), dim3(kEvalThreads), 0, bs, Q, d_items, c0->batch_tickets.p, k == 0 ? 1 : 0);
                        continue;
                    }
                    if (c0->lm_fused) {
                        if (nbv_max > 0) VELO_LAUNCH_T(c0, "eval_visual_batch_kernel", 0, eval_visual_batch_kernel, dim3(nbv_max, n), dim3(kEvalThreads), 0, bs, d_items);
                        if (lean) VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_lean_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, Q, d_items, c0->batch_tickets.p, k == 0 ? first_fused : 0);
                        else VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, Q, d_items, c0->batch_tickets.p, k == 0 ? first_fused : 0);
                        continue;
                    }
                    hipLaunchKernelGGL(eval_icp_batch_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, d_items);
                    hipLaunchKernelGGL(lm_step_batch_kernel, dim3(n), dim3(256), 0, bs, Q, d_items);
                }
                HIP_TRY(hipGetLastError());
                c0->lm_round_name[std::min(r, VELO_MAX_SOLVES - 1)] = c0->lm_kernel_name;   // the kernel THIS round's evaluations ran in

            }
        }
        std::vector<int> h_vis((size_t)n * 2 * VELO_MAX_STATS, 0);
        std::vector<unsigned char*> pin_flags((size_t)n, nullptr);
        std::vector<int*> pin_counts((size_t)n, nullptr);
        {
            unsigned char* q = h_vis_pin;
            for (int i = 0; i < n && any_matches; i++) {
                const size_t fb = ((size_t)3 * (size_t)std::max(ctxs[i]->n_matches, 0) + 15) & ~(size_t)15;
                pin_flags[(size_t)i] = q; pin_counts[(size_t)i] = (int*)(q + fb);
                q += fb + sizeof(int) * 2 * VELO_MAX_STATS;
            }
        }
        if (n <= kFinishJobs) {
            // the call's results in ONE launch that writes the page-locked block itself (chain_finish_kernel): final states, solve logs, failure
            // flags, and the visual flags / block counts of the contexts that have matches
            ChainFinish F;
            std::memset(&F, 0, sizeof(F));
            for (int i = 0; i < n; i++) {
                velo_ctx* c = ctxs[i];
                F.S[i] = c->state.p + (iter_launches & 1);
                if (any_matches && c->n_matches > 0) {
                    F.vflags[i] = c->vflags.p; F.n_vflags[i] = 3 * c->n_matches; F.vis_counts[i] = c->vis_counts.p;
                    F.h_vflags[i] = pin_flags[(size_t)i]; F.h_vis_counts[i] = pin_counts[(size_t)i];
                }
            }
            F.logs = c0->batch_logs.p; F.fail = c0->batch_fail.p; F.h_states = h_states; F.h_logs = h_logs; F.h_fail = h_fail;
            F.n = n; F.n_logs = VELO_MAX_SOLVES; F.n_counts = 2 * VELO_MAX_STATS;
            hipLaunchKernelGGL(chain_finish_kernel, dim3(n), dim3(256), 0, bs, F);
            HIP_TRY(hipGetLastError());
        } else {
            hipLaunchKernelGGL(lm_gather_states_kernel, dim3(n), dim3(128), 0, bs, (const LMBatchItem*)c0->batch_items.p, c0->batch_states.p, iter_launches & 1);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h_states, c0->batch_states.p, sizeof(LMState) * (size_t)n, hipMemcpyDeviceToHost, bs));
            HIP_TRY(hipMemcpyAsync(h_logs, c0->batch_logs.p, sizeof(SolveLog) * (size_t)n * VELO_MAX_SOLVES, hipMemcpyDeviceToHost, bs));
            HIP_TRY(hipMemcpyAsync(h_fail, c0->batch_fail.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, bs));
            for (int i = 0; i < n && any_matches; i++) {             // (pageable destinations make each of these copies a staged, host-blocking one)
                velo_ctx* c = ctxs[i];
                if (c->n_matches <= 0) continue;
                HIP_TRY(hipMemcpyAsync(pin_flags[(size_t)i], c->vflags.p, (size_t)3 * c->n_matches, hipMemcpyDeviceToHost, bs));
                HIP_TRY(hipMemcpyAsync(pin_counts[(size_t)i], c->vis_counts.p, sizeof(int) * 2 * VELO_MAX_STATS, hipMemcpyDeviceToHost, bs));
            }
        }
        // what the summaries say about THIS call's scans, before a frame loaded ahead replaces them
        std::vector<int> nq_call((size_t)n), nt_call((size_t)n);
        bool hinted = false;
        for (int i = 0; i < n; i++) { nq_call[(size_t)i] = ctxs[i]->n_q; nt_call[(size_t)i] = ctxs[i]->T->n_tgt; hinted = hinted || ctxs[i]->nf.hint_valid; }
        if (hinted) {
            if (!c0->nf.call_done) HIP_TRY(hipEventCreateWithFlags(&c0->nf.call_done, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(c0->nf.call_done, bs));
        }
        for (int i = 0; i < n; i++) VELO_TRY(prefetch_issue(ctxs[i]));   // the next frames' uploads run under this chain (velo_hint_next_source)
        bool preloaded = false;
        static const bool ahead_trace = dev_env("VELO_AHEAD_TRACE") != nullptr;   // dev aid: how long the loads enqueued behind the chain take on the stream
        static thread_local hipEvent_t tr0 = nullptr, tr1 = nullptr;
        if (ahead_trace && hinted) { if (!tr0) { HIP_TRY(hipEventCreate(&tr0)); HIP_TRY(hipEventCreate(&tr1)); } HIP_TRY(hipEventRecord(tr0, bs)); }
        if (hinted) VELO_TRY(preload_group(ctxs, n, bs, &preloaded));    // velo_hint_next_frame: the next frame's promotion, ingest and index build behind this chain
        if (ahead_trace && hinted) HIP_TRY(hipEventRecord(tr1, bs));
        if (preloaded) HIP_TRY(hipEventSynchronize(c0->nf.call_done));   // the results are in; the next frame's loads are still running
        else HIP_TRY(hipStreamSynchronize(bs));
        c0->last_chain_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_chain1).count();   // (enqueue -> results in)
        if (turn_trace) { t_results = std::chrono::steady_clock::now(); t_results_valid = true; }
        if (ahead_trace && hinted) {
            const auto th = std::chrono::steady_clock::now();
            HIP_TRY(hipEventSynchronize(tr1));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, tr0, tr1));
            fprintf(stderr, "[velo ahead] %d contexts: loads behind the chain %.0f us on the stream; still running %.0f us after the results were in\n", n, 1e3 * ms,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th).count());
        }
        for (int i = 0; i < n && any_matches; i++) {
            velo_ctx* c = ctxs[i];
            if (c->n_matches <= 0) continue;
            c->h_vflags.assign(pin_flags[(size_t)i], pin_flags[(size_t)i] + (size_t)3 * c->n_matches);
            std::memcpy(h_vis.data() + (size_t)i * 2 * VELO_MAX_STATS, pin_counts[(size_t)i], sizeof(int) * 2 * VELO_MAX_STATS);
        }
        bool ok = true;
        for (int i = 0; i < n; i++) ok = ok && !h_fail[i] && h_states[i].done != 0;
        if (ok) {
            for (int i = 0; i < n; i++) {
                velo_ctx* c = ctxs[i];
                velo_summary* Si = S[(size_t)i];
                const uint64_t nq = (uint64_t)nq_call[(size_t)i];
                for (int k = 0; k < rounds; k++) {
                    const SolveLog& L = h_logs[(size_t)i * VELO_MAX_SOLVES + k];
                    Si->n_assoc_rounds++; Si->n_queries = nq_call[(size_t)i];
                    const uint64_t b_assoc = 12ull * nq + 12ull * (uint64_t)nt_call[(size_t)i] + 28ull * nq;
                    Si->assoc_bytes += b_assoc; Si->algorithmic_bytes += b_assoc;
                    velo_solve_summary ss;
                    std::memset(&ss, 0, sizeof(ss));
                    ss.termination = L.termination; ss.lm_iterations = L.iter; ss.evaluations = L.evals; ss.n_icp_valid = L.n_valid;
                    ss.initial_cost = L.initial_cost; ss.final_cost = L.final_cost;
                    if (c->n_matches > 0) {                          // the blocks of the f2f iteration this solve belongs to
                        const int it0 = std::min(k / std::max(P.icp_iterations, 1), VELO_MAX_STATS - 1);
                        ss.n_visual_blocks = h_vis[(size_t)i * 2 * VELO_MAX_STATS + 2 * it0]; ss.n_visual_residuals = h_vis[(size_t)i * 2 * VELO_MAX_STATS + 2 * it0 + 1];
                    }
                    note_evals(c, k, L.evals);
                    Si->eval_kernel_launches += L.evals;
                    Si->algorithmic_bytes += (uint64_t)L.evals * (36ull * (uint64_t)L.n_valid + 32ull * (uint64_t)ss.n_visual_blocks + 224ull);
                    if (c0->timing >= 2) {                           // the group's launches are logged on its first context
                        const char* rname = c0->lm_round_name[std::min(k, VELO_MAX_SOLVES - 1)];
                        kacc_add(c0, rname, 0.0, 0, 0, (uint64_t)L.evals * (36ull * (uint64_t)L.n_valid + 224ull));
                        if (ss.n_visual_blocks > 0) kacc_add(c0, vis_in_launch ? rname : "eval_visual_batch_kernel", 0.0, 0, 0, (uint64_t)L.evals * 32ull * (uint64_t)ss.n_visual_blocks);
                    }
                    Si->solves[Si->n_solves++] = ss;
                }
                c->last_n_valid = h_states[i].n_valid;
                if (c->timing) VELO_TRY(read_assoc_timing(c, Si));
                for (int k = 0; k < 6; k++) x[6 * (size_t)i + k] = h_states[i].x[k];
                if (T) velo_pose_vec_to_mat(x + 6 * (size_t)i, T + 16 * (size_t)i);
            }
            return VELO_OK;
        }
        // a solve outran its predicted launches: repeat the call with a host round trip per solve (same kernels, same results)
        if (preloaded) {                                                 // ... on the pair it registered: the frame loaded ahead goes back
            HIP_TRY(hipStreamSynchronize(bs));
            for (int i = 0; i < n; i++) VELO_TRY(undo_preload(ctxs[i]));
        }
        HIP_TRY(hipMemsetAsync(c0->batch_fail.p, 0, sizeof(int) * (size_t)n, bs));
        for (int i = 0; i < n; i++) {
            velo_ctx* c = ctxs[i];
            c->chain_misses++;
            c->timing_rewind(tmarks[(size_t)i]);
            c->nv_clean[0] = c->nv_clean[1] = false;                        // drained association launches did not clear the next round's counter
            c->ask_clean[0] = c->ask_clean[1] = false;
            note_miss(c);
            std::memset(S[(size_t)i], 0, sizeof(velo_summary));
            S[(size_t)i]->n_target = c->T->n_tgt;
            c->assoc_events_used = 0;
        }
    }
    for (int iter = 1; iter <= P.f2f_iterations; iter++) {                              // velo.h:616
        for (int i = 0; i < n; i++) {
            VELO_TRY(do_build_visual(ctxs[i], xc[(size_t)i].data(), false, iter, nullptr));   // velo.h:622-792
            ctxs[i]->have_corr = false; ctxs[i]->last_n_valid = 0;
        }
        for (int icp_iter = 0; icp_iter < P.icp_iterations; icp_iter++) {               // velo.h:800
            int nb_max = 0, nbv_max = 0, first_chunk = 2;
            VELO_TRY(do_associate_group(ctxs, n, xc, iter, assoc_launched));              // velo.h:806-894, on the shared stream
            for (int i = 0; i < n; i++) {
                velo_ctx* c = ctxs[i];
                int qb, qe;
                q_range(c, &qb, &qe);
                velo_summary* Si = S[(size_t)i];
                Si->n_assoc_rounds++; Si->n_queries = c->n_q;
                const uint64_t nq = (uint64_t)c->n_q;
                const uint64_t b_assoc = 12ull * nq + 12ull * (uint64_t)c->T->n_tgt + 28ull * nq;
                Si->assoc_bytes += b_assoc; Si->algorithmic_bytes += b_assoc;
                Si->assoc_kernel_launches += assoc_launched[(size_t)i];
                LMBatchItem& it = h_items[i];
                it.A = eval_args(c, nullptr);
                const EvalPlan E = eval_plan(it.A);
                it.S = c->state.p; it.xd = c0->batch_x.p + 8 * (size_t)i;
                it.n_valid = c->have_corr ? c->n_valid.p + c->nv_idx : nullptr;
                it.nb_icp = E.nb_icp; it.nb_vis = E.nb_vis; it.n_rows = E.total();
                it.A.vis_row0 = E.nb_icp;
                nb_max = std::max(nb_max, E.nb_icp); nbv_max = std::max(nbv_max, E.nb_vis);
                for (int k = 0; k < 6; k++) h_x[8 * (size_t)i + k] = xc[(size_t)i][(size_t)k];
                const int solve_idx = std::min(Si->n_solves, VELO_MAX_SOLVES - 1);
                first_chunk = std::max(first_chunk, std::min(c->pred_evals[solve_idx] + 1, max_iters));
            }
            HIP_TRY(hipMemcpyAsync(c0->batch_items.p, h_items, sizeof(LMBatchItem) * (size_t)n, hipMemcpyHostToDevice, bs));
            HIP_TRY(hipMemcpyAsync(c0->batch_x.p, h_x, sizeof(double) * 8 * (size_t)n, hipMemcpyHostToDevice, bs));
            // one launch per LM iteration for the whole group (lm_iter_batch_kernel) when every context has point-to-plane rows only
            // (measured with 8 pairs in flight: 2,040-2,120 pairs/s against 2,410-2,450 with sweep + step as two launches -- the redundant
            //  transitions keep 118 workgroups per context resident for 5 us longer, and at one wave per SIMD; hence VELO_LM_MERGED=2 only)
            bool merged = c0->lm_merged >= 2 && nbv_max == 0;
            bool fused = c0->lm_fused != 0;                                             // point-to-plane sweep + step in one launch, the last workgroup of a context steps
            for (int i = 0; i < n; i++) { merged = merged && h_items[i].nb_icp > 0; fused = fused && h_items[i].nb_icp > 0; }
            fused = fused && !merged;                                                   // (visual blocks: their sweep stays a launch of its own AHEAD of that one -- the
                                                                                        //  step then sums its rows too; riding in the same launch measured no gain)
            const size_t half = (size_t)(kMaxEvalBlocks + kMaxVisBlocks) * kNumAcc;
            int launched = 0, chunk = first_chunk + (merged ? 1 : 0);
            if (!merged) hipLaunchKernelGGL(lm_begin_batch_kernel, dim3(n), dim3(64), 0, bs, (const LMBatchItem*)c0->batch_items.p);
            for (;;) {                                                                  // one ceres::Solve per context, velo.h:897-902
                for (int k = 0; k < chunk; k++) {
                    if (merged) {
                        hipLaunchKernelGGL(lm_iter_batch_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, Q, (const LMBatchItem*)c0->batch_items.p, launched + k, half);
                        continue;
                    }
                    if (fused) {
                        c0->lm_kernel_name = lean ? "eval_step_batch_lean_kernel" : "eval_step_batch_kernel";
                        if (nbv_max > 0) VELO_LAUNCH_T(c0, "eval_visual_batch_kernel", 0, eval_visual_batch_kernel, dim3(nbv_max, n), dim3(kEvalThreads), 0, bs, (const LMBatchItem*)c0->batch_items.p);
                        if (lean) VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_lean_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, Q, (const LMBatchItem*)c0->batch_items.p, c0->batch_tickets.p, 0);
                        else VELO_LAUNCH_T(c0, c0->lm_kernel_name, 0, eval_step_batch_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, Q, (const LMBatchItem*)c0->batch_items.p, c0->batch_tickets.p, 0);
                        continue;
                    }
                    if (nb_max > 0) hipLaunchKernelGGL(eval_icp_batch_kernel, dim3(nb_max, n), dim3(kEvalThreads), 0, bs, (const LMBatchItem*)c0->batch_items.p);
                    if (nbv_max > 0) hipLaunchKernelGGL(eval_visual_batch_kernel, dim3(nbv_max, n), dim3(kEvalThreads), 0, bs, (const LMBatchItem*)c0->batch_items.p);
                    hipLaunchKernelGGL(lm_step_batch_kernel, dim3(n), dim3(256), 0, bs, Q, (const LMBatchItem*)c0->batch_items.p);
                }
                launched += chunk;
                hipLaunchKernelGGL(lm_gather_states_kernel, dim3(n), dim3(128), 0, bs, (const LMBatchItem*)c0->batch_items.p, c0->batch_states.p, merged ? (launched & 1) : 0);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(h_states, c0->batch_states.p, sizeof(LMState) * (size_t)n, hipMemcpyDeviceToHost, bs));
                HIP_TRY(hipStreamSynchronize(bs));
                bool all_done = true;
                for (int i = 0; i < n; i++) all_done = all_done && h_states[i].done != 0;
                if (all_done) break;
                if (launched > max_iters + 16) return fail(VELO_ERR_STATE, "LM did not terminate after %d sweeps", launched);
                chunk = 3;
            }
            for (int i = 0; i < n; i++) {
                velo_ctx* c = ctxs[i];
                const LMState& st = h_states[i];
                velo_summary* Si = S[(size_t)i];
                for (int k = 0; k < 6; k++) xc[(size_t)i][(size_t)k] = st.x[k];
                velo_solve_summary ss;
                std::memset(&ss, 0, sizeof(ss));
                ss.termination = st.termination; ss.lm_iterations = st.iter; ss.evaluations = st.evals;
                c->last_n_valid = st.n_valid; ss.n_icp_valid = st.n_valid;
                visual_counts(c, &ss.n_visual_blocks, &ss.n_visual_residuals);
                ss.initial_cost = st.initial_cost; ss.final_cost = st.cost;
                const int solve_idx = std::min(Si->n_solves, VELO_MAX_SOLVES - 1);
                note_evals(c, solve_idx, ss.evaluations);
                Si->eval_kernel_launches += st.evals;
                Si->algorithmic_bytes += (uint64_t)ss.evaluations * (36ull * (uint64_t)ss.n_icp_valid + 32ull * (uint64_t)ss.n_visual_blocks + 224ull);
                if (c0->timing >= 2 && fused) {                     // the evaluations' bytes: point-to-plane rows to the fused sweep + step, visual blocks to their sweep
                    kacc_add(c0, c0->lm_kernel_name, 0.0, 0, 0, (uint64_t)ss.evaluations * (36ull * (uint64_t)ss.n_icp_valid + 224ull));
                    if (ss.n_visual_blocks > 0) kacc_add(c0, "eval_visual_batch_kernel", 0.0, 0, 0, (uint64_t)ss.evaluations * 32ull * (uint64_t)ss.n_visual_blocks);
                }
                if (Si->n_solves < VELO_MAX_SOLVES) Si->solves[Si->n_solves] = ss;
                Si->n_solves++;
            }
        }
    }
    for (int i = 0; i < n; i++) {
        velo_ctx* c = ctxs[i];
        if (c->timing) {
            VELO_TRY(read_assoc_timing(c, S[(size_t)i]));
        }
        for (int k = 0; k < 6; k++) x[6 * (size_t)i + k] = xc[(size_t)i][(size_t)k];
        if (T) velo_pose_vec_to_mat(x + 6 * (size_t)i, T + 16 * (size_t)i);
    }
    return VELO_OK;
}

// upload (optional) + register: the scans of job i go into context i (velo_set_target / velo_set_source semantics), then the batch runs
struct JobVisual { const velo_match* const* m = nullptr; const int32_t* n = nullptr; };   // per-job matches of velo_register_batch_visual (or none)
static int load_job_visual(velo_ctx* c, const JobVisual& V, int i, hipStream_t on = nullptr) {
    if (!V.n) return VELO_OK;
    if (V.n[i] < 0 || (V.n[i] > 0 && (!V.m || !V.m[i]))) return fail(VELO_ERR_INVALID, "job %d: bad visual arguments", i);
    return set_visual_impl(c, V.n[i] > 0 ? V.m[i] : nullptr, V.n[i], false, on);
}
// A context that loaded an announced frame ahead (velo_hint_next_frame) holds it already: the job that brings exactly that frame -- a promoted
// target and the announced source -- loads nothing; any other job is an error (the context is one frame ahead of what the caller thinks).
// -> 1: the job's scans are in, 0: load as usual, < 0: status
static int take_preloaded(velo_ctx* c, const velo_scan_ref* tg, const velo_scan_ref* sr) {
    if (c->nf.state == velo_ctx::NextFrame::CONSUMED) c->nf.state = velo_ctx::NextFrame::NONE;   // (left by a call that failed between its loads and its registration)
    if (c->nf.state != velo_ctx::NextFrame::LOADED) return 0;
    const velo_scan_ref& r = c->nf.ref;
    // (the same cloud: address, stride, residence and ring table -- by content, as the context holds it: two descriptors of one frame match)
    const bool match = tg && (tg->on_device & VELO_SCAN_PROMOTE) && sr && sr->xyz == r.xyz && sr->stride_bytes == r.stride_bytes && sr->ring_offsets &&
                       sr->n_rings == r.n_rings && (sr->on_device & 1) == (r.on_device & 1) && (int)c->h_src_off.size() == r.n_rings + 1 &&
                       std::equal(c->h_src_off.begin(), c->h_src_off.end(), sr->ring_offsets);
    if (!match) return fail(VELO_ERR_STATE, "the frame announced with velo_hint_next_frame has been loaded ahead: the next job must promote the source and bring that frame");
    c->nf.state = velo_ctx::NextFrame::CONSUMED;
    return 1;
}
static int load_job(velo_ctx* c, const velo_scan_ref* tg, const velo_scan_ref* sr) {
    { const int t = take_preloaded(c, tg, sr); if (t < 0) return t; if (t > 0) return VELO_OK; }
    if (tg && (tg->on_device & VELO_SCAN_PROMOTE)) VELO_TRY(velo_source_to_target(c));
    else if (tg) VELO_TRY(velo_set_target(c, tg->xyz, tg->stride_bytes, tg->ring_offsets, tg->n_rings, tg->on_device & 1));
    if (sr) VELO_TRY(velo_set_source(c, sr->xyz, sr->stride_bytes, sr->ring_offsets, sr->n_rings, sr->on_device & 1));
    return VELO_OK;
}
// the same in two halves: everything that needs no answer from the device (uploads, ring tables, the bounding-box request), then the rest
static int load_job_begin(velo_ctx* c, const velo_scan_ref* tg, const velo_scan_ref* sr) {
    { const int t = take_preloaded(c, tg, sr); if (t < 0) return t; if (t > 0) return VELO_OK; }
    if (tg && (tg->on_device & VELO_SCAN_PROMOTE)) VELO_TRY(promote_begin(c));
    else if (tg) VELO_TRY(set_target_begin(c, tg->xyz, tg->stride_bytes, tg->ring_offsets, tg->n_rings, 0, 0, tg->on_device & 1));
    if (sr) VELO_TRY(set_source_begin(c, sr->xyz, sr->stride_bytes, sr->ring_offsets, sr->n_rings, sr->on_device & 1));
    return VELO_OK;
}
static int load_job_end(velo_ctx* c, bool tg, bool sr) {
    if (c->nf.state == velo_ctx::NextFrame::CONSUMED) return VELO_OK;       // loaded ahead, one call ago
    if (tg) VELO_TRY(target_finalize_end(c));
    if (sr) VELO_TRY(source_finalize(c));
    return VELO_OK;
}

static int batch_impl(velo_ctx** ctxs, int32_t n, const velo_scan_ref* targets, const velo_scan_ref* sources, double* x, double* T, velo_summary* summaries,
                      JobVisual V = JobVisual()) {
    if (!ctxs || n < 0 || (n > 0 && !x)) return fail(VELO_ERR_INVALID, "bad batch arguments");
    for (int i = 0; i < n; i++) {                                    // one registration per context: a context listed twice would race with itself
        if (!ctxs[i]) return fail(VELO_ERR_INVALID, "batch entry %d is null", i);
        for (int j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) return fail(VELO_ERR_INVALID, "batch entries %d and %d are the same context", j, i);
    }
    struct BatchLoad {                                               // marks the contexts while their scans are loaded (index sizing, build_grid)
        velo_ctx** c; int n;
        BatchLoad(velo_ctx** c_, int n_) : c(c_), n(n_) { for (int i = 0; i < n; i++) c[i]->batch_load = n >= 2; }
        // (src_raw points into the caller's buffer or the staging area and is only good inside this call: a failed load must not leave it armed)
        ~BatchLoad() { for (int i = 0; i < n; i++) { c[i]->batch_load = false; c[i]->src_raw.on = false; } }
    } batch_load(ctxs, n);
    // Targets flagged VELO_SCAN_SHARED with identical descriptors (scan-to-map: many scans against one map) are loaded and indexed
    // ONCE, by the first job that names them; the other jobs' contexts take that target by reference (velo_share_target).
    std::vector<velo_scan_ref> tgt_local;
    if (targets) {
        bool any = false;
        for (int i = 0; i < n; i++) any = any || (targets[i].on_device & VELO_SCAN_SHARED) != 0;
        if (any) {
            tgt_local.assign(targets, targets + n);
            auto owner_of = [&](int i) {
                for (int j = 0; j < i; j++) {
                    const velo_scan_ref &a = targets[i], &b = targets[j];
                    if ((b.on_device & VELO_SCAN_SHARED) && a.xyz == b.xyz && a.stride_bytes == b.stride_bytes && a.ring_offsets == b.ring_offsets &&
                        a.n_rings == b.n_rings && a.on_device == b.on_device && ctxs[i]->device == ctxs[j]->device) return j;
                }
                return -1;
            };
            // The sharers let go of what they hold from their owner BEFORE it loads: a target other contexts still hold is left to them
            // (own_target), so the owner of a map shared in the last call would allocate a whole new index every call -- seven buffers
            // of up to 90 MB, and as many frees when the last sharer moves on -- instead of rebuilding in place.
            for (int i = 0; i < n; i++) {
                if (!(targets[i].on_device & VELO_SCAN_SHARED)) continue;
                const int owner = owner_of(i);
                if (owner >= 0 && ctxs[i]->T && ctxs[i]->T == ctxs[owner]->T) {
                    HIP_TRY(hipSetDevice(ctxs[i]->device));
                    HIP_TRY(hipStreamSynchronize(ctxs[i]->stream));
                    // never a null T (every other path assumes one): an empty target of its own has the same effect on the owner's use_count
                    ctxs[i]->T = std::make_shared<TargetData>(); ctxs[i]->have_target = false; ctxs[i]->have_corr = false; ctxs[i]->have_partials = false;
                }
            }
            for (int i = 0; i < n; i++) {
                if (!(targets[i].on_device & VELO_SCAN_SHARED)) continue;
                int owner = -1;
                for (int j = 0; j < i && owner < 0; j++) {
                    const velo_scan_ref &a = targets[i], &b = targets[j];
                    if ((b.on_device & VELO_SCAN_SHARED) && a.xyz == b.xyz && a.stride_bytes == b.stride_bytes && a.ring_offsets == b.ring_offsets &&
                        a.n_rings == b.n_rings && a.on_device == b.on_device && ctxs[i]->device == ctxs[j]->device) owner = j;
                }
                if (owner < 0) VELO_TRY(load_job(ctxs[i], &targets[i], nullptr));
                else VELO_TRY(velo_share_target(ctxs[i], ctxs[owner]));
                tgt_local[(size_t)i].xyz = nullptr; tgt_local[(size_t)i].n_rings = -1;      // marks "already loaded"
            }
        }
    }
    auto target_of = [&](int i) -> const velo_scan_ref* {
        if (!targets) return nullptr;
        if (!tgt_local.empty()) return tgt_local[(size_t)i].n_rings < 0 ? nullptr : &tgt_local[(size_t)i];
        return targets + i;
    };
    if (n == 1) {                                                     // one job: the single-pair path (one chain of launches, one-launch LM iterations)
        VELO_TRY(load_job_visual(ctxs[0], V, 0));
        VELO_TRY(load_job(ctxs[0], target_of(0), sources));
        return velo_frame_to_frame(ctxs[0], x, T, summaries);
    }
    // (the jobs' matches are loaded next to their scans, on the thread that drives the context; the copies stay queued on the contexts'
    //  streams, which every path below synchronises or continues on)
    if (batch_can_lockstep(ctxs, n, targets != nullptr, sources != nullptr)) {
        // G lock-step groups, one host thread and one stream each: while one group is in its (chip-filling) association
        // launches or waits for a status copy, another group's LM launches run -- the groups hide each other's bubbles
        // Measured on C2 (pairs/s, 3 runs each): 8 contexts: 1 group 1,425, 2 groups 1,790-1,920, 4 groups 1,990-2,200, one thread per
        // context 1,600; 16 contexts: 2 groups 2,010-2,110, 4 groups 1,420-1,510 (four association kernels interleave), 8 groups 1,740-1,780.
        static const int groups_env = getenv("VELO_BATCH_GROUPS") ? std::max(atoi(getenv("VELO_BATCH_GROUPS")), 1) : 0;
        // (round 2, pairs/s by contexts / groups: 9: 3 groups 3,070, 4 groups 2,820; 10: 2 / 3 / 4 / 5 groups 1,990 / 2,900 / 2,490 / 2,590;
        //  11: 3 groups 2,490, 4 groups 2,700 -- a launch serves up to four contexts, so groups of five split theirs 4 + 1)
        const int G = groups_env > 0 ? std::min(groups_env, n / 2) : (n >= 12 ? 2 : ((n == 9 || n == 10) ? 3 : std::min(4, n / 2)));
        if (G <= 1) {
            for (int i = 0; i < n; i++) { VELO_TRY(load_job_visual(ctxs[i], V, i, ctxs[0]->stream)); VELO_TRY(load_job(ctxs[i], target_of(i), sources ? sources + i : nullptr)); }
            return f2f_batch_lockstep(ctxs, n, x, T, summaries);
        }
        std::vector<int> gst((size_t)G, VELO_OK);
        std::vector<std::string> gerr((size_t)G);
        static const bool batch_trace = dev_env("VELO_BATCH_TRACE") != nullptr;       // dev aid: host-side timeline of every group to stderr
        const auto t_call = std::chrono::steady_clock::now();
        auto run_group = [&](int gi) {
            const int b = (int)((int64_t)n * gi / G), e = (int)((int64_t)n * (gi + 1) / G);
            const auto t0 = std::chrono::steady_clock::now();
            // this group's index builds, then its registrations: no barrier across groups, so one group's association launches
            // run under another group's index builds.  (Helper threads that load a group's contexts in parallel were measured
            // slower, 2.32-2.34 k vs 2.42-2.47 k pairs/s: more host threads contending for the runtime's submission path.)
            // (in two passes: all contexts' uploads and bounding-box requests are in flight before the first context waits for its answer)
            for (int i = b; i < e; i++) {
                int st = load_job_visual(ctxs[i], V, i, ctxs[b]->stream);   // (on the group's stream: the registration runs there)
                if (st == VELO_OK) st = load_job_begin(ctxs[i], target_of(i), sources ? sources + i : nullptr);
                if (st != VELO_OK) { gst[(size_t)gi] = st; gerr[(size_t)gi] = g_err; return; }
            }
            const auto t1 = std::chrono::steady_clock::now();
            for (int i = b; i < e; i++) {
                const int st = load_job_end(ctxs[i], target_of(i) != nullptr, sources != nullptr);
                if (st != VELO_OK) { gst[(size_t)gi] = st; gerr[(size_t)gi] = g_err; return; }
            }
            const auto t2 = std::chrono::steady_clock::now();
            gst[(size_t)gi] = f2f_batch_lockstep(ctxs + b, e - b, x + 6 * (size_t)b, T ? T + 16 * (size_t)b : nullptr, summaries ? summaries + b : nullptr, true, gi);
            if (gst[(size_t)gi] != VELO_OK) gerr[(size_t)gi] = g_err;
            if (batch_trace) {
                const auto t3 = std::chrono::steady_clock::now();
                auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point bb) { return std::chrono::duration<double, std::micro>(bb - a).count(); };
                fprintf(stderr, "[velo batch] group %d: start +%.0f us, loads begun %.0f, loads ended %.0f, registrations %.0f us\n", gi, us(t_call, t0), us(t0, t1), us(t1, t2), us(t2, t3));
            }
        };
        WorkerPool::instance().run(G, run_group);                // the calling thread drives the first group itself, resident workers the others
        for (int gi = 0; gi < G; gi++) if (gst[(size_t)gi] != VELO_OK) { g_err = gerr[(size_t)gi]; return gst[(size_t)gi]; }
        return VELO_OK;
    }
    std::vector<int> status((size_t)n, VELO_OK);
    std::vector<std::string> errs((size_t)n);
    WorkerPool::instance().run(n, [&](int i) {
        status[i] = load_job_visual(ctxs[i], V, i);
        if (status[i] == VELO_OK) status[i] = load_job(ctxs[i], target_of(i), sources ? sources + i : nullptr);
        if (status[i] == VELO_OK) status[i] = velo_frame_to_frame(ctxs[i], x + 6 * (size_t)i, T ? T + 16 * (size_t)i : nullptr, summaries ? summaries + i : nullptr);
        if (status[i] != VELO_OK) errs[i] = g_err;
    });
    for (int i = 0; i < n; i++) if (status[i] != VELO_OK) { g_err = errs[i]; return status[i]; }
    return VELO_OK;
}

int velo_frame_to_frame_batch(velo_ctx** ctxs, int32_t n, double* x, double* T, velo_summary* summaries) {
    return batch_impl(ctxs, n, nullptr, nullptr, x, T, summaries);
}

// The drive loop of n sequences for n_frames frames in ONE call (main.cpp:207-413 for n sequences; the reference runs its sequences as
// independent processes, run.fish:2): every lock-step group walks ITS drives' frames on its own host thread -- promote the frame the
// contexts hold to target (sd_prev, main.cpp:233,380), load the new frame, register from the constant-velocity guess, hand the pose over
// (velo_pose_handoff: main.cpp:311-331,408) -- and starts frame f + 1 as soon as ITS frame f is done.  No barrier across the groups
// between frames: one group's synchronisation, result read-back and hand-off run under the other groups' chains, a group whose solves
// took fewer iterations does not wait for the slowest one, and the groups' chip-filling association launches drift apart instead of
// meeting at every step.  Per pair the work and the results are those of n_frames velo_register_batch[_visual] calls with
// VELO_SCAN_PROMOTE targets followed by velo_pose_handoff (tests compare them bit for bit).
static int sequences_impl(velo_ctx** ctxs, int32_t n, int32_t n_frames, const velo_scan_ref* frames, const velo_match* const* matches, const int32_t* n_matches,
                          double* poses, double* x_guess, double* x_out, double* T_out, velo_summary* summaries, int32_t flags) {
    if (!ctxs || n < 0 || n_frames < 0 || (n > 0 && n_frames > 0 && (!frames || !poses || !x_guess || !x_out))) return fail(VELO_ERR_INVALID, "bad sequence arguments");
    if (matches && !n_matches) return fail(VELO_ERR_INVALID, "n_matches is null");
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return fail(VELO_ERR_INVALID, "sequence entry %d is null", i);
        for (int j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) return fail(VELO_ERR_INVALID, "sequence entries %d and %d are the same context", j, i);
        if (!ctxs[i]->have_source) return fail(VELO_ERR_STATE, "sequence %d: the context holds no frame to start from (velo_set_source)", i);
    }
    if (n == 0 || n_frames == 0) return VELO_OK;
    struct BatchLoad {
        velo_ctx** c; int n;
        BatchLoad(velo_ctx** c_, int n_) : c(c_), n(n_) { for (int i = 0; i < n; i++) c[i]->batch_load = n >= 2; }
        ~BatchLoad() { for (int i = 0; i < n; i++) { c[i]->batch_load = false; c[i]->src_raw.on = false; } }
    } batch_load(ctxs, n);
    velo_scan_ref promote;
    std::memset(&promote, 0, sizeof(promote));
    promote.stride_bytes = 16; promote.on_device = VELO_SCAN_ON_DEVICE | VELO_SCAN_PROMOTE;
    const bool lockstep = n >= 2 && batch_can_lockstep(ctxs, n, true, true);
    static const int groups_env = getenv("VELO_BATCH_GROUPS") ? std::max(atoi(getenv("VELO_BATCH_GROUPS")), 1) : 0;
    const int G = !lockstep ? n : (groups_env > 0 ? std::max(1, std::min(groups_env, n / 2)) : (n >= 12 ? 2 : ((n == 9 || n == 10) ? 3 : std::max(1, std::min(4, n / 2)))));
    std::vector<int> gst((size_t)G, VELO_OK);
    std::vector<std::string> gerr((size_t)G);
    std::atomic<bool> stop{false};
    // VELO_SEQ_LOCKSTEP: the groups start every frame together (what a caller that makes one velo_register_batch call per frame gets, without
    // the caller in the loop): a counting barrier between frames, generation by generation
    const bool lockstep_frames = (flags & VELO_SEQ_LOCKSTEP) != 0 && G > 1;
    std::mutex bar_m;
    std::condition_variable bar_cv;
    int bar_count = 0, bar_gen = 0;
    auto frame_barrier = [&]() {
        std::unique_lock<std::mutex> lk(bar_m);
        const int gen = bar_gen;
        if (++bar_count == G) { bar_count = 0; bar_gen++; bar_cv.notify_all(); }
        else bar_cv.wait(lk, [&]() { return bar_gen != gen; });
    };
    auto run_group = [&](int gi) {
        const int b = (int)((int64_t)n * gi / G), e = (int)((int64_t)n * (gi + 1) / G), m = e - b;
        std::vector<double> Tl((size_t)16 * m), xl((size_t)6 * m);
        // (a failing group keeps meeting the others at the barrier until the last frame: nobody waits for a group that has left)
        int failed = VELO_OK;
        auto bail = [&](int st) { gst[(size_t)gi] = st; gerr[(size_t)gi] = g_err; stop.store(true); failed = st; };
        static const bool seq_trace = dev_env("VELO_SEQ_TRACE") != nullptr;        // dev aid: where a group's host thread spends a frame
        double t_load = 0.0, t_reg = 0.0, t_hand = 0.0;
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point bb) { return std::chrono::duration<double, std::micro>(bb - a).count(); };
        const auto t_begin = now();
        for (int f = 0; f < n_frames && (lockstep_frames || !stop.load()); f++) {
            if (lockstep_frames) { if (f > 0) frame_barrier(); if (failed != VELO_OK || stop.load()) continue; }
            const auto t0 = now();
            const velo_scan_ref* fr = frames + (size_t)f * n;
            JobVisual V;
            if (matches) { V.m = matches + (size_t)f * n; V.n = n_matches + (size_t)f * n; }
            velo_summary* Sf = summaries ? summaries + (size_t)f * n + b : nullptr;
            for (int i = 0; i < m; i++) for (int k = 0; k < 6; k++) xl[(size_t)6 * i + k] = x_guess[(size_t)6 * (b + i) + k];
            int st = VELO_OK;
            // the next frame of the group's drives: uploaded under this frame's chain, promoted / ingested / indexed behind it (velo_hint_next_frame)
            // A/B (diagnostics build, VELO_LATE_PRELOAD=1): every other group loads its frame at the START of its step (the same three launches)
            // instead of behind the previous step's chain -- its chains run half a round out of phase with the other groups' at no extra work
            static const int late_env = dev_env("VELO_LATE_PRELOAD") ? atoi(dev_env("VELO_LATE_PRELOAD")) : 0;
            const bool late_group = late_env != 0 && lockstep_frames && (gi & 1) && m > 1;
            if (!late_group && (f + 1 < n_frames || (flags & VELO_SEQ_ANNOUNCE))) for (int i = b; i < e; i++) (void)velo_hint_next_frame(ctxs[i], frames + (size_t)(f + 1) * n + i);
            if (late_group) {
                std::vector<hipStream_t> own((size_t)m);
                for (int i = 0; i < m; i++) { own[(size_t)i] = ctxs[b + i]->stream; ctxs[b + i]->stream = ctxs[b]->stream; ctxs[b + i]->nf.hint = fr[b + i]; ctxs[b + i]->nf.hint_valid = true; }
                bool any = false;
                st = preload_group(ctxs + b, m, ctxs[b]->stream, &any);
                for (int i = 0; i < m; i++) { ctxs[b + i]->stream = own[(size_t)i]; ctxs[b + i]->nf.hint_valid = false; }
            }
            if (st != VELO_OK) { bail(st); if (!lockstep_frames) return; continue; }
            if (m == 1) {                                            // a drive of its own: the single-pair path
                st = load_job_visual(ctxs[b], V, b);
                if (st == VELO_OK) st = load_job(ctxs[b], &promote, fr + b);
                if (st == VELO_OK) st = velo_frame_to_frame(ctxs[b], xl.data(), Tl.data(), Sf);
            } else {
                for (int i = b; i < e && st == VELO_OK; i++) {
                    st = load_job_visual(ctxs[i], V, i, ctxs[b]->stream);   // (on the group's stream: the registration runs there)
                    if (st == VELO_OK) st = load_job_begin(ctxs[i], &promote, fr + i);
                }
                for (int i = b; i < e && st == VELO_OK; i++) st = load_job_end(ctxs[i], true, true);
                const auto t1 = now();
                t_load += us(t0, t1);
                {   // A/B (diagnostics build): every other group starts its chain late -- do the groups' association phases stay apart?
                    static const int stagger_us = dev_env("VELO_GROUP_STAGGER_US") ? atoi(dev_env("VELO_GROUP_STAGGER_US")) : 0;
                    static const int stagger_mode = dev_env("VELO_GROUP_STAGGER_MODE") ? atoi(dev_env("VELO_GROUP_STAGGER_MODE")) : 0;   // 0: odd groups; 1: the upper half; 2: gi * us
                    const int mult = stagger_mode == 2 ? gi : (stagger_mode == 1 ? (gi >= G / 2 ? 1 : 0) : (gi & 1));
                    if (stagger_us > 0 && mult > 0 && (lockstep_frames || f == 0)) {   // (free-running groups: once, at the first frame)
                        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds((long long)stagger_us * mult);
                        while (std::chrono::steady_clock::now() < until) { }
                    }
                }
                if (st == VELO_OK) st = f2f_batch_lockstep(ctxs + b, m, xl.data(), Tl.data(), Sf, G > 1, lockstep_frames ? gi : 0);
                t_reg += us(t1, now());
            }
            if (st != VELO_OK) { bail(st); if (!lockstep_frames) return; continue; }
            const auto t2 = now();
            std::memcpy(x_out + ((size_t)f * n + b) * 6, xl.data(), sizeof(double) * 6 * (size_t)m);
            if (T_out) std::memcpy(T_out + ((size_t)f * n + b) * 16, Tl.data(), sizeof(double) * 16 * (size_t)m);
            st = velo_pose_handoff(m, poses + (size_t)16 * b, Tl.data(), x_guess + (size_t)6 * b);      // main.cpp:408, 311-331
            if (st != VELO_OK) { bail(st); if (!lockstep_frames) return; continue; }
            t_hand += us(t2, now());
        }
        if (seq_trace) fprintf(stderr, "[velo seq] group %d: %d frames in %.0f us: loads %.0f, registrations %.0f, hand-over %.0f us per frame\n", gi, n_frames,
                               us(t_begin, now()), t_load / n_frames, t_reg / n_frames, t_hand / n_frames);
    };
    WorkerPool::instance().run(G, run_group);
    for (int gi = 0; gi < G; gi++) if (gst[(size_t)gi] != VELO_OK) { g_err = gerr[(size_t)gi]; return gst[(size_t)gi]; }
    return VELO_OK;
}

// main.cpp:216,349 load a scan per frame: the caller that knows which cloud it will hand over as the NEXT source says so, and the library
// uploads it on a copy stream of its own while the current registration's chain of launches runs (issued by the thread that is about to
// wait for that chain).  Host clouds only; the very next velo_set_source / batch job that names the same pointer and size takes the uploaded
// copy, anything else drops it.  No effect on results.
int velo_hint_next_source(velo_ctx* c, const velo_scan_ref* next) {
    if (!c) return fail(VELO_ERR_INVALID, "null context");
    c->pf.hinted = false;
    if (!next || (next->on_device & VELO_SCAN_ON_DEVICE) || !next->xyz || !next->ring_offsets || next->n_rings <= 0 || next->stride_bytes < 12) return VELO_OK;
    const int n = next->ring_offsets[next->n_rings];
    if (n <= 0) return VELO_OK;
    c->pf.host = next->xyz; c->pf.bytes = (size_t)(n - 1) * (size_t)next->stride_bytes + 12; c->pf.hinted = true; c->pf.ready = false;
    return VELO_OK;
}

// The step of a drive announced one call ahead (main.cpp:216,233,349,380: every frame promotes the previous scan and loads a new one): the next
// call WILL promote this context's source to target and bring `next` as the new source.  A chained registration then enqueues exactly those
// loads behind its own launches before its thread waits, so they run while the host reads the results and hands the pose over; the next
// call finds the frame in place.  Until that call the context is one frame ahead: any other job on it is VELO_ERR_STATE.  A call that had to be
// repeated host-driven gets its own pair back first.  Results never change.  Host clouds are uploaded ahead as velo_hint_next_source does.
int velo_hint_next_frame(velo_ctx* c, const velo_scan_ref* next) {
    if (!c) return fail(VELO_ERR_INVALID, "null context");
    c->nf.hint_valid = false;
    if (!next || !next->xyz || !next->ring_offsets || next->n_rings <= 0 || next->stride_bytes < 12 || (next->on_device & (VELO_SCAN_PROMOTE | VELO_SCAN_SHARED))) return VELO_OK;
    c->nf.hint = *next;
    c->nf.hint_valid = true;
    return velo_hint_next_source(c, next);
}

int velo_register_sequences(velo_ctx** ctxs, int32_t n, int32_t n_frames, const velo_scan_ref* frames, const velo_match* const* matches, const int32_t* n_matches,
                            double* poses, double* x_guess, double* x_out, double* T_out, velo_summary* summaries, int32_t flags) {
    return sequences_impl(ctxs, n, n_frames, frames, matches, n_matches, poses, x_guess, x_out, T_out, summaries, flags);
}

int velo_register_batch(velo_ctx** ctxs, int32_t n, const velo_scan_ref* targets, const velo_scan_ref* sources, double* x, double* T, velo_summary* summaries) {
    return batch_impl(ctxs, n, targets, sources, x, T, summaries);
}

int velo_register_batch_visual(velo_ctx** ctxs, int32_t n, const velo_scan_ref* targets, const velo_scan_ref* sources, const velo_match* const* matches,
                               const int32_t* n_matches, double* x, double* T, velo_summary* summaries) {
    if (n > 0 && !n_matches) return fail(VELO_ERR_INVALID, "n_matches is null");
    JobVisual V; V.m = matches; V.n = n_matches;
    return batch_impl(ctxs, n, targets, sources, x, T, summaries, V);
}

// util::pose_mat2vec (utility.h:67-82): 6-vector -> 4x4, row-major out.  Column j of R is R(omega) e_j, which is what
// ceres::AngleAxisToRotationMatrix [3P] writes column-major and utility.h:73-77 transposes back.
int velo_pose_vec_to_mat(const double x[6], double T[16]) {
    if (!x || !T) return fail(VELO_ERR_INVALID, "null argument");
    for (int i = 0; i < 16; i++) T[i] = 0.0;
    T[15] = 1.0;
    const double theta2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    if (theta2 > std::numeric_limits<double>::epsilon()) {
        const double theta = std::sqrt(theta2), wx = x[0] / theta, wy = x[1] / theta, wz = x[2] / theta;
        double c, s;
        velo_sincos(theta, &s, &c);
        T[0] = c + wx * wx * (1 - c);       T[4] = wz * s + wx * wy * (1 - c);  T[8] = -wy * s + wx * wz * (1 - c);
        T[1] = wx * wy * (1 - c) - wz * s;  T[5] = c + wy * wy * (1 - c);       T[9] = wx * s + wy * wz * (1 - c);
        T[2] = wy * s + wx * wz * (1 - c);  T[6] = -wx * s + wy * wz * (1 - c); T[10] = c + wz * wz * (1 - c);
    } else {
        T[0] = 1;      T[4] = x[2];   T[8] = -x[1];
        T[1] = -x[2];  T[5] = 1;      T[9] = x[0];
        T[2] = x[1];   T[6] = -x[0];  T[10] = 1;
    }
    T[3] = x[3]; T[7] = x[4]; T[11] = x[5];
    return VELO_OK;
}

// util::pose_vec2mat (utility.h:83-96): 4x4 -> 6-vector via the quaternion route of ceres::RotationMatrixToAngleAxis [3P]
int velo_pose_mat_to_vec(const double T[16], double x[6]) {
    if (!x || !T) return fail(VELO_ERR_INVALID, "null argument");
    const double R[3][3] = {{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}};
    double q[4] = {0, 0, 0, 0};
    const double tr = R[0][0] + R[1][1] + R[2][2];
    if (tr >= 0.0) {
        double t = std::sqrt(tr + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = (R[2][1] - R[1][2]) * t; q[2] = (R[0][2] - R[2][0]) * t; q[3] = (R[1][0] - R[0][1]) * t;
    } else {
        int i = 0;
        if (R[1][1] > R[0][0]) i = 1;
        if (R[2][2] > R[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
        q[i + 1] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[k][j] - R[j][k]) * t; q[j + 1] = (R[j][i] + R[i][j]) * t; q[k + 1] = (R[k][i] + R[i][k]) * t;
    }
    const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2 > 0.0) {
        const double s = std::sqrt(s2);
        const double two_theta = 2.0 * ((q[0] < 0.0) ? std::atan2(-s, -q[0]) : std::atan2(s, q[0]));
        const double k = two_theta / s;
        x[0] = q[1] * k; x[1] = q[2] * k; x[2] = q[3] * k;
    } else {
        x[0] = q[1] * 2.0; x[1] = q[2] * 2.0; x[2] = q[3] * 2.0;
    }
    x[3] = T[3]; x[4] = T[7]; x[5] = T[11];
    return VELO_OK;
}

// The pose hand-off of the drive loop for n sequences at once (main.cpp:311-331,408): pose[k] = pose[k-1] * dpose (main.cpp:408), then the
// next frame's constant-velocity guess pose_vec2mat(pose[k-1]^-1 * pose[k]) (main.cpp:315-317,331).  Row-major 4x4s; plain double
// arithmetic in the order Eigen's fixed-size products take (sum over k = 0..3); the inverse is the general 4x4 inverse Eigen's
// Matrix4d::inverse() computes by cofactors -- for a rigid pose it equals [R^T | -R^T t] to rounding, and the guess only seeds the solve.
static void mat4_mul(const double* A, const double* B, double* Cm) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double v = 0.0;
        for (int k = 0; k < 4; k++) v += A[4 * i + k] * B[4 * k + j];
        Cm[4 * i + j] = v;
    }
}
static bool mat4_inv(const double* m, double* inv) {
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    if (det == 0.0 || !std::isfinite(det)) return false;
    const double r = 1.0 / det;
    for (int i = 0; i < 16; i++) inv[i] = a[i] * r;
    return true;
}
int velo_pose_handoff(int32_t n, double* poses, const double* dpose, double* x_next) {
    if (n < 0 || (n > 0 && (!poses || !dpose))) return fail(VELO_ERR_INVALID, "null/negative argument");
    for (int i = 0; i < n; i++) {
        double* P = poses + 16 * (size_t)i;
        double Pn[16], Pi[16], dT[16];
        mat4_mul(P, dpose + 16 * (size_t)i, Pn);                     // main.cpp:408
        if (!mat4_inv(P, Pi)) return fail(VELO_ERR_INVALID, "pose %d is singular", i);
        mat4_mul(Pi, Pn, dT);                                        // main.cpp:315-317 (one frame later)
        if (x_next) VELO_TRY(velo_pose_mat_to_vec(dT, x_next + 6 * (size_t)i));   // main.cpp:331
        std::memcpy(P, Pn, sizeof(Pn));
    }
    return VELO_OK;
}

int velo_comm_unique_id(char id[128]) {
    if (!id) return fail(VELO_ERR_INVALID, "null id");
    static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than the ABI's 128 bytes");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    std::memset(id, 0, 128);
    std::memcpy(id, &u, sizeof(u));
    return VELO_OK;
}

int velo_comm_init(velo_ctx* c, const char id[128], int32_t rank, int32_t world) {
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(VELO_ERR_INVALID, "bad comm arguments");
    HIP_TRY(hipSetDevice(c->device));
    if (c->comm) { NCCL_TRY(ncclCommDestroy(c->comm)); c->comm = nullptr; }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    NCCL_TRY(ncclCommInitRank(&c->comm, world, u, rank));
    c->shard_rank = rank; c->shard_world = world;
    c->have_corr = false;
    return VELO_OK;
}

static void peer_release(velo_ctx* c) {
    for (int r = 0; r < kMaxPeers; r++) {
        if (c->peer_mapped[r]) { (void)hipIpcCloseMemHandle(c->peer_mapped[r]); c->peer_mapped[r] = nullptr; }
    }
    for (int r = 0; r < kMaxPeers; r++) {
        if (c->peer_area_mapped[r]) { (void)hipIpcCloseMemHandle(c->peer_area_mapped[r]); c->peer_area_mapped[r] = nullptr; }
    }
    c->peer_on = false; c->peer_recs_on = false;
    std::memset(&c->peer, 0, sizeof(c->peer));
    std::memset(&c->peer_recs, 0, sizeof(c->peer_recs));
}

int velo_comm_peer_export(velo_ctx* c, char handle[64]) {
    if (!c || !handle) return fail(VELO_ERR_INVALID, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes in the ABI");
    HIP_TRY(hipSetDevice(c->device));
    // Every export hands out a NEW slab.  On the recovery path (a VELO_ERR_COMM timeout, then export + attach again on every rank) a
    // slow peer's timed-out call may still be storing old-epoch blocks while a fast rank is already here; with sequence numbers
    // restarting at attach, such a block written into a re-used slab could be taken for a new one.  The old slab is therefore
    // retired, not cleared and re-used: stale stores land in memory nobody reads any more (4.5 KB per recovery, freed with the context).
    if (c->peer_slab) { c->peer_retired.push_back(c->peer_slab); c->peer_slab = nullptr; }
    // ... but not for ever: a context that exports per leg or per recovery would grow by an allocation granule each time.  Only a call that
    // timed out (5 s bound) before the LAST TWO exports could still be storing into an older slab; those are freed here.
    while (c->peer_retired.size() > 2) { (void)hipFree(c->peer_retired.front()); c->peer_retired.erase(c->peer_retired.begin()); }
    {
        // fine-grained device memory: stores of a peer on another GPU become visible while the kernels run
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, sizeof(PeerSlab), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipMalloc(&p, sizeof(PeerSlab)));
        }
        c->peer_slab = (PeerSlab*)p;
    }
    // The slab is cleared HERE, before its handle leaves this call, and never again: a peer may store into it as soon as it has attached,
    // and nothing orders that against this rank's own attach.  (Every rank exports before any rank can attach -- the host program's
    // exchange of the handles is that barrier -- so no store of the new epoch can precede this clear.)
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemset(c->peer_slab, 0, sizeof(PeerSlab)));
    HIP_TRY(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, c->peer_slab));
    std::memcpy(handle, &h, 64);
    return VELO_OK;
}

int velo_comm_peer_attach(velo_ctx* c, const char* handles, int32_t rank, int32_t world) {
    if (!c || !handles || world < 1 || world > kMaxPeers || rank < 0 || rank >= world) return fail(VELO_ERR_INVALID, "bad peer arguments (world <= %d)", kMaxPeers);
    if (!c->peer_slab) return fail(VELO_ERR_STATE, "velo_comm_peer_export must be called first");
    if (c->comm) return fail(VELO_ERR_STATE, "an RCCL communicator is attached; destroy it first");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    peer_release(c);
    VELO_TRY(c->peer_seq.reserve(1)); VELO_TRY(c->peer_err.reserve(1)); VELO_TRY(c->peer_kseq.reserve(1));
    if (!c->h_agree) HIP_TRY(hipHostMalloc((void**)&c->h_agree, sizeof(int) * 64, hipHostMallocDefault));
    HIP_TRY(hipMemset(c->peer_seq.p, 0, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->peer_kseq.p, 0, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->peer_err.p, 0, sizeof(int)));
    // (the slab itself was cleared by velo_comm_peer_export: a peer that attached earlier may already be storing into it)
    // Chain mode over peers enqueues a predicted number of LM launches per solve, and every rank must enqueue the SAME number: the
    // ranks agree on the counts at the start of every chained call (peer_agree_kernel, the maximum over ranks), whatever their
    // histories are.  The history still restarts here so that the first calls of a fresh communicator predict alike.
    for (int k = 0; k < VELO_MAX_SOLVES; k++) { c->pred_evals[k] = (k == 0) ? 12 : 5; c->eval_hist_n[k] = 0; }
    for (int r = 0; r < world; r++) {
        if (r == rank) { c->peer.slab[r] = c->peer_slab; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * 64, 64);
        void* p = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer_mapped[r] = p;
        c->peer.slab[r] = (PeerSlab*)p;
    }
    c->peer.seq = c->peer_seq.p; c->peer.kseq = c->peer_kseq.p; c->peer.error = c->peer_err.p; c->peer.rank = rank; c->peer.world = world;
    c->peer_on = true;
    c->shard_rank = rank; c->shard_world = world;
    c->have_corr = false;
    return VELO_OK;
}

int velo_comm_peer_export_records(velo_ctx* c, int32_t max_queries, char handle[64]) {
    if (!c || !handle || max_queries < 1) return fail(VELO_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->peer_area && c->peer_area_queries != max_queries) return fail(VELO_ERR_STATE, "the record area exists already, sized for %d queries", c->peer_area_queries);
    if (!c->peer_area) {
        const size_t recs = 2 * ((size_t)max_queries + 8 * kMaxPeers);
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, recs * sizeof(PartialRec), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipMalloc(&p, recs * sizeof(PartialRec)));
        }
        c->peer_area = (PartialRec*)p;
        c->peer_area_queries = max_queries;
    }
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, c->peer_area));
    std::memcpy(handle, &h, 64);
    return VELO_OK;
}

int velo_comm_peer_attach_records(velo_ctx* c, const char* handles, int32_t max_queries) {
    if (!c || !handles) return fail(VELO_ERR_INVALID, "null argument");
    if (!c->peer_on) return fail(VELO_ERR_STATE, "velo_comm_peer_attach comes first");
    if (!c->peer_area || c->peer_area_queries != max_queries) return fail(VELO_ERR_STATE, "velo_comm_peer_export_records(%d) comes first", max_queries);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int W = c->peer.world, rank = c->peer.rank;
    for (int r = 0; r < W; r++) {
        if (r == rank) { c->peer_recs.area[r] = c->peer_area; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * 64, 64);
        void* p = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer_area_mapped[r] = p;
        c->peer_recs.area[r] = (PartialRec*)p;
    }
    c->peer_recs.rank = rank; c->peer_recs.world = W; c->peer_recs.max_share = 0;
    c->peer_recs.parity_stride = (size_t)max_queries + 8 * kMaxPeers;
    c->peer_recs_on = true;
    c->peer_xseq = 0;
    return VELO_OK;
}

int velo_comm_info(const velo_ctx* c, int32_t* kind, int32_t* rank, int32_t* world) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    int k = 0, w = c->shard_world;
    if (c->peer_on) k = 2;
    else if (c->comm) {
        k = 1;
        int n = 0;
        NCCL_TRY(ncclCommCount(c->comm, &n));                   // read back from the communicator, not from what the caller said
        w = n;
    }
    if (kind) *kind = k;
    if (rank) *rank = c->shard_rank;
    if (world) *world = w;
    return VELO_OK;
}

int velo_comm_destroy(velo_ctx* c) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (c->peer_on) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        peer_release(c);
    }
    if (c->comm) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        NCCL_TRY(ncclCommDestroy(c->comm));
        c->comm = nullptr;
    }
    c->shard_rank = 0; c->shard_world = 1;
    c->have_corr = false;
    return VELO_OK;
}

int velo_comm_set_target_sharded(velo_ctx* c, int enable) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    c->target_sharded = enable != 0;
    c->have_corr = false;
    return VELO_OK;
}

int velo_set_query_shard(velo_ctx* c, int32_t rank, int32_t world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return fail(VELO_ERR_INVALID, "bad shard arguments");
    if (c->comm || c->peer_on) return fail(VELO_ERR_STATE, "a communicator is attached; its rank/world define the shard");
    c->shard_rank = rank; c->shard_world = world;
    c->have_corr = false;
    return VELO_OK;
}

int velo_synchronize(velo_ctx* c) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->timing >= 2 && c->klog_used > 0) {                            // brackets of launches outside a registration (index builds)
        const int used = c->assoc_events_used;
        c->assoc_events_used = 0;
        const int st = read_assoc_timing(c, nullptr);
        c->assoc_events_used = used;
        VELO_TRY(st);
    }
    return VELO_OK;
}

// ---- projectLidarToCamera + featureDepthAssociation (velo.h:329-497) ---------------------------------
int velo_project_lidar(velo_ctx* c, int32_t of_target, const float cam_t[3], const double bounds[4], int32_t* n_valid_total) {
    if (!c || !cam_t || !bounds) return fail(VELO_ERR_INVALID, "null argument");
    if (of_target ? !c->have_target : !c->have_source) return fail(VELO_ERR_STATE, "no %s cloud loaded", of_target ? "target" : "source");
    HIP_TRY(hipSetDevice(c->device));
    const int n = of_target ? c->T->n_tgt : c->n_src;
    const std::vector<int>& h_off = of_target ? c->T->h_tgt_off : c->h_src_off;
    const int nr = (int)h_off.size() - 1;
    c->have_projection = false;
    c->h_proj_off = h_off;
    c->proj_rings = nr; c->proj_points = n; c->proj_of_target = of_target ? 1 : 0;
    c->h_ring_cnt.assign((size_t)std::max(nr, 0), 0);
    VELO_TRY(c->pstack.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->vstack.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->ring_cnt.reserve((size_t)std::max(nr, 1)));
    VELO_TRY(c->proj_off.reserve((size_t)nr + 1));
    if (nr > 0) {
        HIP_TRY(hipMemcpyAsync(c->proj_off.p, h_off.data(), sizeof(int) * ((size_t)nr + 1), hipMemcpyHostToDevice, c->stream));
        CamWindow W;
        W.tx = cam_t[0]; W.ty = cam_t[1]; W.tz = cam_t[2];
        W.min_x = bounds[0]; W.max_x = bounds[1]; W.min_y = bounds[2]; W.max_y = bounds[3];
        hipLaunchKernelGGL(project_ring_kernel, dim3(nr), dim3(256), 0, c->stream, (const float4*)(of_target ? c->T->tgt.p : c->src.p),
                           (const int*)c->proj_off.p, nr, W, c->pstack.p, c->vstack.p, c->ring_cnt.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->h_ring_cnt.data(), c->ring_cnt.p, sizeof(int) * (size_t)nr, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->have_projection = true;
    if (n_valid_total) {
        int64_t tot = 0;
        for (int v : c->h_ring_cnt) tot += v;
        *n_valid_total = (int32_t)tot;
    }
    return VELO_OK;
}

int velo_get_projection(velo_ctx* c, float* proj_xy, float* points_xyz, int32_t capacity_points, int32_t* ring_offsets,
                        int32_t capacity_offsets, int32_t* n_rings) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (!c->have_projection) return fail(VELO_ERR_STATE, "velo_project_lidar has not run");
    const int nr = c->proj_rings;
    if (n_rings) *n_rings = nr;
    std::vector<int> out_off((size_t)nr + 1, 0);
    for (int s = 0; s < nr; s++) out_off[s + 1] = out_off[s] + c->h_ring_cnt[s];
    if (ring_offsets) for (int s = 0; s <= nr && s < capacity_offsets; s++) ring_offsets[s] = out_off[s];
    if ((!proj_xy && !points_xyz) || capacity_points <= 0 || c->proj_points == 0) return VELO_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float4> hp((size_t)c->proj_points), hv((size_t)c->proj_points);
    HIP_TRY(hipMemcpy(hp.data(), c->pstack.p, sizeof(float4) * hp.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hv.data(), c->vstack.p, sizeof(float4) * hv.size(), hipMemcpyDeviceToHost));
    for (int s = 0; s < nr; s++)
        for (int j = 0; j < c->h_ring_cnt[s]; j++) {
            const int o = out_off[s] + j;
            if (o >= capacity_points) return VELO_OK;
            const float4 a = hp[(size_t)c->h_proj_off[s] + j], v = hv[(size_t)c->h_proj_off[s] + j];
            if (proj_xy) { proj_xy[2 * o] = a.x; proj_xy[2 * o + 1] = a.y; }
            if (points_xyz) { points_xyz[3 * o] = v.x; points_xyz[3 * o + 1] = v.y; points_xyz[3 * o + 2] = v.z; }
        }
    return VELO_OK;
}

int velo_depth_association(velo_ctx* c, const float* keypoints_xy, int32_t n, double thresh, float* kp_with_depth_xyz,
                           int32_t capacity_points, int32_t* has_depth, int32_t* n_with_depth) {
    if (!c || n < 0 || (n > 0 && (!keypoints_xy || !has_depth))) return fail(VELO_ERR_INVALID, "bad keypoint arguments");
    if (!c->have_projection) return fail(VELO_ERR_STATE, "velo_project_lidar has not run");
    if (n_with_depth) *n_with_depth = 0;
    if (n == 0) return VELO_OK;
    HIP_TRY(hipSetDevice(c->device));
    VELO_TRY(c->kps.reserve((size_t)n)); VELO_TRY(c->kp_point.reserve((size_t)n)); VELO_TRY(c->kp_out.reserve((size_t)n));
    VELO_TRY(c->kp_flag.reserve((size_t)n + 2)); VELO_TRY(c->kp_excl.reserve((size_t)n + 2)); VELO_TRY(c->kp_has.reserve((size_t)n));
    VELO_TRY(c->cursor.reserve((size_t)n + 2));
    const int n_tiles = cdiv(n, kScanTile);
    VELO_TRY(c->scan_tiles.reserve((size_t)n_tiles + 1));
    VELO_TRY(c->scan_total.reserve(1));
    HIP_TRY(hipMemcpyAsync(c->kps.p, keypoints_xy, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(depth_assoc_kernel, dim3(cdiv(n, 4)), dim3(256), 0, c->stream, (const float2*)c->kps.p, n, (const float4*)c->pstack.p,
                       (const float4*)c->vstack.p, (const int*)c->proj_off.p, (const int*)c->ring_cnt.p, c->proj_rings, thresh, c->kp_point.p, c->kp_flag.p);
    HIP_TRY(hipMemcpyAsync(c->kp_excl.p, c->kp_flag.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(n_tiles), dim3(kScanThreads), 0, c->stream, c->kp_excl.p, n, c->scan_tiles.p);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanThreads), 0, c->stream, c->scan_tiles.p, n_tiles, c->scan_total.p);
    hipLaunchKernelGGL(scan_add_kernel, dim3(cdiv(n + 1, 256)), dim3(256), 0, c->stream, c->kp_excl.p, n, c->scan_tiles.p, c->scan_total.p, c->cursor.p);
    hipLaunchKernelGGL(depth_compact_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const int*)c->kp_flag.p, (const int*)c->kp_excl.p,
                       (const float4*)c->kp_point.p, n, c->kp_has.p, c->kp_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(has_depth, c->kp_has.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_int, c->scan_total.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int nd = c->h_int[0];
    if (n_with_depth) *n_with_depth = nd;
    if (kp_with_depth_xyz && capacity_points > 0 && nd > 0) {
        std::vector<float4> h((size_t)nd);
        HIP_TRY(hipMemcpy(h.data(), c->kp_out.p, sizeof(float4) * (size_t)nd, hipMemcpyDeviceToHost));
        for (int i = 0; i < nd && i < capacity_points; i++) { kp_with_depth_xyz[3 * i] = h[i].x; kp_with_depth_xyz[3 * i + 1] = h[i].y; kp_with_depth_xyz[3 * i + 2] = h[i].z; }
    }
    return VELO_OK;
}

// ---- batched triangulatePoint (velo.h:1027-1130) ----------------------------------------------------
int velo_triangulate_points(velo_ctx* c, const double* camera_poses, int32_t n_frames, const float* cam_trans, int32_t n_cams,
                            const velo_tri_obs* obs, const int32_t* obs_offsets, int32_t n, float* points_xyz,
                            const uint8_t* initial_guess, velo_tri_result* results) {
    if (!c || n < 0 || n_frames < 0 || n_cams < 0) return fail(VELO_ERR_INVALID, "null ctx / negative size");
    if (n == 0) return VELO_OK;
    if (!obs_offsets || !points_xyz) return fail(VELO_ERR_INVALID, "null offsets / points");
    if (obs_offsets[0] != 0) return fail(VELO_ERR_INVALID, "obs_offsets[0] must be 0");
    for (int l = 0; l < n; l++) if (obs_offsets[l + 1] < obs_offsets[l]) return fail(VELO_ERR_INVALID, "obs_offsets must not decrease (landmark %d)", l);
    const int n_obs = obs_offsets[n];
    if (n_obs > 0 && (!obs || !camera_poses)) return fail(VELO_ERR_INVALID, "null observations / poses");
    for (int k = 0; k < n_obs; k++) {
        const velo_tri_obs& o = obs[k];
        if (o.kind != VELO_TRI_OBS_3D && o.kind != VELO_TRI_OBS_2D) return fail(VELO_ERR_INVALID, "observation %d: unknown kind %d", k, o.kind);
        if (o.frame < 0 || o.frame >= n_frames) return fail(VELO_ERR_INVALID, "observation %d: frame %d outside [0, %d)", k, o.frame, n_frames);
        if (o.kind == VELO_TRI_OBS_2D && (o.cam < 0 || o.cam >= n_cams || !cam_trans)) return fail(VELO_ERR_INVALID, "observation %d: camera %d outside [0, %d)", k, o.cam, n_cams);
    }
    HIP_TRY(hipSetDevice(c->device));
    // per-frame constants in double with the host libm: rot = -pose[0..2] (costfunctions.h:318-320), Rodrigues scalars, R columns
    std::vector<TriFrame> hf((size_t)std::max(n_frames, 1));
    for (int f = 0; f < n_frames; f++) {
        const double* cp = camera_poses + 6 * (size_t)f;
        const double xr[6] = {-cp[0], -cp[1], -cp[2], 0.0, 0.0, 0.0};
        PoseScalars S;
        pose_scalars(xr, &S);
        TriFrame& F = hf[(size_t)f];
        std::memset(&F, 0, sizeof(F));
        for (int k = 0; k < 3; k++) { F.w[k] = S.w[k]; F.u[k] = S.u[k]; F.center[k] = cp[3 + k]; }
        F.c = S.c; F.s = S.s; F.omc = S.omc; F.small = S.small;
        for (int j = 0; j < 3; j++) {                                // column j = rotation of e_j, same operation order as the device form
            double e[3] = {0.0, 0.0, 0.0}, o[3];
            e[j] = 1.0;
            if (!F.small) {
                const double c0 = F.u[1] * e[2] - F.u[2] * e[1], c1 = F.u[2] * e[0] - F.u[0] * e[2], c2 = F.u[0] * e[1] - F.u[1] * e[0];
                const double tmp = (F.u[0] * e[0] + F.u[1] * e[1] + F.u[2] * e[2]) * F.omc;
                o[0] = e[0] * F.c + c0 * F.s + F.u[0] * tmp;
                o[1] = e[1] * F.c + c1 * F.s + F.u[1] * tmp;
                o[2] = e[2] * F.c + c2 * F.s + F.u[2] * tmp;
            } else {
                o[0] = e[0] + (F.w[1] * e[2] - F.w[2] * e[1]);
                o[1] = e[1] + (F.w[2] * e[0] - F.w[0] * e[2]);
                o[2] = e[2] + (F.w[0] * e[1] - F.w[1] * e[0]);
            }
            F.R[0 * 3 + j] = o[0]; F.R[1 * 3 + j] = o[1]; F.R[2 * 3 + j] = o[2];
        }
    }
    std::vector<double> hct((size_t)std::max(3 * n_cams, 3), 0.0);
    for (int k = 0; k < 3 * n_cams; k++) hct[(size_t)k] = (double)cam_trans[k];
    VELO_TRY(c->tri_frames.reserve(hf.size())); VELO_TRY(c->tri_cam_t.reserve(hct.size()));
    VELO_TRY(c->tri_obs.reserve((size_t)std::max(n_obs, 1))); VELO_TRY(c->tri_off.reserve((size_t)n + 1));
    VELO_TRY(c->tri_pts.reserve((size_t)3 * n)); VELO_TRY(c->tri_init.reserve((size_t)n)); VELO_TRY(c->tri_res.reserve((size_t)n));
    HIP_TRY(hipMemcpyAsync(c->tri_frames.p, hf.data(), sizeof(TriFrame) * hf.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tri_cam_t.p, hct.data(), sizeof(double) * hct.size(), hipMemcpyHostToDevice, c->stream));
    // blocks enter the problem 3-D first (velo.h:1049-1122): stable partition per landmark, so that block position == index
    std::vector<velo_tri_obs> hobs((size_t)std::max(n_obs, 1));
    for (int l = 0; l < n; l++) {
        int w = obs_offsets[l];
        for (int pass = 0; pass < 2; pass++)
            for (int k = obs_offsets[l]; k < obs_offsets[l + 1]; k++)
                if ((obs[k].kind == VELO_TRI_OBS_2D) == (pass == 1)) hobs[(size_t)w++] = obs[k];
    }
    if (n_obs > 0) HIP_TRY(hipMemcpyAsync(c->tri_obs.p, hobs.data(), sizeof(velo_tri_obs) * (size_t)n_obs, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tri_off.p, obs_offsets, sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tri_pts.p, points_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (initial_guess) HIP_TRY(hipMemcpyAsync(c->tri_init.p, initial_guess, (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                         // hf / hct are stack-owned host vectors
    TriParams P;
    P.lm = lm_params(c->P);
    P.loss_a = c->P.loss_thresh_3D2D; P.loss_w = c->P.weight_3D2D;    // velo.h:1116-1119
    if (c->tri_variant == 0)     // VELO_TRI_VARIANT=0: one thread per landmark (kept for A/B; same results)
        hipLaunchKernelGGL(triangulate_kernel, dim3(cdiv(n, 64)), dim3(64), 0, c->stream, (const TriFrame*)c->tri_frames.p, (const double*)c->tri_cam_t.p,
                           (const velo_tri_obs*)c->tri_obs.p, (const int*)c->tri_off.p, n, P, c->tri_pts.p,
                           (const unsigned char*)(initial_guess ? c->tri_init.p : nullptr), c->tri_res.p);
    else
        hipLaunchKernelGGL(triangulate_wave_kernel, dim3(n), dim3(64), 0, c->stream, (const TriFrame*)c->tri_frames.p, (const double*)c->tri_cam_t.p,
                           (const velo_tri_obs*)c->tri_obs.p, (const int*)c->tri_off.p, n, P, c->tri_pts.p,
                           (const unsigned char*)(initial_guess ? c->tri_init.p : nullptr), c->tri_res.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(points_xyz, c->tri_pts.p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (results) HIP_TRY(hipMemcpyAsync(results, c->tri_res.p, sizeof(velo_tri_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VELO_OK;
}

}  // extern "C"
