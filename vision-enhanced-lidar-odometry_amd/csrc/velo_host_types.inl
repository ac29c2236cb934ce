// velo_host_types.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  Error state, device buffers, the search index, TargetData and velo_ctx: what one context holds.
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
