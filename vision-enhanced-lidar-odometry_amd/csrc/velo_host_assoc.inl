// velo_host_assoc.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  Pose scalars, parameter blocks, evaluation plan, seeds and askers, the association driver of one context (do_associate).
namespace {   // (continued from the previous part)
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
}  // namespace   (continued in the next part)
