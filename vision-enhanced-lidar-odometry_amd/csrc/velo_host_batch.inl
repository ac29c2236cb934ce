// velo_host_batch.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  The lock-step batch driver: group association launches, the next frame loaded behind the chain, f2f_batch_lockstep, jobs, sequences, velo_register_*.
extern "C" {   // (continued from the previous part)

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
        static const bool enq_trace = dev_env("VELO_ENQ_TRACE") != nullptr;         // dev aid: how long the host needs to ENQUEUE a group's chain, and how long it then waits
        const auto t_enq_done = std::chrono::steady_clock::now();
        if (preloaded) HIP_TRY(hipEventSynchronize(c0->nf.call_done));   // the results are in; the next frame's loads are still running
        else HIP_TRY(hipStreamSynchronize(bs));
        if (enq_trace) fprintf(stderr, "[velo enq] %d contexts: entry -> chain enqueued %.0f us (of it before the first launch %.0f us); then waited %.0f us for the results\n", n,
                               std::chrono::duration<double, std::micro>(t_enq_done - t_entry).count(), std::chrono::duration<double, std::micro>(t_chain1 - t_entry).count(),
                               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq_done).count());
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
}  // extern "C"   (continued in the next part)
