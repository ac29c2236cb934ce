// velo_api_context.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  C-ABI: create / destroy / params, set_target / set_source / scan cache / promotion, clouds and visual matches.
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
}  // extern "C"   (continued in the next part)
