// velo_api_solve.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  C-ABI: associate (incl. target-sharded partials and merge), correspondences, evaluate, functors, residual statistics, solve.
extern "C" {   // (continued from the previous part)
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
