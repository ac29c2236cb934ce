// velo_host_chain.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  Chain mode (a whole frame_to_frame call as one chain of launches), kernel times, velo_frame_to_frame.
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
}  // extern "C"   (continued in the next part)
