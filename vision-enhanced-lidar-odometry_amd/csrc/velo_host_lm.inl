// velo_host_lm.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  Peer checks, association timing read-out, the visual gate, the Levenberg-Marquardt driver of one context (do_solve).
namespace {   // (continued from the previous part)
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
}  // namespace   (continued in the next part)
