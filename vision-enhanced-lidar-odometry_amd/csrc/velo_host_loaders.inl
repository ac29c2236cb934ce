// velo_host_loaders.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  Direction image, target / source ingest and finalize: how a scan enters a context.
namespace {   // (continued from the previous part)
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
