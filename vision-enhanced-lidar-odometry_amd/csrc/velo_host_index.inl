// velo_host_index.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  Launch timing, uploads, the index build (build_grid), the query list in patch order.
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
}  // namespace   (continued in the next part)
