// velo_api_pose_comm.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  C-ABI: pose helpers and hand-off, the communicators (RCCL, peer slabs), shards, synchronize.
extern "C" {   // (continued from the previous part)
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
}  // extern "C"   (continued in the next part)
