// velo_api_next_rows.inl -- part of the host side of the C-ABI, included by velo_hip.hip (ONE translation unit; the order of the parts is the order of
// definition).  C-ABI: SURVEY 8(f) rows 3 and 4 -- projection, keypoint depth, batched triangulation.
extern "C" {   // (continued from the previous part)
// ---- projectLidarToCamera + featureDepthAssociation (velo.h:329-497) ---------------------------------
int velo_project_lidar(velo_ctx* c, int32_t of_target, const float cam_t[3], const double bounds[4], int32_t* n_valid_total) {
    if (!c || !cam_t || !bounds) return fail(VELO_ERR_INVALID, "null argument");
    if (of_target ? !c->have_target : !c->have_source) return fail(VELO_ERR_STATE, "no %s cloud loaded", of_target ? "target" : "source");
    HIP_TRY(hipSetDevice(c->device));
    const int n = of_target ? c->T->n_tgt : c->n_src;
    const std::vector<int>& h_off = of_target ? c->T->h_tgt_off : c->h_src_off;
    const int nr = (int)h_off.size() - 1;
    c->have_projection = false;
    c->h_proj_off = h_off;
    c->proj_rings = nr; c->proj_points = n; c->proj_of_target = of_target ? 1 : 0;
    c->h_ring_cnt.assign((size_t)std::max(nr, 0), 0);
    VELO_TRY(c->pstack.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->vstack.reserve((size_t)std::max(n, 1)));
    VELO_TRY(c->ring_cnt.reserve((size_t)std::max(nr, 1)));
    VELO_TRY(c->proj_off.reserve((size_t)nr + 1));
    if (nr > 0) {
        HIP_TRY(hipMemcpyAsync(c->proj_off.p, h_off.data(), sizeof(int) * ((size_t)nr + 1), hipMemcpyHostToDevice, c->stream));
        CamWindow W;
        W.tx = cam_t[0]; W.ty = cam_t[1]; W.tz = cam_t[2];
        W.min_x = bounds[0]; W.max_x = bounds[1]; W.min_y = bounds[2]; W.max_y = bounds[3];
        hipLaunchKernelGGL(project_ring_kernel, dim3(nr), dim3(256), 0, c->stream, (const float4*)(of_target ? c->T->tgt.p : c->src.p),
                           (const int*)c->proj_off.p, nr, W, c->pstack.p, c->vstack.p, c->ring_cnt.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->h_ring_cnt.data(), c->ring_cnt.p, sizeof(int) * (size_t)nr, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->have_projection = true;
    if (n_valid_total) {
        int64_t tot = 0;
        for (int v : c->h_ring_cnt) tot += v;
        *n_valid_total = (int32_t)tot;
    }
    return VELO_OK;
}

int velo_get_projection(velo_ctx* c, float* proj_xy, float* points_xyz, int32_t capacity_points, int32_t* ring_offsets,
                        int32_t capacity_offsets, int32_t* n_rings) {
    if (!c) return fail(VELO_ERR_INVALID, "null ctx");
    if (!c->have_projection) return fail(VELO_ERR_STATE, "velo_project_lidar has not run");
    const int nr = c->proj_rings;
    if (n_rings) *n_rings = nr;
    std::vector<int> out_off((size_t)nr + 1, 0);
    for (int s = 0; s < nr; s++) out_off[s + 1] = out_off[s] + c->h_ring_cnt[s];
    if (ring_offsets) for (int s = 0; s <= nr && s < capacity_offsets; s++) ring_offsets[s] = out_off[s];
    if ((!proj_xy && !points_xyz) || capacity_points <= 0 || c->proj_points == 0) return VELO_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float4> hp((size_t)c->proj_points), hv((size_t)c->proj_points);
    HIP_TRY(hipMemcpy(hp.data(), c->pstack.p, sizeof(float4) * hp.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hv.data(), c->vstack.p, sizeof(float4) * hv.size(), hipMemcpyDeviceToHost));
    for (int s = 0; s < nr; s++)
        for (int j = 0; j < c->h_ring_cnt[s]; j++) {
            const int o = out_off[s] + j;
            if (o >= capacity_points) return VELO_OK;
            const float4 a = hp[(size_t)c->h_proj_off[s] + j], v = hv[(size_t)c->h_proj_off[s] + j];
            if (proj_xy) { proj_xy[2 * o] = a.x; proj_xy[2 * o + 1] = a.y; }
            if (points_xyz) { points_xyz[3 * o] = v.x; points_xyz[3 * o + 1] = v.y; points_xyz[3 * o + 2] = v.z; }
        }
    return VELO_OK;
}

int velo_depth_association(velo_ctx* c, const float* keypoints_xy, int32_t n, double thresh, float* kp_with_depth_xyz,
                           int32_t capacity_points, int32_t* has_depth, int32_t* n_with_depth) {
    if (!c || n < 0 || (n > 0 && (!keypoints_xy || !has_depth))) return fail(VELO_ERR_INVALID, "bad keypoint arguments");
    if (!c->have_projection) return fail(VELO_ERR_STATE, "velo_project_lidar has not run");
    if (n_with_depth) *n_with_depth = 0;
    if (n == 0) return VELO_OK;
    HIP_TRY(hipSetDevice(c->device));
    VELO_TRY(c->kps.reserve((size_t)n)); VELO_TRY(c->kp_point.reserve((size_t)n)); VELO_TRY(c->kp_out.reserve((size_t)n));
    VELO_TRY(c->kp_flag.reserve((size_t)n + 2)); VELO_TRY(c->kp_excl.reserve((size_t)n + 2)); VELO_TRY(c->kp_has.reserve((size_t)n));
    VELO_TRY(c->cursor.reserve((size_t)n + 2));
    const int n_tiles = cdiv(n, kScanTile);
    VELO_TRY(c->scan_tiles.reserve((size_t)n_tiles + 1));
    VELO_TRY(c->scan_total.reserve(1));
    HIP_TRY(hipMemcpyAsync(c->kps.p, keypoints_xy, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(depth_assoc_kernel, dim3(cdiv(n, 4)), dim3(256), 0, c->stream, (const float2*)c->kps.p, n, (const float4*)c->pstack.p,
                       (const float4*)c->vstack.p, (const int*)c->proj_off.p, (const int*)c->ring_cnt.p, c->proj_rings, thresh, c->kp_point.p, c->kp_flag.p);
    HIP_TRY(hipMemcpyAsync(c->kp_excl.p, c->kp_flag.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(n_tiles), dim3(kScanThreads), 0, c->stream, c->kp_excl.p, n, c->scan_tiles.p);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanThreads), 0, c->stream, c->scan_tiles.p, n_tiles, c->scan_total.p);
    hipLaunchKernelGGL(scan_add_kernel, dim3(cdiv(n + 1, 256)), dim3(256), 0, c->stream, c->kp_excl.p, n, c->scan_tiles.p, c->scan_total.p, c->cursor.p);
    hipLaunchKernelGGL(depth_compact_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, (const int*)c->kp_flag.p, (const int*)c->kp_excl.p,
                       (const float4*)c->kp_point.p, n, c->kp_has.p, c->kp_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(has_depth, c->kp_has.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_int, c->scan_total.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int nd = c->h_int[0];
    if (n_with_depth) *n_with_depth = nd;
    if (kp_with_depth_xyz && capacity_points > 0 && nd > 0) {
        std::vector<float4> h((size_t)nd);
        HIP_TRY(hipMemcpy(h.data(), c->kp_out.p, sizeof(float4) * (size_t)nd, hipMemcpyDeviceToHost));
        for (int i = 0; i < nd && i < capacity_points; i++) { kp_with_depth_xyz[3 * i] = h[i].x; kp_with_depth_xyz[3 * i + 1] = h[i].y; kp_with_depth_xyz[3 * i + 2] = h[i].z; }
    }
    return VELO_OK;
}

// ---- batched triangulatePoint (velo.h:1027-1130) ----------------------------------------------------
int velo_triangulate_points(velo_ctx* c, const double* camera_poses, int32_t n_frames, const float* cam_trans, int32_t n_cams,
                            const velo_tri_obs* obs, const int32_t* obs_offsets, int32_t n, float* points_xyz,
                            const uint8_t* initial_guess, velo_tri_result* results) {
    if (!c || n < 0 || n_frames < 0 || n_cams < 0) return fail(VELO_ERR_INVALID, "null ctx / negative size");
    if (n == 0) return VELO_OK;
    if (!obs_offsets || !points_xyz) return fail(VELO_ERR_INVALID, "null offsets / points");
    if (obs_offsets[0] != 0) return fail(VELO_ERR_INVALID, "obs_offsets[0] must be 0");
    for (int l = 0; l < n; l++) if (obs_offsets[l + 1] < obs_offsets[l]) return fail(VELO_ERR_INVALID, "obs_offsets must not decrease (landmark %d)", l);
    const int n_obs = obs_offsets[n];
    if (n_obs > 0 && (!obs || !camera_poses)) return fail(VELO_ERR_INVALID, "null observations / poses");
    for (int k = 0; k < n_obs; k++) {
        const velo_tri_obs& o = obs[k];
        if (o.kind != VELO_TRI_OBS_3D && o.kind != VELO_TRI_OBS_2D) return fail(VELO_ERR_INVALID, "observation %d: unknown kind %d", k, o.kind);
        if (o.frame < 0 || o.frame >= n_frames) return fail(VELO_ERR_INVALID, "observation %d: frame %d outside [0, %d)", k, o.frame, n_frames);
        if (o.kind == VELO_TRI_OBS_2D && (o.cam < 0 || o.cam >= n_cams || !cam_trans)) return fail(VELO_ERR_INVALID, "observation %d: camera %d outside [0, %d)", k, o.cam, n_cams);
    }
    HIP_TRY(hipSetDevice(c->device));
    // per-frame constants in double with the host libm: rot = -pose[0..2] (costfunctions.h:318-320), Rodrigues scalars, R columns
    std::vector<TriFrame> hf((size_t)std::max(n_frames, 1));
    for (int f = 0; f < n_frames; f++) {
        const double* cp = camera_poses + 6 * (size_t)f;
        const double xr[6] = {-cp[0], -cp[1], -cp[2], 0.0, 0.0, 0.0};
        PoseScalars S;
        pose_scalars(xr, &S);
        TriFrame& F = hf[(size_t)f];
        std::memset(&F, 0, sizeof(F));
        for (int k = 0; k < 3; k++) { F.w[k] = S.w[k]; F.u[k] = S.u[k]; F.center[k] = cp[3 + k]; }
        F.c = S.c; F.s = S.s; F.omc = S.omc; F.small = S.small;
        for (int j = 0; j < 3; j++) {                                // column j = rotation of e_j, same operation order as the device form
            double e[3] = {0.0, 0.0, 0.0}, o[3];
            e[j] = 1.0;
            if (!F.small) {
                const double c0 = F.u[1] * e[2] - F.u[2] * e[1], c1 = F.u[2] * e[0] - F.u[0] * e[2], c2 = F.u[0] * e[1] - F.u[1] * e[0];
                const double tmp = (F.u[0] * e[0] + F.u[1] * e[1] + F.u[2] * e[2]) * F.omc;
                o[0] = e[0] * F.c + c0 * F.s + F.u[0] * tmp;
                o[1] = e[1] * F.c + c1 * F.s + F.u[1] * tmp;
                o[2] = e[2] * F.c + c2 * F.s + F.u[2] * tmp;
            } else {
                o[0] = e[0] + (F.w[1] * e[2] - F.w[2] * e[1]);
                o[1] = e[1] + (F.w[2] * e[0] - F.w[0] * e[2]);
                o[2] = e[2] + (F.w[0] * e[1] - F.w[1] * e[0]);
            }
            F.R[0 * 3 + j] = o[0]; F.R[1 * 3 + j] = o[1]; F.R[2 * 3 + j] = o[2];
        }
    }
    std::vector<double> hct((size_t)std::max(3 * n_cams, 3), 0.0);
    for (int k = 0; k < 3 * n_cams; k++) hct[(size_t)k] = (double)cam_trans[k];
    VELO_TRY(c->tri_frames.reserve(hf.size())); VELO_TRY(c->tri_cam_t.reserve(hct.size()));
    VELO_TRY(c->tri_obs.reserve((size_t)std::max(n_obs, 1))); VELO_TRY(c->tri_off.reserve((size_t)n + 1));
    VELO_TRY(c->tri_pts.reserve((size_t)3 * n)); VELO_TRY(c->tri_init.reserve((size_t)n)); VELO_TRY(c->tri_res.reserve((size_t)n));
    HIP_TRY(hipMemcpyAsync(c->tri_frames.p, hf.data(), sizeof(TriFrame) * hf.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tri_cam_t.p, hct.data(), sizeof(double) * hct.size(), hipMemcpyHostToDevice, c->stream));
    // blocks enter the problem 3-D first (velo.h:1049-1122): stable partition per landmark, so that block position == index
    std::vector<velo_tri_obs> hobs((size_t)std::max(n_obs, 1));
    for (int l = 0; l < n; l++) {
        int w = obs_offsets[l];
        for (int pass = 0; pass < 2; pass++)
            for (int k = obs_offsets[l]; k < obs_offsets[l + 1]; k++)
                if ((obs[k].kind == VELO_TRI_OBS_2D) == (pass == 1)) hobs[(size_t)w++] = obs[k];
    }
    if (n_obs > 0) HIP_TRY(hipMemcpyAsync(c->tri_obs.p, hobs.data(), sizeof(velo_tri_obs) * (size_t)n_obs, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tri_off.p, obs_offsets, sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tri_pts.p, points_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (initial_guess) HIP_TRY(hipMemcpyAsync(c->tri_init.p, initial_guess, (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));                         // hf / hct are stack-owned host vectors
    TriParams P;
    P.lm = lm_params(c->P);
    P.loss_a = c->P.loss_thresh_3D2D; P.loss_w = c->P.weight_3D2D;    // velo.h:1116-1119
    if (c->tri_variant == 0)     // VELO_TRI_VARIANT=0: one thread per landmark (kept for A/B; same results)
        hipLaunchKernelGGL(triangulate_kernel, dim3(cdiv(n, 64)), dim3(64), 0, c->stream, (const TriFrame*)c->tri_frames.p, (const double*)c->tri_cam_t.p,
                           (const velo_tri_obs*)c->tri_obs.p, (const int*)c->tri_off.p, n, P, c->tri_pts.p,
                           (const unsigned char*)(initial_guess ? c->tri_init.p : nullptr), c->tri_res.p);
    else
        hipLaunchKernelGGL(triangulate_wave_kernel, dim3(n), dim3(64), 0, c->stream, (const TriFrame*)c->tri_frames.p, (const double*)c->tri_cam_t.p,
                           (const velo_tri_obs*)c->tri_obs.p, (const int*)c->tri_off.p, n, P, c->tri_pts.p,
                           (const unsigned char*)(initial_guess ? c->tri_init.p : nullptr), c->tri_res.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(points_xyz, c->tri_pts.p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (results) HIP_TRY(hipMemcpyAsync(results, c->tri_res.p, sizeof(velo_tri_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VELO_OK;
}

}  // extern "C"
