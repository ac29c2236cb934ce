// Drives include/velo_frame_to_frame.hpp the way main.cpp:388-405 drives the reference's frameToFrame: nested
// containers in, transform[6] in/out, good_matches/residual_type out.  Input file written by tests/test_cpp_adaptor.py.
#include <cstdio>
#include <cstdlib>
#include <array>
#include <map>
#include <string>
#include <vector>

#include "standins.hpp"
#define VELO_HIP_MAT4 standin::Matrix4d          // no Eigen in this image: the exact-signature overload returns the stand-in 4x4
#include "velo_frame_to_frame.hpp"

using namespace standin;

// the reference's own global enum (velo.h:3-8): the exact-signature overload fills vectors of THIS type
enum ResidualType { RESIDUAL_3D3D, RESIDUAL_3D2D, RESIDUAL_2D3D, RESIDUAL_2D2D };

static std::vector<PointCloud::Ptr> read_rings(FILE* f) {
    int32_t nr;
    if (fread(&nr, 4, 1, f) != 1) exit(2);
    std::vector<int32_t> off(nr + 1);
    if (fread(off.data(), 4, nr + 1, f) != (size_t)nr + 1) exit(2);
    std::vector<float> xyz(3 * (size_t)off[nr]);
    if (fread(xyz.data(), 4, xyz.size(), f) != xyz.size()) exit(2);
    std::vector<PointCloud::Ptr> rings;
    for (int r = 0; r < nr; r++) {
        PointCloud::Ptr c(new PointCloud);
        for (int i = off[r]; i < off[r + 1]; i++) c->push_back(PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        rings.push_back(c);
    }
    return rings;
}

// second mode: `test_adaptor --tri file` drives velo_hip::triangulatePoints with the containers main.cpp:640-671 holds
static int run_triangulation(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return 1;
    int32_t n_frames, n_l;
    if (fread(&n_frames, 4, 1, f) != 1) return 2;
    std::vector<std::array<double, 6>> poses(n_frames);
    for (int i = 0; i < n_frames; i++) if (fread(poses[i].data(), 8, 6, f) != 6) return 2;
    if (fread(&n_l, 4, 1, f) != 1) return 2;
    std::vector<int32_t> off(n_l + 1);
    if (fread(off.data(), 4, n_l + 1, f) != (size_t)n_l + 1) return 2;
    std::vector<velo_tri_obs> obs(off[n_l]);
    if (!obs.empty() && fread(obs.data(), sizeof(velo_tri_obs), obs.size(), f) != obs.size()) return 2;
    std::vector<float> p0(3 * (size_t)n_l);
    std::vector<uint8_t> init(n_l);
    if (fread(p0.data(), 4, p0.size(), f) != p0.size() || fread(init.data(), 1, n_l, f) != (size_t)n_l) return 2;
    fclose(f);
    const int num_cams = 2;
    std::vector<std::vector<std::map<int, Point2f>>> obs2(n_l, std::vector<std::map<int, Point2f>>(num_cams));
    std::vector<std::vector<std::map<int, PointXYZ>>> obs3(n_l, std::vector<std::map<int, PointXYZ>>(num_cams));
    PointCloud::Ptr landmarks(new PointCloud);
    std::vector<bool> added(n_l);
    std::vector<int> ids;
    for (int l = 0; l < n_l; l++) {
        for (int k = off[l]; k < off[l + 1]; k++) {
            if (obs[k].kind == VELO_TRI_OBS_3D) obs3[l][obs[k].cam][obs[k].frame] = PointXYZ(obs[k].s[0], obs[k].s[1], obs[k].s[2]);
            else obs2[l][obs[k].cam][obs[k].frame] = Point2f{obs[k].s[0], obs[k].s[1]};
        }
        landmarks->push_back(PointXYZ(p0[3 * l], p0[3 * l + 1], p0[3 * l + 2]));
        added[l] = init[l] != 0;
        ids.push_back(l);
    }
    try {
        velo_hip::Context ctx(0);
        velo_hip::Rig rig;
        velo_hip::triangulatePoints(ctx, rig, ids, obs2, obs3, poses, n_frames, landmarks, added);
        for (int l = 0; l < n_l; l++) printf("p %d %.9g %.9g %.9g\n", l, landmarks->at(l).x, landmarks->at(l).y, landmarks->at(l).z);
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    if (argc >= 3 && std::string(argv[1]) == "--tri") return run_triangulation(argv[2]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<PointCloud::Ptr> scans_M = read_rings(f), scans_S = read_rings(f);
    int32_t nm, skip;
    if (fread(&nm, 4, 1, f) != 1 || fread(&skip, 4, 1, f) != 1) return 2;
    std::vector<velo_match> recs(nm);
    if (nm && fread(recs.data(), sizeof(velo_match), nm, f) != (size_t)nm) return 2;
    double transform[6];
    if (fread(transform, 8, 6, f) != 6) return 2;
    fclose(f);
    double transform0[6];                              // the initial guess, for the second registration of the same pair
    for (int i = 0; i < 6; i++) transform0[i] = transform[i];

    // rebuild the reference-shaped containers: frame1 = 1 (current), frame2 = 0 (previous)
    const int num_cams = 2, frame1 = 1, frame2 = 0;
    std::vector<std::vector<std::pair<int, int>>> matches(num_cams), good_matches(num_cams);
    std::vector<std::vector<velo_hip::ResidualType>> residual_type(num_cams);
    std::vector<std::vector<std::vector<Point2f>>> keypoints(num_cams, std::vector<std::vector<Point2f>>(2));
    std::vector<std::vector<std::vector<int>>> keypoint_ids(num_cams, std::vector<std::vector<int>>(2));
    std::vector<std::vector<std::vector<int>>> has_depth(num_cams, std::vector<std::vector<int>>(2));
    std::vector<std::vector<PointCloud::Ptr>> kwd(num_cams, std::vector<PointCloud::Ptr>(2));
    std::map<int, PointXYZ> landmarks;
    for (int c = 0; c < num_cams; c++) for (int fr = 0; fr < 2; fr++) kwd[c][fr].reset(new PointCloud);
    int next_id = 1000;
    for (int i = 0; i < nm; i++) {
        const velo_match& m = recs[i];
        const int c = m.cam;
        const int k = (int)keypoints[c][frame1].size();
        keypoints[c][frame1].push_back(Point2f{m.p2_1[0], m.p2_1[1]});
        keypoints[c][frame2].push_back(Point2f{m.p2_2[0], m.p2_2[1]});
        const int id = next_id++;
        keypoint_ids[c][frame1].push_back(id);
        keypoint_ids[c][frame2].push_back(id);
        if (m.d1) { has_depth[c][frame1].push_back((int)kwd[c][frame1]->size()); kwd[c][frame1]->push_back(PointXYZ(m.p3_1[0], m.p3_1[1], m.p3_1[2])); }
        else has_depth[c][frame1].push_back(-1);
        if (m.d2 && (i % 5 == 0)) {                    // every fifth 3-D point of frame2 arrives as a triangulated landmark
            landmarks[id] = PointXYZ(m.p3_2[0], m.p3_2[1], m.p3_2[2]);
            has_depth[c][frame2].push_back(-1);
        } else if (m.d2) { has_depth[c][frame2].push_back((int)kwd[c][frame2]->size()); kwd[c][frame2]->push_back(PointXYZ(m.p3_2[0], m.p3_2[1], m.p3_2[2])); }
        else has_depth[c][frame2].push_back(-1);
        matches[c].push_back(std::make_pair(k, k));
    }
    try {
        velo_hip::Context ctx(0);
        velo_params P = ctx.params();
        P.icp_skip = skip;
        ctx.set_params(P);
        velo_hip::Rig rig;
        std::vector<KdTree> kd_trees(scans_S.size());
        Matrix4d T = velo_hip::frameToFrame<Matrix4d>(ctx, rig, matches, keypoints, keypoint_ids, landmarks, kwd, has_depth,
                                                      scans_M, scans_S, kd_trees, frame1, frame2, transform, good_matches,
                                                      residual_type, true);
        printf("x");
        for (int i = 0; i < 6; i++) printf(" %.17g", transform[i]);
        printf("\nT");
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) printf(" %.17g", T(i, j));
        printf("\n");
        for (int c = 0; c < num_cams; c++)
            for (size_t i = 0; i < good_matches[c].size(); i++)
                printf("g %d %d %d %d\n", c, good_matches[c][i].first, good_matches[c][i].second, (int)residual_type[c][i]);
        // The reference's call, character for character (main.cpp:388-405 -> velo.h:598-614): 15 arguments, no context, no rig, the
        // reference's own enum.  The process-default context gets the test's icp_skip (a compile-time constant in kitti.h:8).
        {
            using velo_hip::frameToFrame;
            velo_hip::default_context().set_params(P);
            double t2[6];
            for (int i = 0; i < 6; i++) t2[i] = transform0[i];
            std::vector<std::vector<std::pair<int, int>>> gm_e(num_cams);
            std::vector<std::vector<ResidualType>> rt_e(num_cams);
            Matrix4d Te = frameToFrame(matches, keypoints, keypoint_ids, landmarks, kwd, has_depth, scans_M, scans_S, kd_trees,
                                       frame1, frame2, t2, gm_e, rt_e, true);
            bool same = true;
            for (int i = 0; i < 6; i++) same = same && t2[i] == transform[i];
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) same = same && Te(i, j) == T(i, j);
            for (int c = 0; c < num_cams; c++) {
                same = same && gm_e[c] == good_matches[c] && rt_e[c].size() == residual_type[c].size();
                for (size_t i = 0; same && i < rt_e[c].size(); i++) same = (int)rt_e[c][i] == (int)residual_type[c][i];
            }
            printf("e %d\n", (int)same);
        }
        // the depth rows (main.cpp:594-604) on the target rings the context still holds: the previous frame's keypoints
        rig.depth_assoc_thresh = 0.2;                  // the 128-azimuth test rings are coarser than KITTI's
        for (int c = 0; c < num_cams; c++) {
            std::vector<std::vector<Point2f>> projection;
            std::vector<PointCloud::Ptr> scans_valid;
            velo_hip::projectLidarToCamera(ctx, rig, true, projection, scans_valid, c);
            size_t n_proj = 0;
            for (size_t s = 0; s < projection.size(); s++) {
                if (projection[s].size() != scans_valid[s]->size()) return 4;
                n_proj += projection[s].size();
            }
            PointCloud::Ptr kp3(new PointCloud);
            std::vector<int> hd;
            velo_hip::featureDepthAssociation(ctx, rig, keypoints[c][frame2], kp3, hd);
            printf("d %d %zu %zu %zu\n", c, projection.size(), n_proj, kp3->size());
            for (size_t k = 0; k < hd.size(); k++) {
                if (hd[k] < 0) printf("h %d %zu -1 0 0 0\n", c, k);
                else printf("h %d %zu %d %.9g %.9g %.9g\n", c, k, hd[k], kp3->at(hd[k]).x, kp3->at(hd[k]).y, kp3->at(hd[k]).z);
            }
        }
        // ScansLRU::get with the scans on the device (lru.h:31-61): miss -> read + store, hit -> served from HBM; a LiDAR-only
        // registration against the cached frames must equal the one against fresh uploads
        {
            velo_hip::ScanCache cache(0, 50);
            int reads = 0;
            auto read = [&](int f) -> const std::vector<PointCloud::Ptr>& { reads++; return f == frame2 ? scans_S : scans_M; };
            velo_hip::Context fresh(0), cached(0);
            fresh.set_params(P); cached.set_params(P);
            fresh.set_target(scans_S); fresh.set_source(scans_M);
            double xf[6] = {0, 0, 0, 0, 0, 1}, xc[6] = {0, 0, 0, 0, 0, 1}, Tm[16];
            velo_summary sm;
            velo_hip::check(velo_frame_to_frame(fresh.get(), xf, Tm, &sm), "velo_frame_to_frame");
            const bool h0 = cache.get(frame2, cached, true, read);        // miss: read, indexed, stored
            const bool h1 = cache.get(frame1, cached, false, read);       // miss
            const bool h2 = cache.get(frame2, cached, true, read);        // hit: no read
            const bool h3 = cache.get(frame1, cached, false, read);       // hit
            velo_hip::check(velo_frame_to_frame(cached.get(), xc, Tm, &sm), "velo_frame_to_frame");
            // seam 1 on cached scans: empty ring vectors = "what the context holds"; same call as above otherwise
            double xs[6];
            for (int i = 0; i < 6; i++) xs[i] = transform0[i];
            std::vector<std::vector<std::pair<int, int>>> gm2(num_cams);
            std::vector<std::vector<velo_hip::ResidualType>> rt2(num_cams);
            const std::vector<PointCloud::Ptr> none;
            velo_hip::frameToFrame<Matrix4d>(cached, rig, matches, keypoints, keypoint_ids, landmarks, kwd, has_depth, none, none, kd_trees,
                                             frame1, frame2, xs, gm2, rt2, true);
            bool same = true;
            for (int i = 0; i < 6; i++) same = same && xs[i] == transform[i];
            for (int c = 0; c < num_cams; c++) same = same && gm2[c] == good_matches[c];
            printf("s %d\n", (int)same);
            printf("c %d %d %d %d %d", (int)h0, (int)h1, (int)h2, (int)h3, reads);
            for (int i = 0; i < 6; i++) printf(" %.17g", xf[i]);
            for (int i = 0; i < 6; i++) printf(" %.17g", xc[i]);
            printf("\n");
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
    return 0;
}
