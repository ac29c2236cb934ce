// velo_frame_to_frame.hpp -- header-only C++ adaptor: the reference's frameToFrame(...) parameter list
// (velo.h:598-614, sole caller main.cpp:388-405) on top of the C-ABI of include/velo_hip.h.
//
// It is a template over the container types so that it compiles against the reference's own types
// (pcl::PointCloud<pcl::PointXYZ>::Ptr, pcl::KdTreeFLANN, cv::Point2f, Eigen::Matrix4d) when those headers are present,
// and against light stand-ins (tests/cpp/standins.hpp) when they are not -- this image has neither PCL, OpenCV nor Eigen.
// Requirements on the types, all satisfied by the reference's:
//   CloudPtr   : cloud->size(), cloud->at(i).x/.y/.z                       (pcl::PointCloud<pcl::PointXYZ>::Ptr)
//   Point2     : p.x, p.y                                                  (cv::Point2f)
//   Point3     : p.x, p.y, p.z                                             (pcl::PointXYZ)
//   Mat4       : T(i, j) assignable double                                 (Eigen::Matrix4d)
//   KdTrees    : ignored -- the per-ring KD-trees (lru.h:17-20) are replaced by the device-side grid index that
//                velo_set_target builds; the argument is kept so that the call site does not change.
//
// What runs where:  this adaptor flattens the reference's nested containers (host), exactly as velo.h:627-654 gathers
// the per-match operands; residual-type selection + outlier gate (velo.h:662-792), association (velo.h:806-894),
// residual/Jacobian evaluation (costfunctions.h) and the Ceres solve (velo.h:897-902) run on the GPU behind
// velo_frame_to_frame().  Errors: the reference reports none; here a failed call throws std::runtime_error with
// velo_last_error() (the C-ABI itself never throws).
#ifndef VELO_FRAME_TO_FRAME_HPP_
#define VELO_FRAME_TO_FRAME_HPP_

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "velo_hip.h"

namespace velo_hip {

// same enumerators, same order as the reference's ResidualType (velo.h:3-8)
enum ResidualType { RESIDUAL_3D3D = VELO_RESIDUAL_3D3D, RESIDUAL_3D2D = VELO_RESIDUAL_3D2D,
                    RESIDUAL_2D3D = VELO_RESIDUAL_2D3D, RESIDUAL_2D2D = VELO_RESIDUAL_2D2D };

inline void check(int status, const char* what) {
    if (status != VELO_OK) throw std::runtime_error(std::string(what) + ": " + velo_last_error());
}

// RAII owner of one velo_ctx (one per host thread, like the reference's single-threaded driver loop).
class Context {
public:
    explicit Context(int device = 0) { check(velo_create(&ctx_, device), "velo_create"); }
    ~Context() { velo_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    velo_ctx* get() const { return ctx_; }
    velo_params params() const { velo_params p; check(velo_get_params(ctx_, &p), "velo_get_params"); return p; }
    void set_params(const velo_params& p) { check(velo_set_params(ctx_, &p), "velo_set_params"); }

    // scans of one frame: a vector of ring clouds, as ScanData::scans holds them (lru.h:8)
    template <typename CloudPtr>
    static void flatten(const std::vector<CloudPtr>& rings, std::vector<float>& xyz, std::vector<int32_t>& off) {
        off.assign(1, 0);
        size_t n = 0;
        for (const auto& r : rings) n += r->size();
        xyz.resize(3 * n);
        size_t k = 0;
        for (const auto& r : rings) {
            for (size_t i = 0; i < r->size(); i++) {
                const auto& p = r->at(i);
                xyz[3 * k] = p.x; xyz[3 * k + 1] = p.y; xyz[3 * k + 2] = p.z;
                k++;
            }
            off.push_back((int32_t)k);
        }
    }
    template <typename CloudPtr>
    void set_target(const std::vector<CloudPtr>& scans_S) {       // replaces scans_S + kd_trees
        flatten(scans_S, xyz_, off_);
        check(velo_set_target(ctx_, xyz_.data(), 12, off_.data(), (int32_t)off_.size() - 1, 0), "velo_set_target");
    }
    template <typename CloudPtr>
    void set_source(const std::vector<CloudPtr>& scans_M) {
        flatten(scans_M, xyz_, off_);
        check(velo_set_source(ctx_, xyz_.data(), 12, off_.data(), (int32_t)off_.size() - 1, 0), "velo_set_source");
    }

    // the scan held as source (frame k) becomes the target of the next registration without another upload: the role of
    // scans_lru.get(frame - 1) in the reference's loop (main.cpp:380)
    void source_to_target() { check(velo_source_to_target(ctx_), "velo_source_to_target"); }

private:
    velo_ctx* ctx_ = nullptr;
    std::vector<float> xyz_;
    std::vector<int32_t> off_;
};

// ScansLRU (lru.h:31-61) with the scans kept on the device.  `get` has the reference's meaning -- "give me frame f, reading it
// only if it is not cached" -- with the reading supplied by the caller: load(frame) must return the ring clouds of that frame
// (what `new ScanData(dataset, frame)` does at lru.h:49); it runs on a miss only.  The scan lands in `ctx` as target or source.
class ScanCache {
public:
    explicit ScanCache(int device = 0, int capacity = 50) { check(velo_cache_create(&cache_, device, capacity), "velo_cache_create"); }
    ~ScanCache() { velo_cache_destroy(cache_); }
    ScanCache(const ScanCache&) = delete;
    ScanCache& operator=(const ScanCache&) = delete;
    velo_scan_cache* get() const { return cache_; }
    bool contains(int frame) const { return velo_cache_contains(cache_, frame) != 0; }

    // returns true on a hit.  On a miss the scan is uploaded (and, as target, indexed) once and stored for the next look-up.
    template <typename LoadRings>
    bool get(int frame, Context& ctx, bool as_target, LoadRings load) {
        if (contains(frame)) {
            check(velo_cache_load(cache_, frame, ctx.get(), as_target ? 1 : 0), "velo_cache_load");
            return true;
        }
        if (as_target) ctx.set_target(load(frame)); else ctx.set_source(load(frame));
        check(velo_cache_store(cache_, frame, ctx.get(), as_target ? 1 : 0), "velo_cache_store");
        return false;
    }
    // after a registration: keep the scan `ctx` holds (e.g. the current frame, held as source) for later frames
    void put(int frame, Context& ctx, bool of_target) { check(velo_cache_store(cache_, frame, ctx.get(), of_target ? 1 : 0), "velo_cache_store"); }

private:
    velo_scan_cache* cache_ = nullptr;
};

// The reference's globals the path reads (kitti.h:3,46-47): number of cameras and cam_trans[cam]; for the depth rows also
// the canonical-coordinate window {min_x, max_x, min_y, max_y}[cam] (kitti.h:51,85-97) and depth_assoc_thresh (kitti.h:28).
struct Rig {
    int num_cams = 2;
    std::vector<std::array<float, 3>> cam_trans{{{0.f, 0.f, 0.f}}, {{-0.537f, 0.f, 0.f}}};
    std::vector<std::array<double, 4>> window{{{-0.84466541, 0.8608222, -0.25765342, 0.25705326}},
                                              {{-0.84466541, 0.8608222, -0.25765342, 0.25705326}}};
    double depth_assoc_thresh = 0.015;
};

// frameToFrame, same parameter list and meaning as velo.h:598-614.  `ctx` and `rig` are the two additions (the
// reference keeps their equivalents in globals).  Returns the 4x4 of the solution like util::pose_mat2vec(transform).
template <typename Mat4, typename Point2, typename Point3, typename CloudPtr, typename KdTrees, typename ResidualT>
Mat4 frameToFrame(Context& ctx, const Rig& rig,
                  const std::vector<std::vector<std::pair<int, int>>>& matches,
                  const std::vector<std::vector<std::vector<Point2>>>& keypoints,
                  const std::vector<std::vector<std::vector<int>>>& keypoint_ids,
                  const std::map<int, Point3>& landmarks_at_frame,
                  const std::vector<std::vector<CloudPtr>>& keypoints_with_depth,
                  const std::vector<std::vector<std::vector<int>>>& has_depth,
                  const std::vector<CloudPtr>& scans_M,
                  const std::vector<CloudPtr>& scans_S,
                  const KdTrees& /*kd_trees: superseded by the device grid*/,
                  const int frame1, const int frame2,
                  double transform[6],
                  std::vector<std::vector<std::pair<int, int>>>& good_matches,
                  std::vector<std::vector<ResidualT>>& residual_type,      // the reference's own enum ResidualType (velo.h:3-8) or the one above
                  const bool enable_icp) {
    velo_params P = ctx.params();
    P.enable_icp = enable_icp ? 1 : 0;                                            // velo.h:806
    ctx.set_params(P);
    // An EMPTY ring vector means "the scan this context already holds" -- loaded from a ScanCache, or promoted by
    // source_to_target(); the reference never passes one (a frame has ~64 rings), so its call sites keep their meaning.
    if (!scans_S.empty()) ctx.set_target(scans_S);
    if (!scans_M.empty()) ctx.set_source(scans_M);

    // velo.h:622-654: gather, per match, what the residual-type selection needs
    std::vector<velo_match> recs;
    for (int cam = 0; cam < rig.num_cams; cam++) {
        const auto& mc = matches[cam];
        for (size_t i = 0; i < mc.size(); i++) {
            const int point1 = mc[i].first, point2 = mc[i].second;
            const int id = keypoint_ids[cam][frame2][point2];
            bool d1 = has_depth[cam][frame1][point1] != -1, d2 = has_depth[cam][frame2][point2] != -1;
            velo_match m;
            std::memset(&m, 0, sizeof(m));
            auto lm = landmarks_at_frame.find(id);
            if (lm != landmarks_at_frame.end()) {                                 // velo.h:634-644
                m.p3_2[0] = lm->second.x; m.p3_2[1] = lm->second.y; m.p3_2[2] = lm->second.z;
                d2 = true;
            } else if (d2) {
                const auto& p = keypoints_with_depth[cam][frame2]->at(has_depth[cam][frame2][point2]);
                m.p3_2[0] = p.x; m.p3_2[1] = p.y; m.p3_2[2] = p.z;
            }
            if (d1) {
                const auto& p = keypoints_with_depth[cam][frame1]->at(has_depth[cam][frame1][point1]);
                m.p3_1[0] = p.x; m.p3_1[1] = p.y; m.p3_1[2] = p.z;
            }
            const Point2& a = keypoints[cam][frame1][point1];
            const Point2& b = keypoints[cam][frame2][point2];
            m.p2_1[0] = a.x; m.p2_1[1] = a.y; m.p2_2[0] = b.x; m.p2_2[1] = b.y;
            for (int k = 0; k < 3; k++) m.t_cam[k] = rig.cam_trans[cam][k];
            m.cam = cam; m.point1 = point1; m.point2 = point2;
            m.d1 = d1 ? 1 : 0; m.d2 = d2 ? 1 : 0;
            recs.push_back(m);
        }
    }
    check(velo_set_visual(ctx.get(), recs.empty() ? nullptr : recs.data(), (int32_t)recs.size()), "velo_set_visual");

    double T[16];
    check(velo_frame_to_frame(ctx.get(), transform, T, nullptr), "velo_frame_to_frame");

    // good_matches / residual_type of the LAST outer iteration (cleared per cam per iter at velo.h:624-625)
    int32_t n = 0;
    check(velo_get_good_matches(ctx.get(), nullptr, 0, &n), "velo_get_good_matches");
    std::vector<velo_good_match> gm((size_t)n);
    if (n > 0) check(velo_get_good_matches(ctx.get(), gm.data(), n, &n), "velo_get_good_matches");
    for (int cam = 0; cam < rig.num_cams; cam++) { good_matches[cam].clear(); residual_type[cam].clear(); }
    for (const auto& g : gm) {
        good_matches[g.cam].push_back(std::make_pair(g.point1, g.point2));
        residual_type[g.cam].push_back((ResidualT)g.residual_type);
    }
    Mat4 out;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out(i, j) = T[i * 4 + j];
    return out;
}

// ---- the reference's EXACT parameter list (velo.h:598-614): callable unchanged from main.cpp:388-405 -----------------------------
// The reference keeps the camera rig (kitti.h:3,46-51) and the device (cv::cuda::setDevice, main.cpp:64) in globals; so does this
// overload: a process-default Context (created on first use on device VELO_HIP_DEFAULT_DEVICE, default 0) and a process-default Rig
// (KITTI's, editable through default_rig() before the first call).  Like the reference's globals they are not thread-safe: one
// driver thread, as in main.cpp.  The return type is the reference's Eigen::Matrix4d when Eigen was included before this header;
// define VELO_HIP_MAT4 to any 4x4 type with T(i, j) to override (the tests do: this image has no Eigen).
#ifndef VELO_HIP_DEFAULT_DEVICE
#define VELO_HIP_DEFAULT_DEVICE 0
#endif
inline Context& default_context() { static Context ctx(VELO_HIP_DEFAULT_DEVICE); return ctx; }
inline Rig& default_rig() { static Rig rig; return rig; }
#if !defined(VELO_HIP_MAT4) && defined(EIGEN_WORLD_VERSION)
#define VELO_HIP_MAT4 Eigen::Matrix4d
#endif
#ifdef VELO_HIP_MAT4
template <typename Point2, typename Point3, typename CloudPtr, typename KdTrees, typename ResidualT>
VELO_HIP_MAT4 frameToFrame(const std::vector<std::vector<std::pair<int, int>>>& matches,
                           const std::vector<std::vector<std::vector<Point2>>>& keypoints,
                           const std::vector<std::vector<std::vector<int>>>& keypoint_ids,
                           const std::map<int, Point3>& landmarks_at_frame,
                           const std::vector<std::vector<CloudPtr>>& keypoints_with_depth,
                           const std::vector<std::vector<std::vector<int>>>& has_depth,
                           const std::vector<CloudPtr>& scans_M,
                           const std::vector<CloudPtr>& scans_S,
                           const KdTrees& kd_trees,
                           const int frame1, const int frame2,
                           double transform[6],
                           std::vector<std::vector<std::pair<int, int>>>& good_matches,
                           std::vector<std::vector<ResidualT>>& residual_type,
                           const bool enable_icp) {
    return frameToFrame<VELO_HIP_MAT4>(default_context(), default_rig(), matches, keypoints, keypoint_ids, landmarks_at_frame, keypoints_with_depth,
                                       has_depth, scans_M, scans_S, kd_trees, frame1, frame2, transform, good_matches, residual_type, enable_icp);
}
#endif

// projectLidarToCamera, velo.h:329-334.  The rings are the ones `ctx` already holds -- `of_target` says in which slot (the
// frame just loaded for frameToFrame), so nothing is uploaded again; `projection` and `scans_valid` are filled like the
// reference fills them (one list per ring) for callers that draw or inspect them.  The lists also stay on the device,
// which is what featureDepthAssociation below searches.
template <typename Point2, typename Cloud>
void projectLidarToCamera(Context& ctx, const Rig& rig, bool of_target, std::vector<std::vector<Point2>>& projection,
                          std::vector<std::shared_ptr<Cloud>>& scans_valid, const int cam) {
    int32_t n = 0, nr = 0;
    check(velo_project_lidar(ctx.get(), of_target ? 1 : 0, rig.cam_trans[cam].data(), rig.window[cam].data(), &n), "velo_project_lidar");
    check(velo_get_projection(ctx.get(), nullptr, nullptr, 0, nullptr, 0, &nr), "velo_get_projection");
    std::vector<int32_t> off((size_t)nr + 1);
    std::vector<float> xy(2 * (size_t)n + 2), pts(3 * (size_t)n + 3);
    check(velo_get_projection(ctx.get(), xy.data(), pts.data(), n, off.data(), nr + 1, &nr), "velo_get_projection");
    for (int s = 0; s < nr; s++) {
        projection.push_back(std::vector<Point2>());
        scans_valid.push_back(std::shared_ptr<Cloud>(new Cloud));
        for (int j = off[s]; j < off[s + 1]; j++) {
            Point2 c; c.x = xy[2 * j]; c.y = xy[2 * j + 1];
            projection.back().push_back(c);
            typename std::remove_const<typename std::remove_reference<decltype(scans_valid.back()->at(0))>::type>::type p;
            p.x = pts[3 * j]; p.y = pts[3 * j + 1]; p.z = pts[3 * j + 2];
            scans_valid.back()->push_back(p);
        }
    }
}

// featureDepthAssociation, velo.h:376-382, against the last projectLidarToCamera of `ctx` (the reference passes the lists
// back in; here they never left the device).  Appends to keypoints_with_depth, fills and returns has_depth.
template <typename Point2, typename CloudPtr>
std::vector<int> featureDepthAssociation(Context& ctx, const Rig& rig, const std::vector<Point2>& keypoints,
                                         CloudPtr keypoints_with_depth, std::vector<int>& has_depth) {
    static_assert(sizeof(Point2) == 2 * sizeof(float), "keypoints must be packed (x, y) floats like cv::Point2f");
    has_depth.assign(keypoints.size(), -1);
    if (keypoints.empty()) return has_depth;
    std::vector<int32_t> has(keypoints.size());
    std::vector<float> xyz(3 * keypoints.size());
    int32_t n3d = 0;
    check(velo_depth_association(ctx.get(), &keypoints[0].x, (int32_t)keypoints.size(), rig.depth_assoc_thresh, xyz.data(),
                                 (int32_t)keypoints.size(), has.data(), &n3d), "velo_depth_association");
    for (size_t k = 0; k < keypoints.size(); k++) has_depth[k] = has[k];
    for (int i = 0; i < n3d; i++) {
        typename std::remove_const<typename std::remove_reference<decltype(keypoints_with_depth->at(0))>::type>::type p;
        p.x = xyz[3 * i]; p.y = xyz[3 * i + 1]; p.z = xyz[3 * i + 2];
        keypoints_with_depth->push_back(p);
    }
    return has_depth;
}

// The loop at main.cpp:661-671 -- triangulatePoint (velo.h:1027-1033) for every id in `ids` -- as ONE device call.
// Per landmark the arguments are the reference's: keypoint_obs2[id][cam] / keypoint_obs3[id][cam] are the per-camera
// std::map<int frame, observation>, camera_poses[frame][0..5] the pose vectors, landmarks->at(id) the point (read as initial
// guess when keypoint_added[id], always written), all borrowed for the call.  Observations are flattened in the order the
// reference adds its residual blocks: 3-D ones camera-major in map order, then 2-D ones camera-major in map order.
template <typename Obs2, typename Obs3, typename Poses, typename CloudPtr, typename Flags>
void triangulatePoints(Context& ctx, const Rig& rig, const std::vector<int>& ids, const Obs2& keypoint_obs2, const Obs3& keypoint_obs3,
                       const Poses& camera_poses, int n_frames, CloudPtr landmarks, const Flags& keypoint_added) {
    if (ids.empty()) return;
    std::vector<velo_tri_obs> obs;
    std::vector<int32_t> off(1, 0);
    std::vector<float> pts(3 * ids.size());
    std::vector<uint8_t> init(ids.size());
    for (size_t l = 0; l < ids.size(); l++) {
        const int id = ids[l];
        for (int cam = 0; cam < rig.num_cams; cam++)
            for (const auto& o3 : keypoint_obs3[id][cam]) {
                velo_tri_obs o; o.kind = VELO_TRI_OBS_3D; o.frame = o3.first; o.cam = cam;
                o.s[0] = o3.second.x; o.s[1] = o3.second.y; o.s[2] = o3.second.z;
                obs.push_back(o);
            }
        for (int cam = 0; cam < rig.num_cams; cam++)
            for (const auto& o2 : keypoint_obs2[id][cam]) {
                velo_tri_obs o; o.kind = VELO_TRI_OBS_2D; o.frame = o2.first; o.cam = cam;
                o.s[0] = o2.second.x; o.s[1] = o2.second.y; o.s[2] = 0.f;
                obs.push_back(o);
            }
        off.push_back((int32_t)obs.size());
        const auto& p = landmarks->at(id);
        pts[3 * l] = p.x; pts[3 * l + 1] = p.y; pts[3 * l + 2] = p.z;
        init[l] = keypoint_added[id] ? 1 : 0;
    }
    std::vector<double> poses(6 * (size_t)n_frames);
    for (int f = 0; f < n_frames; f++) for (int i = 0; i < 6; i++) poses[6 * (size_t)f + i] = camera_poses[f][i];
    std::vector<float> ct(3 * (size_t)rig.num_cams);
    for (int cam = 0; cam < rig.num_cams; cam++) for (int i = 0; i < 3; i++) ct[3 * (size_t)cam + i] = rig.cam_trans[cam][i];
    check(velo_triangulate_points(ctx.get(), poses.data(), n_frames, ct.data(), rig.num_cams, obs.empty() ? nullptr : obs.data(), off.data(),
                                  (int32_t)ids.size(), pts.data(), init.data(), nullptr), "velo_triangulate_points");
    for (size_t l = 0; l < ids.size(); l++) {
        auto& p = landmarks->points[ids[l]];
        p.x = pts[3 * l]; p.y = pts[3 * l + 1]; p.z = pts[3 * l + 2];
    }
}

// ---- the pose chain of the drive loop (main.cpp:179,305-331,407-437): host arithmetic, no GPU ----------------------------------------
// What main.cpp does around frameToFrame, as one small object: the constant-velocity guess from the last two poses
// (main.cpp:311-320,331; {0,0,0,0,0,1} for the first pair, main.cpp:170), the chained pose ceres_poses_mat[frame] =
// ceres_poses_mat[frame-1] * dpose (main.cpp:408), the agreement of a registration with its prediction (main.cpp:416-424) and the rule
// that skips an edge over dframe > 1 frames on poor agreement (main.cpp:426-437, thresholds kitti.h:33-35).  4x4s are row-major doubles.
class PoseChain {
public:
    PoseChain() { std::array<double, 16> I{}; I[0] = I[5] = I[10] = I[15] = 1.0; poses_.push_back(I); }
    size_t frames() const { return poses_.size(); }
    const std::array<double, 16>& pose(size_t k) const { return poses_.at(k); }
    // the guess frameToFrame starts from for frame `frames()` against frame `frames() - 1`  (the array main.cpp calls `transform`)
    void predict(double transform[6]) const {
        if (poses_.size() < 2) { const double first[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 1.0}; std::memcpy(transform, first, sizeof(first)); return; }
        std::memcpy(transform, next_.data(), sizeof(double) * 6);
    }
    // the registration of the new frame returned dpose: chain it (main.cpp:408); returns the agreement 6-vector with the prediction
    std::array<double, 6> push(const double dpose[16]) {
        double guess[6], dT[16];
        predict(guess);
        check(velo_pose_vec_to_mat(guess, dT), "velo_pose_vec_to_mat");                       // main.cpp:315-319: what the prediction was, as a matrix
        std::array<double, 6> ag = agreement(dpose, dT);
        std::array<double, 16> P = poses_.back();
        check(velo_pose_handoff(1, P.data(), dpose, next_.data()), "velo_pose_handoff");      // main.cpp:408 + the next frame's guess (main.cpp:311-331)
        poses_.push_back(P);
        return ag;
    }
    // main.cpp:416-417: pose_vec2mat(dpose * dT^-1), dT rigid
    static std::array<double, 6> agreement(const double dpose[16], const double dT[16]) {
        double inv[16] = {0}, M[16];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inv[4 * i + j] = dT[4 * j + i];            // [R^T | -R^T t]
        for (int i = 0; i < 3; i++) inv[4 * i + 3] = -(inv[4 * i] * dT[3] + inv[4 * i + 1] * dT[7] + inv[4 * i + 2] * dT[11]);
        inv[15] = 1.0;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double v = 0.0; for (int k = 0; k < 4; k++) v += dpose[4 * i + k] * inv[4 * k + j]; M[4 * i + j] = v; }
        std::array<double, 6> a;
        check(velo_pose_mat_to_vec(M, a.data()), "velo_pose_mat_to_vec");
        return a;
    }
    // main.cpp:426-437 (odometry pass): nullptr = keep the edge, else why it is skipped.  dframe == 1 edges are never skipped.
    static const char* edge_rejected(const std::array<double, 6>& a, int dframe, double agreement_t_thresh = 0.1, double agreement_r_thresh = 0.05,
                                     double loop_close_thresh = 10.0) {
        if (dframe <= 1) return nullptr;
        const double t = std::sqrt(a[3] * a[3] + a[4] * a[4] + a[5] * a[5]), r = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        const double lim = agreement_t_thresh * dframe < loop_close_thresh ? agreement_t_thresh * dframe : loop_close_thresh;
        if (t > lim) return "poor t agreement";
        if (r > agreement_r_thresh) return "poor r agreement";
        return nullptr;
    }

private:
    std::vector<std::array<double, 16>> poses_;      // ceres_poses_mat (main.cpp:179), pose 0 = identity
    std::array<double, 6> next_{};                   // the guess for the next pair, from velo_pose_handoff
};

}  // namespace velo_hip
#endif  // VELO_FRAME_TO_FRAME_HPP_
