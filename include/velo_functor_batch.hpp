// Seam 2 of SURVEY.md 8(b): the reference's residual functors (costfunctions.h:17-220) are plain structs whose
// `template<class T> bool operator()(const T* x, T* residual) const` Ceres instantiates per block through
// AutoDiffCostFunction<F, dim, 6>.  A per-block call cannot cross to a GPU, so the drop-in at this seam is BY VALUE and
// BATCHED: the functor OBJECTS the caller already constructs (velo.h:683-689, 710-718, 744-752, 777-785, 875-891) are packed
// into velo_functor records -- their public members are the constructor arguments -- and evaluated in one launch
// (velo_evaluate_functors): raw residuals and the Jacobian autodiff would return.
//
// Header-only, C++11, no dependency on the reference's header: the packers are templates over "a struct with these
// members", which is what cost3D3D / cost3D2D / cost2D3D / cost2D2D / cost3DPD are.
#pragma once
#include <vector>

#include "velo_hip.h"

namespace velo_hip {

template <class F> inline velo_functor pack_cost3D3D(const F& f) {      // costfunctions.h:60-90
    velo_functor r = {VELO_RESIDUAL_3D3D, 0, {f.m_x, f.m_y, f.m_z, f.s_x, f.s_y, f.s_z, 0, 0, 0}};
    return r;
}
template <class F> inline velo_functor pack_cost3D2D(const F& f) {      // costfunctions.h:92-130
    velo_functor r = {VELO_RESIDUAL_3D2D, 0, {f.m_x, f.m_y, f.m_z, f.s_x, f.s_y, f.t_x, f.t_y, f.t_z, 0}};
    return r;
}
template <class F> inline velo_functor pack_cost2D3D(const F& f) {      // costfunctions.h:132-172
    velo_functor r = {VELO_RESIDUAL_2D3D, 0, {f.m_x, f.m_y, f.m_z, f.s_x, f.s_y, f.t_x, f.t_y, f.t_z, 0}};
    return r;
}
template <class F> inline velo_functor pack_cost2D2D(const F& f) {      // costfunctions.h:174-220
    velo_functor r = {VELO_RESIDUAL_2D2D, 0, {f.m_x, f.m_y, f.s_x, f.s_y, f.t_x, f.t_y, f.t_z, 0, 0}};
    return r;
}
template <class F> inline velo_functor pack_cost3DPD(const F& f) {      // costfunctions.h:17-58
    velo_functor r = {VELO_FUNCTOR_3DPD, 0, {f.point_x, f.point_y, f.point_z, f.normal_x, f.normal_y, f.normal_z,
                                             f.offset_x, f.offset_y, f.offset_z}};
    return r;
}

inline int functor_dim(const velo_functor& f) {
    return f.kind == VELO_RESIDUAL_3D3D ? 3 : (f.kind == VELO_RESIDUAL_3D2D || f.kind == VELO_RESIDUAL_2D3D) ? 2 : 1;
}

// A list of functors evaluated together.  residual(i) / jacobian(i) point at functor i's rows (dim x 1, dim x 6 row-major).
class FunctorBatch {
public:
    void clear() { recs_.clear(); }
    int size() const { return (int)recs_.size(); }
    int add(const velo_functor& f) { recs_.push_back(f); return (int)recs_.size() - 1; }
    const velo_functor& at(int i) const { return recs_[i]; }
    // VELO_OK or the C-ABI's error code (velo_last_error() has the text)
    int evaluate(velo_ctx* c, const double x[6], bool want_jacobian = true) {
        r_.assign(3 * recs_.size(), 0.0);
        J_.assign(want_jacobian ? 18 * recs_.size() : 0, 0.0);
        return velo_evaluate_functors(c, recs_.empty() ? nullptr : &recs_[0], (int32_t)recs_.size(), x,
                                      r_.empty() ? nullptr : &r_[0], want_jacobian && !J_.empty() ? &J_[0] : nullptr);
    }
    const double* residual(int i) const { return &r_[3 * (size_t)i]; }
    const double* jacobian(int i) const { return J_.empty() ? nullptr : &J_[18 * (size_t)i]; }

private:
    std::vector<velo_functor> recs_;
    std::vector<double> r_, J_;
};

}  // namespace velo_hip
