// velo_ceres_cost.hpp -- seam 3 of SURVEY.md 8(b): a ceres::CostFunction over the C-ABI, for callers that keep Ceres as the
// minimiser (velo.h:897-902) but want the association and the residual/Jacobian evaluation on the GPU.
//
// One batched cost function replaces the ~100k AutoDiffCostFunction<cost3DPD,1,6> blocks (velo.h:875-892) and the visual
// blocks (velo.h:683-689,710-718,744-752,777-785) of one association round.  Ceres applies a loss per RESIDUAL BLOCK, so a
// single block cannot use Ceres' loss objects: the robustifier is baked into the rows instead (rho'' <= 0 for Cauchy and
// Arctan => Ceres' corrector is the scaling r*sqrt(rho'), J*sqrt(rho') -- SURVEY.md row L1) and the block is added with a NULL
// loss.  The minimiser then sees exactly the least-squares problem Ceres builds internally from the reference's blocks.
//
// The header needs <ceres/ceres.h> only for the base class; this image has no Ceres, so tests/cpp compiles it against a
// two-method stand-in of ceres::CostFunction (tests/cpp/ceres_standin.hpp) to exercise the Evaluate contract:
//   parameters[0] = the 6 pose parameters; residuals[num_residuals]; jacobians may be NULL, jacobians[0] may be NULL,
//   otherwise row-major num_residuals x 6.
#ifndef VELO_CERES_COST_HPP_
#define VELO_CERES_COST_HPP_

#include <vector>

#include "velo_hip.h"

namespace velo_hip {

// Usage per association round (single-threaded like the reference, velo.h:900; a context is not thread-safe):
//   velo_associate(c, transform, iter, &n_valid);  velo_build_visual(c, transform, iter, &n_blocks);
//   problem.AddResidualBlock(new velo_hip::BatchedCost(c), /*loss=*/nullptr, transform);
class BatchedCost : public ceres::CostFunction {
public:
    explicit BatchedCost(velo_ctx* c) : c_(c) {
        int32_t n_rows = 0;
        const double x0[6] = {0, 0, 0, 0, 0, 0};
        velo_evaluate_rows(c_, x0, nullptr, nullptr, 0, &n_rows);          // row count of the current blocks
        set_num_residuals(n_rows);
        mutable_parameter_block_sizes()->push_back(6);
    }
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
        int32_t n = 0;
        double* J = (jacobians && jacobians[0]) ? jacobians[0] : nullptr;
        if (!J) { scratch_.resize(6 * (size_t)num_residuals()); J = scratch_.data(); }
        return velo_evaluate_rows(c_, parameters[0], residuals, J, num_residuals(), &n) == VELO_OK && n == num_residuals();
    }

private:
    velo_ctx* c_;
    mutable std::vector<double> scratch_;
};

}  // namespace velo_hip
#endif  // VELO_CERES_COST_HPP_
