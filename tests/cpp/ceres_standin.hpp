// The two-method surface of ceres::CostFunction that include/velo_ceres_cost.hpp derives from -- here ONLY so that OUR adaptor
// can be compiled and its Evaluate contract exercised in an image without Ceres.  No reference source is built against it.
#pragma once
#include <cstdint>
#include <vector>
namespace ceres {
class CostFunction {
public:
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
    int num_residuals() const { return num_residuals_; }
    const std::vector<int32_t>& parameter_block_sizes() const { return sizes_; }
protected:
    void set_num_residuals(int n) { num_residuals_ = n; }
    std::vector<int32_t>* mutable_parameter_block_sizes() { return &sizes_; }
private:
    int num_residuals_ = 0;
    std::vector<int32_t> sizes_;
};
}  // namespace ceres
