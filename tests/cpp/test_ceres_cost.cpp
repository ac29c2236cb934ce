// Exercises include/velo_ceres_cost.hpp the way ceres::Problem would call it: Evaluate with jacobians, with jacobians == NULL
// and with jacobians[0] == NULL.  Input: the case file of tests/test_cpp_adaptor.py; output: doubles (n_rows, r[n], J[n*6], r2[n]).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ceres_standin.hpp"
#include "velo_ceres_cost.hpp"

static bool read_cloud(FILE* f, std::vector<float>& xyz, std::vector<int32_t>& off) {
    int32_t nr;
    if (fread(&nr, 4, 1, f) != 1) return false;
    off.resize(nr + 1);
    if (fread(off.data(), 4, nr + 1, f) != (size_t)nr + 1) return false;
    xyz.resize(3 * (size_t)off[nr]);
    return fread(xyz.data(), 4, xyz.size(), f) == xyz.size();
}

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<float> src, tgt;
    std::vector<int32_t> soff, toff;
    if (!read_cloud(f, src, soff) || !read_cloud(f, tgt, toff)) return 2;
    int32_t nm, skip;
    if (fread(&nm, 4, 1, f) != 1 || fread(&skip, 4, 1, f) != 1) return 2;
    std::vector<velo_match> m(nm);
    if (nm && fread(m.data(), sizeof(velo_match), nm, f) != (size_t)nm) return 2;
    double x[6];
    if (fread(x, 8, 6, f) != 6) return 2;
    fclose(f);
    velo_ctx* c = nullptr;
    if (velo_create(&c, 0) != VELO_OK) { fprintf(stderr, "%s\n", velo_last_error()); return 3; }
    velo_params P;
    velo_get_params(c, &P); P.icp_skip = skip; velo_set_params(c, &P);
    int32_t nv = 0, nb = 0;
    if (velo_set_target(c, tgt.data(), 12, toff.data(), (int32_t)toff.size() - 1, 0) != VELO_OK ||
        velo_set_source(c, src.data(), 12, soff.data(), (int32_t)soff.size() - 1, 0) != VELO_OK ||
        velo_set_visual(c, nm ? m.data() : nullptr, nm) != VELO_OK ||
        velo_associate(c, x, 1, &nv) != VELO_OK || velo_build_visual(c, x, 1, &nb) != VELO_OK) { fprintf(stderr, "%s\n", velo_last_error()); return 3; }
    velo_hip::BatchedCost cost(c);
    const int n = cost.num_residuals();
    if (cost.parameter_block_sizes().size() != 1 || cost.parameter_block_sizes()[0] != 6 || n <= 0) return 4;
    std::vector<double> r(n), J(6 * (size_t)n), r2(n), r3(n);
    const double* params[1] = {x};
    double* jac[1] = {J.data()};
    if (!cost.Evaluate(params, r.data(), jac)) return 5;
    if (!cost.Evaluate(params, r2.data(), nullptr)) return 5;        // cost-only evaluation
    double* jac_null[1] = {nullptr};
    if (!cost.Evaluate(params, r3.data(), jac_null)) return 5;       // parameter block held constant
    for (int i = 0; i < n; i++) if (r2[i] != r[i] || r3[i] != r[i]) return 6;
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 1;
    const double nd = (double)n;
    fwrite(&nd, 8, 1, o); fwrite(r.data(), 8, n, o); fwrite(J.data(), 8, J.size(), o);
    fclose(o);
    velo_destroy(c);
    printf("rows %d valid %d visual_blocks %d\n", n, nv, nb);
    return 0;
}
