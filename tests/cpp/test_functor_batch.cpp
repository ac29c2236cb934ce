// Exercises include/velo_functor_batch.hpp with stand-ins that have the member names of the reference's functor structs
// (the adaptor is templated on "a struct with these members").  Input: doubles from stdin-free binary file: n, then per
// functor kind + 9 constants, then x[6].  Output: per functor 3 residuals + 18 Jacobian entries as doubles.
#include <cstdio>
#include <vector>

#include "velo_functor_batch.hpp"

namespace refshape {   // shapes only -- no arithmetic lives here
struct P3D3D { double m_x, m_y, m_z, s_x, s_y, s_z; };
struct P3D2D { double m_x, m_y, m_z, s_x, s_y, t_x, t_y, t_z; };
struct P2D2D { double m_x, m_y, s_x, s_y, t_x, t_y, t_z; };
struct P3DPD { double point_x, point_y, point_z, normal_x, normal_y, normal_z, offset_x, offset_y, offset_z; };
}

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    double nd;
    if (fread(&nd, 8, 1, f) != 1) return 2;
    const int n = (int)nd;
    velo_hip::FunctorBatch B;
    for (int i = 0; i < n; i++) {
        double rec[10];
        if (fread(rec, 8, 10, f) != 10) return 2;
        const double* c = rec + 1;
        switch ((int)rec[0]) {
            case 0: { refshape::P3D3D s = {c[0], c[1], c[2], c[3], c[4], c[5]}; B.add(velo_hip::pack_cost3D3D(s)); break; }
            case 1: { refshape::P3D2D s = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]}; B.add(velo_hip::pack_cost3D2D(s)); break; }
            case 2: { refshape::P3D2D s = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]}; B.add(velo_hip::pack_cost2D3D(s)); break; }
            case 3: { refshape::P2D2D s = {c[0], c[1], c[2], c[3], c[4], c[5], c[6]}; B.add(velo_hip::pack_cost2D2D(s)); break; }
            default: { refshape::P3DPD s = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]}; B.add(velo_hip::pack_cost3DPD(s)); break; }
        }
    }
    double x[6];
    if (fread(x, 8, 6, f) != 6) return 2;
    fclose(f);
    velo_ctx* ctx = nullptr;
    if (velo_create(&ctx, 0) != VELO_OK) { fprintf(stderr, "%s\n", velo_last_error()); return 3; }
    if (B.evaluate(ctx, x, false) != VELO_OK || B.jacobian(0) != nullptr) return 4;     // residual-only evaluation
    std::vector<double> r_only(B.residual(0), B.residual(0) + 3 * (size_t)n);
    if (B.evaluate(ctx, x, true) != VELO_OK) { fprintf(stderr, "%s\n", velo_last_error()); return 4; }
    for (size_t i = 0; i < r_only.size(); i++) if (r_only[i] != B.residual(0)[i]) return 5;
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 1;
    int dims = 0;
    for (int i = 0; i < n; i++) { fwrite(B.residual(i), 8, 3, o); fwrite(B.jacobian(i), 8, 18, o); dims += velo_hip::functor_dim(B.at(i)); }
    fclose(o);
    velo_destroy(ctx);
    printf("functors %d rows %d\n", n, dims);
    return 0;
}
