// C++11 caller of velo_hip::PoseChain (include/velo_frame_to_frame.hpp): the host half of the reference's drive loop
// (main.cpp:305-331,407-437) against values worked out by hand.  No GPU needed.
#include <cmath>
#include <cstdio>

#include "standins.hpp"
#include "velo_frame_to_frame.hpp"

static bool near(double a, double b, double tol) { return std::fabs(a - b) <= tol; }

int main() {
    int bad = 0;
    velo_hip::PoseChain chain;
    double x[6];
    chain.predict(x);                                                     // first pair: main.cpp:170
    bad += !(x[0] == 0 && x[1] == 0 && x[2] == 0 && x[3] == 0 && x[4] == 0 && x[5] == 1.0);
    // frame 1: 1.1 m along z, 0.1 m to the side: 0.1 m off the start-up guess in x and in z
    double T1[16] = {1, 0, 0, 0.1, 0, 1, 0, 0, 0, 0, 1, 1.1, 0, 0, 0, 1};
    std::array<double, 6> a1 = chain.push(T1);
    bad += !near(a1[3], 0.1, 1e-15) + !near(a1[5], 0.1, 1e-14) + !near(a1[0], 0.0, 1e-15);
    chain.predict(x);                                                     // constant velocity: the pair's own motion
    bad += !near(x[3], 0.1, 1e-15) + !near(x[5], 1.1, 1e-15);
    // frame 2: the same motion plus 0.06 rad about y: agreement = a pure rotation about y (dpose = R * dT_pred with dT_pred = T1)
    const double c = std::cos(0.06), s = std::sin(0.06);
    double R[16] = {c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1}, T2[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double v = 0; for (int k = 0; k < 4; k++) v += R[4 * i + k] * T1[4 * k + j]; T2[4 * i + j] = v; }
    std::array<double, 6> a2 = chain.push(T2);
    bad += !near(a2[1], 0.06, 1e-12) + !near(a2[0], 0.0, 1e-12) + !near(a2[3], 0.0, 1e-12) + !near(a2[5], 0.0, 1e-12);
    bad += chain.frames() != 3;
    // pose 2 = T1 * T2 (main.cpp:408)
    double P[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double v = 0; for (int k = 0; k < 4; k++) v += T1[4 * i + k] * T2[4 * k + j]; P[4 * i + j] = v; }
    for (int k = 0; k < 16; k++) bad += !near(chain.pose(2)[k], P[k], 1e-14);
    // main.cpp:426-437
    std::array<double, 6> t{{0, 0, 0, 0.21, 0, 0}}, r{{0.051, 0, 0, 0, 0, 0}}, ok{{0.049, 0, 0, 0.19, 0, 0}};
    bad += velo_hip::PoseChain::edge_rejected(t, 1) != nullptr;
    bad += velo_hip::PoseChain::edge_rejected(t, 2) == nullptr || velo_hip::PoseChain::edge_rejected(t, 3) != nullptr;
    bad += velo_hip::PoseChain::edge_rejected(r, 4) == nullptr || velo_hip::PoseChain::edge_rejected(ok, 2) != nullptr;
    std::array<double, 6> far{{0, 0, 0, 10.1, 0, 0}};
    bad += velo_hip::PoseChain::edge_rejected(far, 500) == nullptr;
    std::printf("pose chain %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
