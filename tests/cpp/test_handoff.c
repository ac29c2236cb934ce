/* A plain C99 caller of the C-ABI (include/velo_hip.h must compile as C): the pose hand-off of the drive loop, main.cpp:311-331,408,
 * for two sequences over three frames, checked against values worked out by hand.  No GPU needed: velo_pose_handoff, velo_pose_vec_to_mat
 * and velo_pose_mat_to_vec are host arithmetic.  Also takes the addresses of the entry points a drive uses, so a missing export fails the link. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "velo_hip.h"

static int near(double a, double b, double tol) { return fabs(a - b) <= tol; }

int main(void) {
    double poses[2][16], T[2][16], x[2][6];
    const double step0[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 1.0};             /* main.cpp:170 */
    const double yaw[6] = {0.0, 0.1, 0.0, 0.0, 0.0, 0.0};                /* 0.1 rad about the camera's y axis */
    int i, k, bad = 0;
    /* (typed pointers: the declarations of the header must match the exports) */
    int (*e0)(velo_ctx**, int32_t, const velo_scan_ref*, const velo_scan_ref*, double*, double*, velo_summary*) = &velo_register_batch;
    int (*e1)(velo_ctx**, int32_t, const velo_scan_ref*, const velo_scan_ref*, const velo_match* const*, const int32_t*, double*, double*, velo_summary*) = &velo_register_batch_visual;
    int (*e2)(velo_ctx*) = &velo_source_to_target;
    for (i = 0; i < 2; i++) { memset(poses[i], 0, sizeof(poses[i])); poses[i][0] = poses[i][5] = poses[i][10] = poses[i][15] = 1.0; }
    /* frame 1: both sequences move 1 m along z; sequence 1 also turns */
    if (velo_pose_vec_to_mat(step0, T[0]) != VELO_OK || velo_pose_vec_to_mat(yaw, T[1]) != VELO_OK) return 2;
    T[1][11] = 1.0;
    if (velo_pose_handoff(2, &poses[0][0], &T[0][0], &x[0][0]) != VELO_OK) return 3;
    for (k = 0; k < 6; k++) bad += !near(x[0][k], step0[k], 1e-15);
    bad += !near(x[1][1], 0.1, 1e-12) + !near(x[1][5], 1.0, 1e-12) + !near(x[1][0], 0.0, 1e-12);
    bad += !near(poses[0][11], 1.0, 1e-15) + !near(poses[1][11], 1.0, 1e-15) + !near(poses[1][0], cos(0.1), 1e-15) + !near(poses[1][2], sin(0.1), 1e-15);
    /* frame 2: the same relative motions again: sequence 0 is 2 m along z; sequence 1 has turned 0.2 rad and moved along its own z */
    if (velo_pose_handoff(2, &poses[0][0], &T[0][0], &x[0][0]) != VELO_OK) return 4;
    bad += !near(poses[0][11], 2.0, 1e-15);
    bad += !near(poses[1][0], cos(0.2), 1e-14) + !near(poses[1][3], sin(0.1), 1e-14) + !near(poses[1][11], 1.0 + cos(0.1), 1e-14);
    bad += !near(x[1][1], 0.1, 1e-12) + !near(x[1][5], 1.0, 1e-12);        /* constant velocity: the guess is the pair's own motion */
    /* argument checks come back as status codes */
    bad += velo_pose_handoff(1, NULL, &T[0][0], NULL) == VELO_OK;
    bad += velo_pose_handoff(0, NULL, NULL, NULL) != VELO_OK;
    bad += (e0 == NULL) + (e1 == NULL) + (e2 == NULL);
    printf("handoff %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
