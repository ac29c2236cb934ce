"""CPU: the pose hand-off's host logic against HAND-COMPUTED matrices -- not against the same code:
agreement = pose_vec2mat(dpose * dT^-1) (main.cpp:416-424) and the rule that skips an edge on poor agreement (main.cpp:426-437,
thresholds kitti.h:33-35).  velo_pose_mat_to_vec / velo_pose_vec_to_mat are host code of the C-ABI library (no GPU needed)."""
import numpy as np

import velo_amd  # noqa: F401
from velo_amd import odometry


def _trans(x, y, z):
    T = np.eye(4)
    T[:3, 3] = (x, y, z)
    return T


def _rot_z(a):
    T = np.eye(4)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    return T


def test_agreement_is_the_pose_between_prediction_and_registration():
    dT = _trans(0.0, 0.0, 1.0)
    assert np.allclose(odometry.agreement_of(dT, dT), 0.0, atol=1e-15)
    # registration 0.3 m to the side of the prediction: (dpose dT^-1) = translate(0.3, 0, 0)
    assert np.allclose(odometry.agreement_of(_trans(0.3, 0.0, 1.0), dT), [0, 0, 0, 0.3, 0, 0], atol=1e-15)
    # registration = Rz(0.06) after the predicted motion: dpose dT^-1 = Rz(0.06), a pure rotation about z
    a = odometry.agreement_of(_rot_z(0.06) @ dT, dT)
    assert np.allclose(a, [0, 0, 0.06, 0, 0, 0], atol=1e-12)
    # registration = predicted motion after Rz(0.06): dpose dT^-1 = dT Rz dT^-1 -- the same rotation, seen from 1 m along z: no lever arm
    # about the z axis itself, but a prediction 2 m along x gives one: R t - t with t = (2, 0, 0)
    dT2 = _trans(2.0, 0.0, 0.0)
    a = odometry.agreement_of(dT2 @ _rot_z(0.06), dT2)
    want_t = np.array([2.0, 0, 0]) - _rot_z(0.06)[:3, :3] @ np.array([2.0, 0, 0])
    assert np.allclose(a[:3], [0, 0, 0.06], atol=1e-12) and np.allclose(a[3:], want_t, atol=1e-12)


def test_edges_are_skipped_like_main_cpp_426_437():
    ag = lambda t, r: np.array([r, 0.0, 0.0, t, 0.0, 0.0])      # noqa: E731
    # dframe == 1: the value is printed, the edge always stays
    assert odometry.edge_is_rejected(ag(5.0, 1.0), 1) is None
    # translation: more than min(0.1 * dframe, 10)
    assert odometry.edge_is_rejected(ag(0.19, 0.0), 2) is None
    assert odometry.edge_is_rejected(ag(0.21, 0.0), 2) == "poor t agreement"
    assert odometry.edge_is_rejected(ag(0.29, 0.0), 3) is None and odometry.edge_is_rejected(ag(0.31, 0.0), 3) == "poor t agreement"
    assert odometry.edge_is_rejected(ag(9.9, 0.0), 200) is None and odometry.edge_is_rejected(ag(10.1, 0.0), 200) == "poor t agreement"
    # rotation: more than 0.05 rad, whatever dframe > 1
    assert odometry.edge_is_rejected(ag(0.0, 0.049), 2) is None and odometry.edge_is_rejected(ag(0.0, 0.051), 5) == "poor r agreement"
    # the translation test comes first (main.cpp:426 before :433)
    assert odometry.edge_is_rejected(ag(1.0, 1.0), 2) == "poor t agreement"
    assert (odometry.AGREEMENT_T_THRESH, odometry.AGREEMENT_R_THRESH, odometry.LOOP_CLOSE_THRESH) == (0.1, 0.05, 10.0)   # kitti.h:33-35


def test_native_handoff_equals_numpy_matrix_arithmetic():
    """velo_pose_handoff (the drive loop's hand-off for n sequences, main.cpp:311-331,408) against plain numpy."""
    from velo_amd import api
    rng = np.random.default_rng(11)
    n = 5
    P = np.stack([api.pose_vec_to_mat(np.concatenate([rng.normal(size=3) * 0.3, rng.normal(size=3) * 20])) for _ in range(n)])
    D = np.stack([api.pose_vec_to_mat(np.concatenate([rng.normal(size=3) * 0.02, [0.02, -0.01, 1.0 + 0.1 * rng.normal()]])) for _ in range(n)])
    P_in = np.ascontiguousarray(P.copy())
    x = api.pose_handoff(P_in, D)
    assert np.allclose(P_in, P @ D, atol=1e-13)                                  # poses advanced in place
    for i in range(n):
        want = api.pose_mat_to_vec(np.linalg.inv(P[i]) @ (P[i] @ D[i]))
        assert np.allclose(x[i], want, atol=1e-12)
        assert np.allclose(x[i], api.pose_mat_to_vec(D[i]), atol=1e-12)          # constant velocity: the pair's own motion
    # the chain of a 3-frame drive by hand: identity, then two translations along z
    P2 = np.ascontiguousarray(np.eye(4)[None].copy())
    x1 = api.pose_handoff(P2, _trans(0, 0, 1.0)[None])
    x2 = api.pose_handoff(P2, _trans(0.1, 0, 1.2)[None])
    assert np.allclose(P2[0], _trans(0.1, 0, 2.2)) and np.allclose(x1[0], [0, 0, 0, 0, 0, 1.0]) and np.allclose(x2[0], [0, 0, 0, 0.1, 0, 1.2])
