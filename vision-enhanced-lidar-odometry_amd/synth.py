"""Seeded synthetic HDL-64E scans, KITTI .bin layout and the reference's ring segmenter.

Everything here is HOST-side input preparation (numpy) for the scan-matching core; it is the
"synthetic 120k-pt HDL-64E scan pair" BASELINE.json names.

What follows the reference:
  * `.bin` record layout float32 (x, y, z, reflectance)            -- kitti.h:130-148
  * ring split rule "x > 0 and sign(y) flipped"                    -- kitti.h:164-176
  * per-ring reorder  new[i] = old[n-1-((i + n/2) % n)]            -- kitti.h:178-183
  * points stored in the camera-0 frame (velo_to_cam applied)      -- kitti.h:162,180
  * stereo rig: cam_trans[0] = 0, cam_trans[1] = (-0.537, 0, 0)    -- kitti.h:76-78 (KITTI grey pair)

The random numbers come from a counter-based generator written here (splitmix64 + Box-Muller) so
that a (seed, index) pair always yields the same value, on any machine and in any evaluation order.
"""
from __future__ import annotations

import numpy as np

# --- HDL-64E geometry ------------------------------------------------------
N_BEAMS = 64
N_AZIMUTH = 1875
ELEV_TOP_DEG = 2.0
ELEV_BOTTOM_DEG = -24.8
SENSOR_HEIGHT = 1.73

# idealised velodyne -> camera-0 rigid transform (x_c = -y_v, y_c = -z_v, z_c = x_v) + offset,
# the shape of KITTI's Tr (kitti.h:100-107).
VELO_TO_CAM = np.array(
    [[0.0, -1.0, 0.0, 0.0],
     [0.0, 0.0, -1.0, -0.08],
     [1.0, 0.0, 0.0, -0.27],
     [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)

CAM_TRANS = np.array([[0.0, 0.0, 0.0], [-0.537, 0.0, 0.0]], dtype=np.float32)

# KITTI-grey-shaped intrinsics and image size (kitti.h:37-38); the canonical-coordinate window of each camera is computed the
# way kitti.h:85-97 does it: K^-1 * (0,0,1) and K^-1 * (w,h,1) in float, divided by z, widened to double.
IMG_WIDTH, IMG_HEIGHT = 1226, 370
CAM_K = np.array([[718.856, 0.0, 607.1928], [0.0, 718.856, 185.2157], [0.0, 0.0, 1.0]], dtype=np.float32)
DEPTH_ASSOC_THRESH = 0.015                                   # kitti.h:28


def cam_window() -> np.ndarray:
    """(min_x, max_x, min_y, max_y) as doubles, same for both cameras of the idealised rig."""
    kinv = np.linalg.inv(CAM_K.astype(np.float64)).astype(np.float32)
    lo = kinv @ np.array([0.0, 0.0, 1.0], dtype=np.float32)
    hi = kinv @ np.array([IMG_WIDTH, IMG_HEIGHT, 1.0], dtype=np.float32)
    return np.array([np.float32(lo[0] / lo[2]), np.float32(hi[0] / hi[2]), np.float32(lo[1] / lo[2]), np.float32(hi[1] / hi[2])],
                    dtype=np.float64)


def keypoints_in_window(n: int, seed: int = 7, window=None) -> np.ndarray:
    """n synthetic keypoints in canonical coordinates, uniform over the camera window (float32 [n,2])."""
    w = cam_window() if window is None else np.asarray(window, dtype=np.float64)
    u = uniform01(seed, 2 * n, stream=9).reshape(n, 2)
    return np.stack([w[0] + u[:, 0] * (w[1] - w[0]), w[2] + u[:, 1] * (w[3] - w[2])], axis=1).astype(np.float32)


# --- counter-based RNG -----------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n doubles in (0, 1); value i depends only on (seed, stream, i)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64(np.full(1, seed, dtype=np.uint64) * np.uint64(0x632BE59BD9B4E019)
                          + np.uint64(stream) * np.uint64(0xD1342543DE82EF95))[0]
        bits = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + key)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    u1 = uniform01(seed, n, 2 * stream)
    u2 = uniform01(seed, n, 2 * stream + 1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# --- poses -----------------------------------------------------------------------------------
def rotvec_to_matrix(w) -> np.ndarray:
    """Rodrigues, double. Same map as ceres::AngleAxisToRotationMatrix [3P]."""
    w = np.asarray(w, dtype=np.float64)
    th = float(np.sqrt(w @ w))
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)
    if th * th <= np.finfo(np.float64).eps:
        return np.eye(3) + K
    k = K / th
    return np.eye(3) + np.sin(th) * k + (1.0 - np.cos(th)) * (k @ k)


def matrix_to_rotvec(R) -> np.ndarray:
    """Inverse of the above via the quaternion route (ceres::RotationMatrixToAngleAxis [3P])."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr >= 0.0:
        t = np.sqrt(1.0 + tr)
        q0 = 0.5 * t
        t = 0.5 / t
        q = np.array([q0, (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t])
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4)
        q[i + 1] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[j + 1] = (R[j, i] + R[i, j]) * t
        q[k + 1] = (R[k, i] + R[i, k]) * t
    s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
    if s2 > 0.0:
        s = np.sqrt(s2)
        two_theta = 2.0 * (np.arctan2(-s, -q[0]) if q[0] < 0.0 else np.arctan2(s, q[0]))
        return q[1:] * (two_theta / s)
    return q[1:] * 2.0


def pose_matrix(yaw: float, pitch: float, roll: float, t) -> np.ndarray:
    """4x4 sensor pose in the velodyne convention (x fwd, y left, z up); R = Rz(yaw) Ry(pitch) Rx(roll)."""
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = np.asarray(t, dtype=np.float64)
    return T


def velo_pose_to_cam_x(T_velo: np.ndarray) -> np.ndarray:
    """Relative velodyne-frame pose (cur -> prev) expressed as the solver's x = (omega, t) in camera-0 frame.

    p_prev_cam = C p_prev_velo = C T_velo C^-1 p_cur_cam   (C = VELO_TO_CAM); cf. velo.h:808-810.
    """
    C = VELO_TO_CAM.astype(np.float64)
    T = C @ T_velo @ np.linalg.inv(C)
    x = np.zeros(6)
    x[:3] = matrix_to_rotvec(T[:3, :3])
    x[3:] = T[:3, 3]
    return x


# --- scene + ray casting -----------------------------------------------------------------------
class Scene:
    """Closed street canyon: ground, two side walls, two end walls and seeded car-sized boxes."""

    def __init__(self, seed: int = 0, n_boxes: int = 24, half_width: float = 8.0, half_length: float = 45.0, box_horizon=None):
        self.ground_z = -SENSOR_HEIGHT
        self.box_horizon = box_horizon
        self.half_width = half_width
        self.half_length = half_length
        u = uniform01(seed, 4 * n_boxes, stream=7).reshape(n_boxes, 4)
        cx = -half_length + 4.0 + u[:, 0] * (2 * half_length - 8.0)
        side = np.where(u[:, 1] < 0.5, -1.0, 1.0)
        cy = side * (2.6 + u[:, 2] * (half_width - 4.0))      # keep the driving lane free
        yaw_small = (u[:, 3] - 0.5) * 0.0                     # axis-aligned boxes (exact slab test)
        del yaw_small
        sx, sy, sz = 4.0, 1.8, 1.5
        self.box_min = np.stack([cx - sx / 2, cy - sy / 2, np.full(n_boxes, self.ground_z)], axis=1)
        self.box_max = np.stack([cx + sx / 2, cy + sy / 2, np.full(n_boxes, self.ground_z + sz)], axis=1)

    def cast(self, origin: np.ndarray, dirs: np.ndarray) -> np.ndarray:
        """Range along each unit direction to the first surface (double)."""
        o = np.asarray(origin, dtype=np.float64)
        d = np.asarray(dirs, dtype=np.float64)
        big = 1e30
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
        t_best = np.full(d.shape[0], big)

        def plane(axis: int, value: float):
            t = (value - o[axis]) * inv[:, axis]
            return np.where(np.isfinite(t) & (t > 1e-6), t, big)

        t_best = np.minimum(t_best, plane(2, self.ground_z))
        t_best = np.minimum(t_best, plane(1, self.half_width))
        t_best = np.minimum(t_best, plane(1, -self.half_width))
        t_best = np.minimum(t_best, plane(0, self.half_length))
        t_best = np.minimum(t_best, plane(0, -self.half_length))
        for bmin, bmax in zip(self.box_min, self.box_max):
            if self.box_horizon is not None and (bmin[0] - o[0] > self.box_horizon or o[0] - bmax[0] > self.box_horizon):
                continue                                      # long roads (drives): boxes beyond the horizon are not drawn
            t0 = (bmin - o) * inv
            t1 = (bmax - o) * inv
            tn = np.nanmax(np.minimum(t0, t1), axis=1)
            tf = np.nanmin(np.maximum(t0, t1), axis=1)
            hit = (tn <= tf) & (tn > 1e-6)
            t_best = np.where(hit & (tn < t_best), tn, t_best)
        return t_best


def beam_directions() -> np.ndarray:
    """(N_BEAMS*N_AZIMUTH, 3) unit vectors, ring-major (ring 0 = top beam), azimuth from +x toward +y."""
    elev = np.deg2rad(np.linspace(ELEV_TOP_DEG, ELEV_BOTTOM_DEG, N_BEAMS))
    az = (np.arange(N_AZIMUTH) + 0.5) * (2.0 * np.pi / N_AZIMUTH)
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (N_BEAMS, N_AZIMUTH))], axis=2)
    return d.reshape(-1, 3)


def hdl64_scan(scene: Scene, T_world_sensor: np.ndarray, noise_seed: int, sigma: float = 0.02,
               n_beams: int = N_BEAMS, n_azimuth: int = N_AZIMUTH) -> np.ndarray:
    """One sweep in the SENSOR (velodyne) frame, float32 (n, 3), ring-major file order like a KITTI .bin."""
    if n_beams == N_BEAMS and n_azimuth == N_AZIMUTH:
        d_s = beam_directions()
    else:
        elev = np.deg2rad(np.linspace(ELEV_TOP_DEG, ELEV_BOTTOM_DEG, n_beams))
        az = (np.arange(n_azimuth) + 0.5) * (2.0 * np.pi / n_azimuth)
        ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
        d_s = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :],
                        np.broadcast_to(se, (n_beams, n_azimuth))], axis=2).reshape(-1, 3)
    R = T_world_sensor[:3, :3]
    o = T_world_sensor[:3, 3]
    d_w = d_s @ R.T
    rng = scene.cast(o, d_w)
    rng = rng + sigma * normal01(noise_seed, rng.shape[0], stream=1)
    return (d_s * rng[:, None]).astype(np.float32)


# --- KITTI layout + reference ring segmenter ------------------------------------------------
def write_kitti_bin(path: str, pts_velo: np.ndarray) -> None:
    rec = np.zeros((pts_velo.shape[0], 4), dtype=np.float32)
    rec[:, :3] = pts_velo
    rec.tofile(path)


def read_kitti_bin(path: str) -> np.ndarray:
    """kitti.h:121-152 -- float32 quadruples, reflectance dropped."""
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)[:, :3].copy()


def segment_points(pts_velo: np.ndarray, velo_to_cam: np.ndarray = VELO_TO_CAM):
    """Ring split + reorder exactly as kitti.h:154-185.

    Returns (xyz_cam float32 (n,3) ring-major, ring_offsets int32 (Rs+1,)).
    The camera-frame copy is produced with float32 arithmetic like pcl::transformPointCloud<float>.
    """
    p = np.asarray(pts_velo, dtype=np.float32)
    n = p.shape[0]
    M = velo_to_cam.astype(np.float32)
    cam = (p[:, 0:1] * M[:3, 0][None, :] + p[:, 1:2] * M[:3, 1][None, :]
           + p[:, 2:3] * M[:3, 2][None, :] + M[:3, 3][None, :]).astype(np.float32)
    prev_y = np.concatenate([[np.float32(0)], p[:-1, 1]])
    brk = (np.arange(n) > 0) & (p[:, 0] > 0) & ((p[:, 1] > 0) != (prev_y > 0))
    ring_id = np.cumsum(brk.astype(np.int64))
    n_rings = int(ring_id[-1]) + 1 if n else 0
    counts = np.bincount(ring_id, minlength=n_rings)
    offsets = np.zeros(n_rings + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(counts)
    out = np.empty_like(cam)
    for s in range(n_rings):
        m = int(counts[s])
        i = np.arange(m)
        src = m - 1 - ((i + m // 2) % m)
        out[offsets[s]:offsets[s + 1]] = cam[offsets[s] + src]
    return out, offsets.astype(np.int32)


# --- the BASELINE.json workloads ---------------------------------------------------------------
TRUE_MOTION = dict(yaw=0.02, pitch=0.002, roll=0.002, t=(1.0, 0.02, 0.01))
INITIAL_GUESS = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.0])     # main.cpp:170


def scan_pair(n_beams: int = N_BEAMS, n_azimuth: int = N_AZIMUTH, scene_seed: int = 0, sigma: float = 0.02,
              motion=None, noise_seeds=(1, 2), start=(0.0, 0.0, 0.0)):
    """(source=current frame, target=previous frame) ring clouds + true x.

    motion: dict(yaw, pitch, roll, t) of the relative pose (default TRUE_MOTION); noise_seeds: range-noise seeds of the
    two sweeps; start: sensor position of the previous frame in the scene.  The defaults are the canonical pair of BASELINE configs[1].
    Returns dict(src_xyz, src_off, tgt_xyz, tgt_off, x_true, x0).
    """
    scene = Scene(scene_seed)
    T_prev = pose_matrix(0.0, 0.0, 0.0, tuple(start))
    T_rel = pose_matrix(**(motion or TRUE_MOTION))
    T_cur = T_prev @ T_rel
    a = hdl64_scan(scene, T_prev, noise_seed=noise_seeds[0], sigma=sigma, n_beams=n_beams, n_azimuth=n_azimuth)
    b = hdl64_scan(scene, T_cur, noise_seed=noise_seeds[1], sigma=sigma, n_beams=n_beams, n_azimuth=n_azimuth)
    tgt_xyz, tgt_off = segment_points(a)
    src_xyz, src_off = segment_points(b)
    return dict(src_xyz=src_xyz, src_off=src_off, tgt_xyz=tgt_xyz, tgt_off=tgt_off,
                x_true=velo_pose_to_cam_x(T_rel), x0=INITIAL_GUESS.copy())


def distinct_pairs(n: int, n_beams: int = N_BEAMS, n_azimuth: int = N_AZIMUTH, sigma: float = 0.02):
    """n DIFFERENT scan pairs for a batch (bench.py: the pairs in flight must not be copies of one pair): pair 0 is the canonical
    scan_pair(); pair k > 0 has its own scene (box layout), noise, place on the road and motion -- speeds 0.6 .. 1.4 m per frame, yaw
    -0.03 .. 0.04 rad, small pitch / roll / lateral drift, all from a counter-based RNG.  Its initial guess is what the reference's
    drive loop hands frameToFrame (main.cpp:311-331): the PREVIOUS frame's motion, i.e. the true motion up to one frame's
    acceleration (here up to +-0.1 m along the road, +-0.01 rad of yaw); pair 0 keeps the reference's start-up guess (main.cpp:170)."""
    out = [scan_pair(n_beams, n_azimuth, sigma=sigma)]
    for k in range(1, n):
        u = uniform01(900 + k, 12, stream=3)
        motion = dict(yaw=-0.03 + 0.07 * u[0], pitch=0.004 * (u[1] - 0.5), roll=0.004 * (u[2] - 0.5),
                      t=(0.6 + 0.8 * u[3], 0.06 * (u[4] - 0.5), 0.02 * (u[5] - 0.5)))
        d = scan_pair(n_beams, n_azimuth, scene_seed=k, sigma=sigma, motion=motion, noise_seeds=(1 + 10 * k, 2 + 10 * k),
                      start=(-20.0 + 40.0 * u[6], 2.0 * (u[7] - 0.5), 0.0))
        prev = dict(yaw=motion["yaw"] + 0.02 * (u[8] - 0.5), pitch=motion["pitch"] + 0.002 * (u[9] - 0.5), roll=motion["roll"],
                    t=(motion["t"][0] + 0.2 * (u[10] - 0.5), motion["t"][1] + 0.02 * (u[11] - 0.5), motion["t"][2]))
        d["x0"] = velo_pose_to_cam_x(pose_matrix(**prev))
        out.append(d)
    return out


def drive_plan(n_frames: int, seed: int = 0):
    """The trajectory of one DRIVE (bench.py's default workload): scene parameters + the sensor pose of every frame.  The motion changes
    from frame to frame like a car's: speed 0.6 .. 1.4 m per frame with up to +-0.1 m of acceleration per frame, a smooth yaw rate of up to
    +-0.03 rad per frame (the car follows a gently winding lane line inside the free lane), small pitch / roll / vertical jitter -- so the constant-velocity prediction the reference's loop hands frameToFrame (main.cpp:311-331) is off by one frame's
    acceleration.  The road is as long as the drive needs (90 m for up to ~55 frames; boxes farther than 80 m along it are not drawn)."""
    u = uniform01(7000 + seed, 8 * (n_frames + 1), stream=5).reshape(n_frames + 1, 8)
    half_length = max(45.0, 0.5 * 1.4 * n_frames + 15.0)
    scene_kw = dict(seed=seed, n_boxes=int(round(24 * half_length / 45.0)), half_length=half_length, box_horizon=80.0 if half_length > 45.0 else None)
    # the lane line: two sinusoids of the travelled distance (|y| <= 1.2 m: the boxes start 1.7 m from the centre line); the heading follows it
    A1, L1, P1 = 0.3 + 0.3 * u[0, 3], 35.0 + 15.0 * u[0, 4], 2.0 * np.pi * u[0, 5]
    A2, L2, P2 = 0.2 + 0.4 * u[0, 6], 80.0 + 60.0 * u[0, 7], 2.0 * np.pi * u[0, 1]

    def lane(s):
        a1, a2 = 2.0 * np.pi * s / L1 + P1, 2.0 * np.pi * s / L2 + P2
        return A1 * np.sin(a1) + A2 * np.sin(a2), A1 * (2.0 * np.pi / L1) * np.cos(a1) + A2 * (2.0 * np.pi / L2) * np.cos(a2)

    x_start = -half_length + 12.0 + 6.0 * u[0, 0]
    v, dist = 0.6 + 0.8 * u[0, 2], 0.0
    poses = []
    for k in range(n_frames):
        if k > 0:
            v = float(np.clip(v + 0.2 * (u[k, 0] - 0.5), 0.6, 1.4))
            dist += v
        y, slope = lane(dist)
        poses.append(pose_matrix(float(np.arctan(slope)), 0.004 * (u[k, 2] - 0.5), 0.004 * (u[k, 3] - 0.5),
                                 (x_start + dist, float(y), 0.02 * (u[k, 5] - 0.5))))
    x_true = [velo_pose_to_cam_x(np.linalg.inv(poses[k]) @ poses[k + 1]) for k in range(n_frames - 1)]
    return dict(seed=seed, scene_kw=scene_kw, poses_velo=poses, x_true=x_true)


def drive_frame(plan, k: int, n_beams: int = N_BEAMS, n_azimuth: int = N_AZIMUTH, sigma: float = 0.02):
    """Frame k of a planned drive as camera-frame rings (xyz, ring_offsets) -- frames are independent of each other (own noise seed),
    so a pool may produce them in any order."""
    scene = Scene(**plan["scene_kw"])
    pts = hdl64_scan(scene, plan["poses_velo"][k], noise_seed=100_000 * (plan["seed"] + 1) + k, sigma=sigma, n_beams=n_beams, n_azimuth=n_azimuth)
    return segment_points(pts)


def drive(n_frames: int, seed: int = 0, n_beams: int = N_BEAMS, n_azimuth: int = N_AZIMUTH, sigma: float = 0.02):
    """drive_plan + every frame: dict(frames=[(xyz, ring_offsets)] * n_frames, x_true=[relative pose of frame k+1 -> k as the solver's
    6-vector] * (n_frames - 1), poses_velo).  Frame k+1 is registered against frame k, every pair once."""
    plan = drive_plan(n_frames, seed)
    plan["frames"] = [drive_frame(plan, k, n_beams, n_azimuth, sigma) for k in range(n_frames)]
    return plan


def scan_to_map(n_target: int = 2_000_000, scene_seed: int = 0, sigma: float = 0.02,
                n_beams: int = N_BEAMS, n_azimuth: int = N_AZIMUTH, n_queries: int = 1):
    """Config 4/5: accumulated map (scans every 1 m along x, all in the newest map pose's frame) as target.
    n_queries > 1: the returned dict also carries "queries", n_queries DIFFERENT scans to register against the map (query 0 is the
    canonical one of the top-level keys; the others are taken 0.6 .. 1.4 m further along the road with their own yaw and noise; their
    initial guess is the previous frame's motion, like distinct_pairs)."""
    scene = Scene(scene_seed)
    per_scan = n_beams * n_azimuth
    n_scans = -(-n_target // per_scan)
    x_start = -8.0
    poses = [pose_matrix(0.0, 0.0, 0.0, (x_start + k, 0.0, 0.0)) for k in range(n_scans)]
    T_ref = poses[-1]
    C = VELO_TO_CAM.astype(np.float64)
    clouds, offs = [], [np.zeros(1, dtype=np.int64)]
    total = 0
    for k, T in enumerate(poses):
        pts = hdl64_scan(scene, T, noise_seed=100 + k, sigma=sigma, n_beams=n_beams, n_azimuth=n_azimuth)
        xyz, off = segment_points(pts)                       # camera frame of scan k
        Tk = C @ np.linalg.inv(T_ref) @ T @ np.linalg.inv(C)  # scan-k camera frame -> reference camera frame
        xyz = (xyz.astype(np.float64) @ Tk[:3, :3].T + Tk[:3, 3]).astype(np.float32)
        clouds.append(xyz)
        offs.append(off[1:].astype(np.int64) + total)
        total += xyz.shape[0]
    tgt_xyz = np.concatenate(clouds)[:n_target]
    all_off = np.concatenate(offs)
    keep = all_off[all_off < n_target]
    tgt_off = np.concatenate([keep, [n_target]]).astype(np.int32)
    T_rel = pose_matrix(**TRUE_MOTION)
    T_cur = T_ref @ T_rel
    b = hdl64_scan(scene, T_cur, noise_seed=2, sigma=sigma, n_beams=n_beams, n_azimuth=n_azimuth)
    src_xyz, src_off = segment_points(b)
    out = dict(src_xyz=src_xyz, src_off=src_off, tgt_xyz=tgt_xyz, tgt_off=tgt_off,
               x_true=velo_pose_to_cam_x(T_rel), x0=INITIAL_GUESS.copy())
    if n_queries > 1:
        qs = [dict(src_xyz=src_xyz, src_off=src_off, x_true=out["x_true"], x0=out["x0"])]
        for k in range(1, n_queries):
            u = uniform01(700 + k, 8, stream=4)
            motion = dict(yaw=-0.03 + 0.07 * u[0], pitch=0.004 * (u[1] - 0.5), roll=0.004 * (u[2] - 0.5),
                          t=(0.6 + 0.8 * u[3], 0.06 * (u[4] - 0.5), 0.02 * (u[5] - 0.5)))
            Tq = pose_matrix(**motion)
            pts = hdl64_scan(scene, T_ref @ Tq, noise_seed=2 + 10 * k, sigma=sigma, n_beams=n_beams, n_azimuth=n_azimuth)
            sx, so = segment_points(pts)
            prev = dict(motion, yaw=motion["yaw"] + 0.02 * (u[6] - 0.5), t=(motion["t"][0] + 0.2 * (u[7] - 0.5), motion["t"][1], motion["t"][2]))
            qs.append(dict(src_xyz=sx, src_off=so, x_true=velo_pose_to_cam_x(Tq), x0=velo_pose_to_cam_x(pose_matrix(**prev))))
        out["queries"] = qs
    return out


def stereo_matches(n_per_cam: int = 1000, seed: int = 3, x_true=None, outlier_frac: float = 0.10,
                   sigma: float = 7e-4, mix: str = "reproj"):
    """Config 3: visual match records for both cameras.

    Each record is what velo.h:627-654 gathers for one match before it picks residual types:
      d1/d2 depth flags, 3-D point in frame1 (current) / frame2 (previous), canonical 2-D obs in both.
    mix="reproj": half the matches have only d1 (-> cost3D2D), half only d2 (-> cost2D3D).
    mix="all"   : additionally both-depth (3D3D+3D2D+2D3D) and no-depth (2D2D) matches.
    """
    if x_true is None:
        x_true = velo_pose_to_cam_x(pose_matrix(**TRUE_MOTION))
    R = rotvec_to_matrix(x_true[:3])
    t = np.asarray(x_true[3:], dtype=np.float64)
    n = 2 * n_per_cam
    u = uniform01(seed, 3 * n, stream=0).reshape(n, 3)
    z = 4.0 + 36.0 * u[:, 2]
    # current-frame 3-D point in camera-0 coordinates, inside a ~90x35 degree frustum
    P1 = np.stack([(u[:, 0] - 0.5) * 1.6 * z, (u[:, 1] - 0.5) * 0.5 * z, z], axis=1)
    P2 = P1 @ R.T + t                                       # same point in the previous frame
    cam = np.repeat(np.arange(2), n_per_cam)
    tc = CAM_TRANS.astype(np.float64)[cam]
    g = normal01(seed, 4 * n, stream=1).reshape(n, 4) * sigma
    q1 = (P1 + tc)
    q2 = (P2 + tc)
    p2_1 = q1[:, :2] / q1[:, 2:3] + g[:, 0:2]
    p2_2 = q2[:, :2] / q2[:, 2:3] + g[:, 2:4]
    out = uniform01(seed, n, stream=5) < outlier_frac
    p2_2 = np.where(out[:, None], p2_2 + 0.05 * (uniform01(seed, 2 * n, stream=6).reshape(n, 2) - 0.5), p2_2)
    kind = np.arange(n) % (2 if mix == "reproj" else 4)
    d1 = (kind == 0) | (kind == 2)
    d2 = (kind == 1) | (kind == 2)
    dn = normal01(seed, 6 * n, stream=2).reshape(n, 6) * 0.01   # lidar-depth noise on the 3-D points
    rec = dict(
        cam=cam.astype(np.int32),
        point1=np.arange(n, dtype=np.int32), point2=np.arange(n, dtype=np.int32),
        d1=d1.astype(np.uint8), d2=d2.astype(np.uint8),
        p3_1=(P1 + dn[:, :3]).astype(np.float32), p3_2=(P2 + dn[:, 3:]).astype(np.float32),
        p2_1=p2_1.astype(np.float32), p2_2=p2_2.astype(np.float32),
        t_cam=CAM_TRANS[cam].astype(np.float32),
    )
    return rec


def velodyne_sequence(n_frames: int, scene_seed: int = 0, sigma: float = 0.02, n_beams: int = N_BEAMS, n_azimuth: int = N_AZIMUTH,
                      step=None):
    """Raw Velodyne-frame sweeps (n,4 float32 records, reflectance 0) along a gently curving drive + the true sensor poses
    expressed like the reference's pose chain (camera-0 frame, pose 0 = identity)."""
    scene = Scene(scene_seed)
    step = step or dict(yaw=0.01, pitch=0.0005, roll=0.0005, t=(0.8, 0.01, 0.0))
    T = pose_matrix(0.0, 0.0, 0.0, (-20.0, 0.0, 0.0))
    T0 = T.copy()
    C = VELO_TO_CAM.astype(np.float64)
    frames, poses = [], []
    for k in range(n_frames):
        pts = hdl64_scan(scene, T, noise_seed=1000 + k, sigma=sigma, n_beams=n_beams, n_azimuth=n_azimuth)
        rec = np.zeros((pts.shape[0], 4), dtype=np.float32)
        rec[:, :3] = pts
        frames.append(rec)
        poses.append(C @ np.linalg.inv(T0) @ T @ np.linalg.inv(C))
        T = T @ pose_matrix(**step)
    return frames, poses


def triangulation_problem(n_landmarks: int = 2000, n_frames: int = 12, seed: int = 13, sigma_2d: float = 7e-4, sigma_3d: float = 0.03,
                          outlier_frac: float = 0.05):
    """Synthetic input of the landmark triangulation step (main.cpp:640-671): a short camera trajectory, landmarks seen from
    3..n_frames (frame, camera) pairs as canonical 2-D observations, some of them also with a LiDAR-derived 3-D observation.
    Observations are listed like the reference adds its residual blocks: 3-D first (camera-major, frame ascending), then 2-D.
    Returns dict(camera_poses [F,6], cam_trans [2,3], obs (structured: kind, frame, cam, s[3]), obs_offsets [L+1],
    points0 [L,3] f32 (initial guesses), initial_guess [L] u8, truth [L,3])."""
    F = n_frames
    k = np.arange(F, dtype=np.float64)
    poses = np.stack([0.002 * np.sin(0.7 * k), 0.015 * k, 0.001 * np.cos(0.5 * k), 0.02 * k, 0.01 * np.sin(k), 0.9 * k], axis=1)
    poses[0] = 0.0                                            # the first pose is the identity (theta == 0 branch)
    R = [rotvec_to_matrix(p[:3]) for p in poses]
    u = uniform01(seed, 8 * n_landmarks, stream=0).reshape(n_landmarks, 8)
    z = 6.0 + 40.0 * u[:, 2]
    mid = poses[F // 2, 3:]
    truth = np.stack([(u[:, 0] - 0.5) * 1.2 * z, (u[:, 1] - 0.5) * 0.4 * z, z], axis=1) + mid
    n_seen = 3 + np.floor(u[:, 3] * (F - 2)).astype(int)      # frames that see the landmark
    start = np.floor(u[:, 4] * (F - n_seen + 1)).astype(int)
    kinds, frames, cams, svals, off = [], [], [], [], [0]
    g = normal01(seed, 10 * n_landmarks * F, stream=1).reshape(n_landmarks, F, 10)
    ou = uniform01(seed, 2 * n_landmarks * F, stream=2).reshape(n_landmarks, F, 2)
    tc = CAM_TRANS.astype(np.float64)
    for l in range(n_landmarks):
        o3, o2 = [], []
        if l % 50 == 7:                                       # a landmark nobody observed (empty problem)
            off.append(off[-1])
            continue
        for cam in range(2):
            for f in range(start[l], start[l] + n_seen[l]):
                M = R[f].T @ (truth[l] - poses[f, 3:])
                if M[2] < 1.0:
                    continue
                if cam == 0 and l % 3 != 1 and ou[l, f, 0] < 0.4:          # a third of the landmarks never get LiDAR depth
                    o3.append((0, f, cam, *(M + sigma_3d * g[l, f, 0:3])))
                Mc = M + tc[cam]
                q = Mc[:2] / Mc[2] + sigma_2d * g[l, f, 3 + 2 * cam:5 + 2 * cam]
                if ou[l, f, 1] < outlier_frac:
                    q = q + 0.05 * g[l, f, 7:9]
                if l % 11 != 4:                                               # some landmarks have 3-D observations only
                    o2.append((1, f, cam, q[0], q[1], 0.0))
        for rec in o3 + o2:
            kinds.append(rec[0]); frames.append(rec[1]); cams.append(rec[2]); svals.append(rec[3:6])
        off.append(off[-1] + len(o3) + len(o2))
    obs = np.zeros(len(kinds), dtype=[("kind", np.int32), ("frame", np.int32), ("cam", np.int32), ("s", np.float32, 3)])
    obs["kind"], obs["frame"], obs["cam"] = kinds, frames, cams
    obs["s"] = np.asarray(svals, dtype=np.float32).reshape(-1, 3)
    init = (u[:, 5] < 0.5).astype(np.uint8)
    points0 = (truth + 0.5 * normal01(seed, 3 * n_landmarks, stream=3).reshape(n_landmarks, 3)).astype(np.float32)
    return dict(camera_poses=poses, cam_trans=CAM_TRANS.copy(), obs=obs, obs_offsets=np.asarray(off, dtype=np.int32),
                points0=points0, initial_guess=init, truth=truth)
