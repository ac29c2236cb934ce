"""Pose hand-off around the path ("next" row 2 of SURVEY.md 8(f)): host logic only, mirrors main.cpp:305-331,407-437 and
kitti.h:202-216.  It makes a standalone LiDAR odometry loop runnable on (synthetic) sequences:

    for every frame k >= 1:   dT   = T[k-2]^-1 T[k-1]            constant-velocity prediction        main.cpp:311-320
                              x0   = pose_vec2mat(dT)            (a 6-vector despite the name)         main.cpp:331, utility.h:83-96
                              dpose = frameToFrame(scan k -> scan k-1, x0)                              main.cpp:388-405
                              T[k] = T[k-1] * dpose                                                     main.cpp:408
                              agreement = pose_vec2mat(dpose * dT^-1)                                   main.cpp:416-424
The first frame pair starts from the reference's default transform {0,0,0,0,0,1} (main.cpp:170).
"""
from __future__ import annotations

import numpy as np

from . import api

AGREEMENT_T_THRESH = 0.1      # kitti.h:33
AGREEMENT_R_THRESH = 0.05     # kitti.h:34
LOOP_CLOSE_THRESH = 10.0      # kitti.h:35


def agreement_of(dpose: np.ndarray, dT: np.ndarray) -> np.ndarray:
    """main.cpp:416-417: pose_vec2mat(dpose * dT^-1) -- how far the registration moved away from its prediction, as a 6-vector."""
    return api.pose_mat_to_vec(np.asarray(dpose) @ np.linalg.inv(np.asarray(dT)))


def edge_is_rejected(agreement: np.ndarray, dframe: int):
    """main.cpp:426-437 for the odometry pass (ba == 0): an edge over dframe > 1 frames is skipped (`continue`: it never reaches the
    pose graph) when its translation disagrees with the prediction by more than min(agreement_t_thresh * dframe, loop_close_thresh)
    or its rotation by more than agreement_r_thresh.  dframe == 1 edges are never skipped (the value is only printed).
    Returns None or the reason."""
    if dframe <= 1:
        return None
    t, r = float(np.linalg.norm(agreement[3:])), float(np.linalg.norm(agreement[:3]))
    if t > min(AGREEMENT_T_THRESH * dframe, LOOP_CLOSE_THRESH):
        return "poor t agreement"
    if r > AGREEMENT_R_THRESH:
        return "poor r agreement"
    return None


FIRST_GUESS = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.0])     # main.cpp:170


def kitti_pose_line(T: np.ndarray) -> str:
    """kitti.h:202-216: the 3x4 of a pose, row-major, operator<< default formatting (6 significant digits)."""
    return " ".join(f"{float(v):.6g}" for v in np.asarray(T)[:3, :4].reshape(-1)) + " "


class LidarOdometer:
    """Frame-to-frame LiDAR odometry on the device.  Every scan is uploaded and segmented ONCE, as the source of its own
    registration; for the next frame it is promoted to target on the device (`velo_source_to_target`, the role the reference's
    ScansLRU cache plays for sd_prev, main.cpp:233,380) -- only its search index is built then."""

    def __init__(self, device: int = 0, velo_to_cam=None, ndiagonal: int = 1, cache_capacity: int = 50, **params):
        """ndiagonal > 1: every new frame is also registered against frames k-2 .. k-ndiagonal (the reference's dframes[0] with
        ENABLE_ISAM, main.cpp:148-152,306-350); those older scans come out of a device-resident ScanCache (lru.h:31-61)."""
        from . import synth
        self.ctx = api.Context(device, **params)
        self.ndiagonal = int(ndiagonal)
        self.cache = api.ScanCache(device, max(cache_capacity, self.ndiagonal + 1)) if self.ndiagonal > 1 else None   # the window must fit
        self.edges = []          # (frame_from, frame_to, 4x4 relative pose) of the extra registrations that passed the agreement check
        self.rejected_edges = [] # (frame_from, frame_to, 4x4 relative pose, agreement 6-vector, reason): skipped like main.cpp:426-437
        self.velo_to_cam = np.asarray(velo_to_cam if velo_to_cam is not None else synth.VELO_TO_CAM, dtype=np.float32)
        self.poses = []          # ceres_poses_mat (main.cpp:179), camera-0 frame
        self.prev_records = None
        self.agreements = []
        self.summaries = []

    def push(self, records) -> np.ndarray:
        """records: raw Velodyne (n,4) float32 of the new frame, file order.  Returns the new 4x4 pose."""
        if self.prev_records is None:
            self.poses.append(np.eye(4))
            self.prev_records = records
            self.ctx.set_scan_velodyne(False, records, self.velo_to_cam)       # frame 0 waits on the device as "source"
            return self.poses[-1]
        k = len(self.poses)
        if k > 1:
            dT = np.linalg.inv(self.poses[k - 2]) @ self.poses[k - 1]         # main.cpp:315-317
            x0 = api.pose_mat_to_vec(dT)                                       # main.cpp:331
        else:
            x0 = FIRST_GUESS.copy()
            dT = api.pose_vec_to_mat(x0)                                       # main.cpp:319
        self.ctx.source_to_target()                                            # target = previous frame (sd_prev), already on the device
        self.ctx.set_scan_velodyne(False, records, self.velo_to_cam)           # source = current frame (sd)
        x, dpose, s = self.ctx.frame_to_frame(x0)
        self.poses.append(self.poses[k - 1] @ dpose)                           # main.cpp:408
        self.agreements.append(agreement_of(dpose, dT))                        # main.cpp:417 (dframe == 1: printed, never acted on)
        self.summaries.append(s)
        self.prev_records = records
        if self.cache is not None:
            self.cache.store(k - 1, self.ctx, True)                            # frame k-1 with the index just built for it
            for d in range(2, self.ndiagonal + 1):                             # main.cpp:306-307
                if k - d < 0:
                    break
                dT = np.linalg.inv(self.poses[k - d]) @ self.poses[k]          # main.cpp:324-326
                self.cache.load(k - d, self.ctx, True)                         # sd_prev = lru.get(dataset, frame - dframe), main.cpp:350
                _, dpose_d, _ = self._register_edge(k - d, k, api.pose_mat_to_vec(dT))
                ag = agreement_of(dpose_d, dT)                                 # main.cpp:416-424
                why = edge_is_rejected(ag, d)                                  # main.cpp:426-437
                if why is None:
                    self.edges.append((k - d, k, dpose_d))
                else:
                    self.rejected_edges.append((k - d, k, dpose_d, ag, why))
            # the context still holds frame k as its source, which is all the next push needs
        return self.poses[-1]

    def _register_edge(self, frame_from: int, frame_to: int, x0):
        """One extra registration of the current source against the loaded older target (a seam for tests that inject a bad edge)."""
        return self.ctx.frame_to_frame(x0)

    def write_kitti(self, path: str):
        with open(path, "w") as f:                                             # main.cpp:758-763
            for T in self.poses:
                f.write(kitti_pose_line(T) + "\n")

    def close(self):
        if self.cache is not None:
            self.cache.close()
        self.ctx.close()


def read_velo_to_cam(calib_path: str) -> np.ndarray:
    """The `Tr:` row of a KITTI odometry calib.txt (3x4, row-major) as a 4x4 -- what loadCalibration keeps as velo_to_cam
    (kitti.h:53-119)."""
    with open(calib_path) as f:
        for line in f:
            if line.startswith("Tr"):
                v = [float(t) for t in line.split()[1:13]]
                M = np.eye(4, dtype=np.float32)
                M[:3, :4] = np.asarray(v, dtype=np.float32).reshape(3, 4)
                return M
    raise ValueError(f"no Tr row in {calib_path}")


def run_directory(velodyne_dir: str, out_path: str, calib_path: str = None, device: int = 0, max_frames: int = None, **params):
    """LiDAR-only odometry over a directory of KITTI `.bin` sweeps (velodyne/000000.bin ...; kitti.h:121-152): poses in the
    reference's output format (main.cpp:758-763).  Returns the odometer (poses, agreements, summaries)."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(velodyne_dir, "*.bin")))
    if max_frames is not None:
        files = files[:max_frames]
    if not files:
        raise FileNotFoundError(f"no .bin sweeps in {velodyne_dir}")
    odo = LidarOdometer(device, velo_to_cam=read_velo_to_cam(calib_path) if calib_path else None, **params)
    for path in files:
        odo.push(np.fromfile(path, dtype=np.float32).reshape(-1, 4))
    odo.write_kitti(out_path)
    return odo


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="LiDAR-only frame-to-frame odometry over KITTI velodyne sweeps on an MI355X")
    ap.add_argument("velodyne_dir")
    ap.add_argument("out", help="pose file, one 3x4 per line (KITTI odometry format)")
    ap.add_argument("--calib", default=None, help="KITTI calib.txt (its Tr row); default: the synthetic rig's velo_to_cam")
    ap.add_argument("--icp-skip", type=int, default=1, help="query stride per ring (the reference's constant is 200)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-frames", type=int, default=None)
    a = ap.parse_args()
    o = run_directory(a.velodyne_dir, a.out, a.calib, a.device, a.max_frames, icp_skip=a.icp_skip)
    print(f"{len(o.poses)} poses -> {a.out}")
    o.close()
