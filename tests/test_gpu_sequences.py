"""-m gpu: velo_register_sequences -- the drive loop of n sequences for K frames in ONE call (main.cpp:207-413 for n sequences; the groups walk
their drives independently) -- against the frame-by-frame calls it replaces: velo_register_batch[_visual] with VELO_SCAN_PROMOTE targets +
velo_pose_handoff.  Poses, 4x4s, hand-overs and every solve's summary must be equal bit for bit."""
import os
import sys

import numpy as np
import pytest

import velo_amd  # noqa: F401
from velo_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _summary_tuple(s):
    return [(s.solves[k].termination, s.solves[k].lm_iterations, s.solves[k].evaluations, s.solves[k].n_icp_valid, s.solves[k].n_visual_blocks,
             s.solves[k].initial_cost, s.solves[k].final_cost) for k in range(s.n_solves)]


def _stepwise(drives, vis, n_frames, host):
    import bench
    B = len(drives)
    ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
    frames = [[(np.ascontiguousarray(f[0]) if host else f[0], f[1]) for f in d["frames"]] for d in drives]
    w = bench.DriveWalker(api, ctxs, frames, 0, vis)
    out = []
    for _ in range(n_frames - 1):
        xs, Ts, Ss = w.step()
        out.append((xs.copy(), Ts.copy(), [_summary_tuple(s) for s in Ss], w.P_prev.copy(), w.x0.copy()))
    for c in ctxs:
        c.close()
    return out


@pytest.mark.parametrize("B,with_vis", [(4, False), (1, False), (3, False), (5, False), (4, True)])
def test_sequences_equal_frame_by_frame_calls(hip_lib, B, with_vis):
    import bench
    n_frames = 6
    drives = [synth.drive(n_frames, seed=20 + s, n_beams=16, n_azimuth=160) for s in range(B)]
    vis = [[synth.stereo_matches(150, seed=7 + 100 * i + k, x_true=drives[i]["x_true"][k]) for k in range(n_frames - 1)] for i in range(B)] if with_vis else None
    ref = _stepwise(drives, vis, n_frames, host=False)
    ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
    w = bench.DriveWalker(api, ctxs, [d["frames"] for d in drives], 0, vis)
    # two calls: 2 frames, then the remaining 3 -- a call continues where the last one stopped (the contexts hold their drives' current frames)
    got = []
    for K, lockstep in ((2, False), (3, True)):                      # free-running groups, then frames in lock step: the same registrations
        xs, Ts, Ss = w.walk(w.prepare(K), lockstep=lockstep)
        for f in range(K):
            got.append((xs[f], Ts[f], [_summary_tuple(s) for s in Ss[f]]))
    assert len(got) == len(ref) == n_frames - 1
    for k, ((x0, T0, s0, P0, g0), (x1, T1, s1)) in enumerate(zip(ref, got)):
        assert np.array_equal(x0, x1), (k, x0, x1)
        assert np.array_equal(T0.reshape(B, 16), T1.reshape(B, 16)) and s0 == s1, k
    assert np.array_equal(ref[-1][3], w.P_prev) and np.array_equal(ref[-1][4], w.x0)      # accumulated poses and the next guesses
    for c in ctxs:
        assert c.chain_stats()[0] == n_frames - 1                                       # every pair went through ONE chain of launches
        c.close()


def test_sequences_take_frames_from_host_memory(hip_lib):
    """frames in pageable host memory (uploaded inside the call) give the same registrations as frames resident in HBM"""
    import bench
    B, n_frames = 4, 5
    drives = [synth.drive(n_frames, seed=40 + s, n_beams=16, n_azimuth=160) for s in range(B)]
    ref = _stepwise(drives, None, n_frames, host=False)
    ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
    w = bench.DriveWalker(api, ctxs, [[(np.ascontiguousarray(f[0]), f[1]) for f in d["frames"]] for d in drives], 0)
    xs, Ts, Ss = w.walk(w.prepare(n_frames - 1))
    for k in range(n_frames - 1):
        assert np.array_equal(ref[k][0], xs[k]) and ref[k][2] == [_summary_tuple(s) for s in Ss[k]]
    for c in ctxs:
        c.close()


def test_hinted_uploads_change_nothing_but_where_the_copy_runs(hip_lib):
    """velo_hint_next_source: frames in host memory, announced one step ahead (what bench.py --host-inputs does), give the registrations of
    resident frames bit for bit; a hint for a cloud that is NOT handed over next is dropped."""
    import bench
    B, n_frames = 4, 6
    drives = [synth.drive(n_frames, seed=60 + s, n_beams=16, n_azimuth=160) for s in range(B)]
    ref = _stepwise(drives, None, n_frames, host=False)
    hinted = _stepwise(drives, None, n_frames, host=True)           # DriveWalker hints the next frames when they are numpy arrays
    for k in range(n_frames - 1):
        assert np.array_equal(ref[k][0], hinted[k][0]) and ref[k][2] == hinted[k][2]
    # a wrong announcement: context 0 is told frame 3 comes next, frame 2 comes
    c = api.Context(0, icp_skip=1)
    fr = [(np.ascontiguousarray(f[0]), f[1]) for f in drives[0]["frames"]]
    refs = [api.scan_refs([f], 0) for f in fr]
    c.set_source(*fr[0])
    api.hint_next_sources([c], refs[3][0])
    x1, _, _ = api.register_batch([c], None, None, synth.INITIAL_GUESS[None, :], refs=(api.promote_refs(1), refs[1]))
    api.hint_next_sources([c], refs[3][0])
    x2, _, _ = api.register_batch([c], None, None, x1, refs=(api.promote_refs(1), refs[2]))
    assert np.array_equal(x1[0], ref[0][0][0]) and np.abs(x2[0] - ref[1][0][0]).max() < 1e-6      # (the guess differs from the hand-off's in its last bits)
    c.close()


def test_sequences_report_bad_arguments_as_status_codes(hip_lib):
    c = api.Context(0, icp_skip=1)
    d = synth.drive(3, seed=1, n_beams=8, n_azimuth=64)
    refs, keep, n = api.sequence_refs([d["frames"]], 0, first=1)
    P, g = np.tile(np.eye(4), (1, 1, 1)), np.tile(synth.INITIAL_GUESS, (1, 1))
    with pytest.raises(api.VeloError):                                  # the context holds no frame to start from
        api.register_sequences([c], refs, n, P, g)
    c.set_source(*d["frames"][0])
    xs, Ts, Ss = api.register_sequences([c], refs, n, P, g)
    assert xs.shape == (2, 1, 6) and Ss[1][0].n_solves == 6
    with pytest.raises(api.VeloError):                                  # the same context twice
        api.register_sequences([c, c], api.sequence_refs([d["frames"], d["frames"]], 0, first=1)[0], 1, np.tile(np.eye(4), (2, 1, 1)), np.tile(synth.INITIAL_GUESS, (2, 1)))
    c.close()
