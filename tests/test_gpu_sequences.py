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


def _stepwise(drives, vis, n_frames, host, ahead=None, stats=None, one_call=None):
    import bench
    B = len(drives)
    ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
    frames = [[(np.ascontiguousarray(f[0]) if host else f[0], f[1]) for f in d["frames"]] for d in drives]
    w = bench.DriveWalker(api, ctxs, frames, 0, vis, ahead=ahead, one_call=one_call)
    out = []
    for _ in range(n_frames - 1):
        xs, Ts, Ss = w.step()
        out.append((xs.copy(), Ts.copy(), [_summary_tuple(s) for s in Ss], w.P_prev.copy(), w.x0.copy()))
    for c in ctxs:
        if stats is not None:
            stats.append(c.chain_stats())
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


def _same_steps(a, b):
    assert len(a) == len(b)
    for k, (p, q) in enumerate(zip(a, b)):
        assert np.array_equal(p[0], q[0]), (k, p[0], q[0])                              # poses
        assert np.array_equal(p[1], q[1]) and p[2] == q[2], k                           # 4x4s, every solve's summary
        assert np.array_equal(p[3], q[3]) and np.array_equal(p[4], q[4]), k             # accumulated poses, next guesses


@pytest.mark.parametrize("B,with_vis,host", [(4, False, False), (8, False, False), (2, False, True), (5, True, False), (1, False, False)])
def test_frames_loaded_ahead_change_nothing_but_when_the_loads_run(hip_lib, B, with_vis, host):
    """velo_hint_next_frame: every step announces the next one, whose promotion, ingest and index build are enqueued behind the step's own
    chain of launches (what bench.py's drive steps do).  Same registrations bit for bit as steps that load their own frames; one job, the
    single-pair path, does not load ahead and is none the worse for the announcement."""
    n_frames = 6
    drives = [synth.drive(n_frames, seed=80 + s, n_beams=16, n_azimuth=160) for s in range(B)]
    vis = [[synth.stereo_matches(150, seed=9 + 100 * i + k, x_true=drives[i]["x_true"][k]) for k in range(n_frames - 1)] for i in range(B)] if with_vis else None
    st0, st1, st2 = [], [], []
    plain = _stepwise(drives, vis, n_frames, host, ahead=False, stats=st0)
    ahead = _stepwise(drives, vis, n_frames, host, ahead=True, stats=st1, one_call=False)   # velo_hint_next_frame + velo_register_batch + velo_pose_handoff
    _same_steps(plain, ahead)
    one = _stepwise(drives, vis, n_frames, host, ahead=True, stats=st2, one_call=True)      # the step as ONE call: velo_register_sequences, VELO_SEQ_ANNOUNCE
    _same_steps(plain, one)
    assert st0 == st1 == st2 and all(s[0] == n_frames - 1 for s in st1)                 # every pair through one chain of launches, as many repeats


def test_frames_of_more_rings_than_the_group_launches_take_are_loaded_ahead_too(hip_lib):
    """The three launches that load a whole group's next frames carry the ring tables in their arguments: up to 64 rings.  Scans of more rings
    are loaded ahead through the general loaders, launch by launch -- same registrations."""
    B, n_frames = 4, 5
    drives = [synth.drive(n_frames, seed=160 + s, n_beams=72, n_azimuth=120) for s in range(B)]
    assert len(drives[0]["frames"][0][1]) - 1 > 64
    plain = _stepwise(drives, None, n_frames, False, ahead=False)
    one = _stepwise(drives, None, n_frames, False, ahead=True, one_call=True)
    two = _stepwise(drives, None, n_frames, False, ahead=True, one_call=False)
    _same_steps(plain, one)
    _same_steps(plain, two)


@pytest.mark.parametrize("B", [4, 1])                                                   # lock-step groups; the single-pair chain
def test_a_repeated_call_gets_its_pair_back_from_a_frame_loaded_ahead(hip_lib, monkeypatch, B):
    """With no spare launches (VELO_CHAIN_MARGIN=0) calls outrun their chains and are repeated host-driven -- after the next frame was
    enqueued behind them.  The contexts are put back on the pair they registered (the old target's cloud is kept for that), the repeat
    gives the same registration, and the drive goes on from there."""
    pick = (0, 0, 0, 1, 2, 4, 5, 8, 9)                                               # a car that stands, then pulls away, frames left out: the
    n_frames = len(pick)                                                                # constant-velocity guess is off, solves take MORE iterations than one call ago
    drives = [synth.drive(pick[-1] + 1, seed=120 + s, n_beams=32, n_azimuth=400) for s in range(B)]    # (large enough for many-launch solves)
    drives = [dict(frames=[d["frames"][k] for k in pick]) for d in drives]
    plain = _stepwise(drives, None, n_frames, False, ahead=False)
    monkeypatch.setenv("VELO_CHAIN_MARGIN", "0")
    stats = []
    tight = _stepwise(drives, None, n_frames, False, ahead=True, stats=stats, one_call=False)
    stats1 = []
    tight1 = _stepwise(drives, None, n_frames, False, ahead=True, stats=stats1, one_call=True)
    monkeypatch.delenv("VELO_CHAIN_MARGIN", raising=False)
    _same_steps(plain, tight)
    _same_steps(plain, tight1)
    assert sum(s[1] for s in stats) > 0 and stats1 == stats, (stats, stats1)            # at least one call WAS repeated


def test_a_context_one_frame_ahead_takes_only_the_announced_job(hip_lib):
    B, n_frames = 2, 5
    drives = [synth.drive(n_frames, seed=140 + s, n_beams=16, n_azimuth=160) for s in range(B)]
    ctxs = [api.Context(0, icp_skip=1) for _ in range(B)]
    frames = [d["frames"] for d in drives]
    refs = [api.scan_refs([frames[i][k] for i in range(B)], 0) for k in range(n_frames)]
    promote = api.promote_refs(B)
    for i, c in enumerate(ctxs):
        c.set_source(*frames[i][0])
    x0 = np.tile(synth.INITIAL_GUESS, (B, 1))
    api.hint_next_frames(ctxs, refs[2][0])
    x1, _, _ = api.register_batch(ctxs, None, None, x0, refs=(promote, refs[1]))         # frame 2 is loaded behind this call
    with pytest.raises(api.VeloError, match="loaded ahead"):
        api.register_batch(ctxs, None, None, x1, refs=(promote, refs[3]))                # not the announced frame
    with pytest.raises(api.VeloError, match="loaded ahead"):
        ctxs[0].frame_to_frame(x1[0])                                                   # nor a registration of what the context "held"
    x2, _, _ = api.register_batch(ctxs, None, None, x1, refs=(promote, refs[2]))         # the announced job: nothing to load
    # without an announcement the same two steps
    fresh = [api.Context(0, icp_skip=1) for _ in range(B)]
    for i, c in enumerate(fresh):
        c.set_source(*frames[i][0])
    y1, _, _ = api.register_batch(fresh, None, None, x0, refs=(promote, refs[1]))
    y2, _, _ = api.register_batch(fresh, None, None, y1, refs=(promote, refs[2]))
    assert np.array_equal(x1, y1) and np.array_equal(x2, y2)
    # the latest announcement before a call is the one that counts
    api.hint_next_frames(ctxs, refs[0][0])
    api.hint_next_frames(ctxs, refs[4][0])
    x3, _, _ = api.register_batch(ctxs, None, None, x2, refs=(promote, refs[3]))
    x4, _, _ = api.register_batch(ctxs, None, None, x3, refs=(promote, refs[4]))
    y3, _, _ = api.register_batch(fresh, None, None, y2, refs=(promote, refs[3]))
    y4, _, _ = api.register_batch(fresh, None, None, y3, refs=(promote, refs[4]))
    assert np.array_equal(x3, y3) and np.array_equal(x4, y4)
    for c in ctxs + fresh:
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
