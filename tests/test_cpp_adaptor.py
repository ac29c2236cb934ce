"""The header-only C++ adaptor (include/velo_frame_to_frame.hpp) offers the reference's frameToFrame parameter list
(velo.h:598-614).  CPU: it compiles as plain C++11 against stand-in container types and links to the C-ABI library.
GPU: driven like main.cpp:388-405 it returns the oracle's pose, good_matches and residual_type."""
import os
import struct
import subprocess

import numpy as np
import pytest

import helpers as H
import velo_amd  # noqa: F401
from velo_amd import api, build, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def compile_adaptor(tmp_path) -> str:
    build.build_hip()
    exe = str(tmp_path / "test_adaptor")
    csrc = os.path.dirname(build.LIB)
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", CPP,
                    os.path.join(CPP, "test_adaptor.cpp"), "-o", exe, "-L", csrc, "-lvelo_hip", f"-Wl,-rpath,{csrc}",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def write_case(path, d, matches, skip):
    with open(path, "wb") as f:
        for xyz, off in ((d["src_xyz"], d["src_off"]), (d["tgt_xyz"], d["tgt_off"])):
            f.write(struct.pack("i", len(off) - 1))
            f.write(np.asarray(off, np.int32).tobytes())
            f.write(np.ascontiguousarray(xyz, np.float32).tobytes())
        f.write(struct.pack("ii", len(matches), skip))
        f.write(matches.tobytes())
        f.write(np.asarray(d["x0"], np.float64).tobytes())


def test_plain_c99_caller_of_the_pose_handoff(tmp_path):
    """include/velo_hip.h compiles as C99 and the drive loop's host-side entry point (velo_pose_handoff, main.cpp:311-331,408) gives the
    hand-computed chain; velo_register_batch / _visual / velo_source_to_target resolve at link time.  Runs without a GPU."""
    build.build_hip()
    exe = str(tmp_path / "test_handoff")
    csrc = os.path.dirname(build.LIB)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(CPP, "test_handoff.c"), "-o", exe,
                    "-L", csrc, "-lvelo_hip", "-lm", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "handoff ok" in out.stdout, out.stdout + out.stderr


def test_pose_chain_of_the_cxx_adaptor(tmp_path):
    """velo_hip::PoseChain -- the reference's hand-off, chaining, agreement and edge-skipping rule (main.cpp:305-331,407-437) as the C++
    adaptor offers them -- against hand-computed values.  CPU only."""
    build.build_hip()
    exe = str(tmp_path / "test_pose_chain")
    csrc = os.path.dirname(build.LIB)
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", CPP, os.path.join(CPP, "test_pose_chain.cpp"), "-o", exe,
                    "-L", csrc, "-lvelo_hip", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "pose chain ok" in out.stdout, out.stdout + out.stderr


def test_adaptor_compiles_as_cxx11(tmp_path):
    assert os.path.exists(compile_adaptor(tmp_path))


@pytest.mark.gpu
def test_adaptor_matches_oracle(tmp_path, oracle):
    exe = compile_adaptor(tmp_path)
    d = H.small_pair(16, 128)
    m = api.matches_from_dict(synth.stereo_matches(60, mix="all"))
    per_cam = np.concatenate([np.arange(60), np.arange(60)]).astype(np.int32)     # the C++ side indexes keypoints per camera
    m["point1"] = per_cam
    m["point2"] = per_cam
    case = str(tmp_path / "case.bin")
    write_case(case, d, m, 2)
    out = subprocess.run([exe, case], check=True, capture_output=True, text=True).stdout.splitlines()
    x = np.array([float(v) for v in out[0].split()[1:]])
    T = np.array([float(v) for v in out[1].split()[1:]]).reshape(4, 4)
    good = np.array([[int(v) for v in line.split()[1:]] for line in out[2:] if line.startswith("g ")]).reshape(-1, 4)
    orc = oracle.Oracle(icp_skip=2)
    orc.set_target(d["tgt_xyz"], d["tgt_off"])
    orc.set_source(d["src_xyz"], d["src_off"])
    orc.set_visual(m)
    xo, To, so = orc.frame_to_frame(d["x0"])
    assert H.pose_close(x, xo), (x, xo)
    np.testing.assert_allclose(T, To, atol=1e-6)
    g = orc.good_matches()
    # the adaptor returns good_matches per camera (cam-major), like the reference's vectors
    want = np.stack([g["cam"], g["point1"], g["point2"], g["residual_type"]], 1)
    want = want[np.argsort(want[:, 0], kind="stable")]
    assert np.array_equal(good, want)

    # the device-resident ScansLRU through the adaptor: two misses (two reads), two hits, same pose as fresh uploads
    cl = [line.split() for line in out if line.startswith("c ")][0]
    assert [int(v) for v in cl[1:6]] == [0, 0, 1, 1, 2]
    assert cl[6:12] == cl[12:18]
    # frameToFrame with empty ring vectors registers the scans the context holds (here: loaded from the cache): same result
    assert [line for line in out if line.startswith("s ")] == ["s 1"]
    # the exact 15-parameter signature of velo.h:598-614 (process-default context and rig) gives the same answer
    assert [line for line in out if line.startswith("e ")] == ["e 1"]

    # depth rows through the adaptor: projectLidarToCamera + featureDepthAssociation on the target rings, per camera
    import oracle_lib as O
    rig_window = [-0.84466541, 0.8608222, -0.25765342, 0.25705326]          # velo_hip::Rig defaults
    for cam in (0, 1):
        proj, pts, off = O.project_lidar(d["tgt_xyz"], d["tgt_off"], synth.CAM_TRANS[cam], rig_window)
        kps = m["p2_2"][m["cam"] == cam]
        kd, has = O.depth_association(proj, pts, off, kps, 0.2)
        assert len(kd) > 5
        head = [line.split() for line in out if line.startswith(f"d {cam} ")][0]
        assert [int(v) for v in head[2:]] == [len(off) - 1, len(proj), len(kd)]
        rows = [line.split() for line in out if line.startswith(f"h {cam} ")]
        assert len(rows) == len(kps)
        got_has = np.array([int(r[3]) for r in rows])
        assert np.array_equal(got_has, has)
        got_xyz = np.array([[np.float32(v) for v in r[4:7]] for r in rows], dtype=np.float32)[has >= 0]
        assert np.array_equal(got_xyz.view(np.uint32), kd[has[has >= 0]].view(np.uint32))


@pytest.mark.gpu
def test_adaptor_triangulates_like_the_oracle(tmp_path):
    import oracle_lib as O
    exe = compile_adaptor(tmp_path)
    pr = synth.triangulation_problem(150, n_frames=8, seed=17)
    case = str(tmp_path / "tri.bin")
    with open(case, "wb") as f:
        f.write(struct.pack("i", len(pr["camera_poses"])))
        f.write(np.ascontiguousarray(pr["camera_poses"], np.float64).tobytes())
        f.write(struct.pack("i", len(pr["points0"])))
        f.write(np.ascontiguousarray(pr["obs_offsets"], np.int32).tobytes())
        f.write(np.ascontiguousarray(pr["obs"], api.TRI_OBS_DTYPE).tobytes())
        f.write(np.ascontiguousarray(pr["points0"], np.float32).tobytes())
        f.write(np.ascontiguousarray(pr["initial_guess"], np.uint8).tobytes())
    out = subprocess.run([exe, "--tri", case], check=True, capture_output=True, text=True).stdout.splitlines()
    got = np.array([[np.float32(v) for v in line.split()[2:5]] for line in out if line.startswith("p ")], dtype=np.float32)
    want, res = O.triangulate_points(pr["camera_poses"], pr["cam_trans"], pr["obs"], pr["obs_offsets"], pr["points0"], pr["initial_guess"])
    assert got.shape == want.shape
    conv = res["termination"] == 0
    scale = np.maximum(1.0, np.linalg.norm(want, axis=1))
    assert np.all(np.linalg.norm(got - want, axis=1)[conv] / scale[conv] <= 1e-4)
    assert np.all(got.view(np.uint32) == want.view(np.uint32), axis=1).mean() >= 0.99


def compile_ceres_cost(tmp_path) -> str:
    build.build_hip()
    exe = str(tmp_path / "test_ceres_cost")
    csrc = os.path.dirname(build.LIB)
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", CPP,
                    os.path.join(CPP, "test_ceres_cost.cpp"), "-o", exe, "-L", csrc, "-lvelo_hip", f"-Wl,-rpath,{csrc}",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_ceres_cost_adaptor_compiles_as_cxx11(tmp_path):
    assert os.path.exists(compile_ceres_cost(tmp_path))


@pytest.mark.gpu
def test_ceres_cost_adaptor_rows_match_oracle(tmp_path, oracle):
    """Seam 3: the batched ceres::CostFunction (include/velo_ceres_cost.hpp) returns the robustified rows of all current blocks:
    residuals and row-major Jacobian equal the oracle's, with and without the Jacobian requested."""
    exe = compile_ceres_cost(tmp_path)
    d = H.small_pair(16, 128)
    m = api.matches_from_dict(synth.stereo_matches(30, mix="all"))
    case, outp = str(tmp_path / "case.bin"), str(tmp_path / "rows.bin")
    write_case(case, d, m, 2)
    subprocess.run([exe, case, outp], check=True, capture_output=True, text=True)
    raw = np.fromfile(outp, dtype=np.float64)
    n = int(raw[0])
    r, J = raw[1:1 + n], raw[1 + n:1 + 7 * n].reshape(n, 6)
    orc = oracle.Oracle(icp_skip=2)
    orc.set_target(d["tgt_xyz"], d["tgt_off"]); orc.set_source(d["src_xyz"], d["src_off"]); orc.set_visual(m)
    orc.associate(d["x0"], 1); orc.build_visual(d["x0"], 1)
    ro, Jo = orc.evaluate_rows(d["x0"])
    assert n == len(ro) > 100
    assert H.rel_err(r, ro) <= 1e-12 and H.rel_err(J, Jo) <= 1e-12


def compile_functor_batch(tmp_path) -> str:
    build.build_hip()
    exe = str(tmp_path / "test_functor_batch")
    csrc = os.path.dirname(build.LIB)
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-Wno-missing-field-initializers", "-I", os.path.join(ROOT, "include"), "-I", CPP,
                    os.path.join(CPP, "test_functor_batch.cpp"), "-o", exe, "-L", csrc, "-lvelo_hip", f"-Wl,-rpath,{csrc}",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_functor_batch_adaptor_compiles_as_cxx11(tmp_path):
    assert os.path.exists(compile_functor_batch(tmp_path))


@pytest.mark.gpu
def test_functor_batch_adaptor_matches_golden_vectors(tmp_path):
    """Seam 2 through the C++ packers: the committed functor vectors (tests/golden/functors.npz), one pose per run."""
    exe = compile_functor_batch(tmp_path)
    f = np.load(os.path.join(ROOT, "tests", "golden", "functors.npz"))
    for k, x in enumerate(np.unique(f["x"], axis=0)):
        sel = np.all(f["x"] == x, axis=1)
        n = int(sel.sum())
        case, outp = str(tmp_path / f"fn{k}.bin"), str(tmp_path / f"fn{k}.out")
        rec = np.concatenate([f["kind"][sel].astype(np.float64)[:, None], f["c"][sel]], axis=1)
        with open(case, "wb") as fh:
            fh.write(np.float64(n).tobytes()); fh.write(np.ascontiguousarray(rec).tobytes()); fh.write(np.asarray(x, np.float64).tobytes())
        subprocess.run([exe, case, outp], check=True, capture_output=True)
        got = np.fromfile(outp, dtype=np.float64).reshape(n, 21)
        np.testing.assert_allclose(got[:, :3], f["r"][sel], rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(got[:, 3:].reshape(n, 3, 6), f["J"][sel], rtol=1e-12, atol=1e-13)
