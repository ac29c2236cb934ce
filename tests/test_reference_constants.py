"""CPU, in the authoring container only: row P1 of SURVEY.md 8(a) -- the constants the path reads -- checked against the REFERENCE'S OWN
TEXT (/root/reference/kitti.h:3-35, main.cpp:170, lru.h:5-6) instead of against this repository's recollection of it.  The reference is
read at test time and nothing of it is kept here; on a box without /root/reference (the GPU box) the tests skip -- the same constants are
held together by tests/test_abi.py::test_default_params_match_header_python_and_oracle everywhere."""
import os
import re

import numpy as np
import pytest

import oracle_lib as ol
import velo_amd  # noqa: F401
from velo_amd import api, odometry, synth

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "kitti.h")), reason="the reference checkout is not on this box")


def _constants():
    """`name = value` pairs of the two constant lists at the top of kitti.h (const int ...; const double ...;)."""
    text = open(os.path.join(REF, "kitti.h")).read()
    out = {}
    for block in re.findall(r"const\s+(?:int|double)\s+(.*?);", text, flags=re.S):
        block = re.sub(r"//[^\n]*", "", block)
        for name, value in re.findall(r"(\w+)\s*=\s*([-+0-9.eE]+)", block):
            out[name] = float(value)
    return out


def test_default_params_are_the_references_constants():
    K = _constants()
    assert len(K) >= 30 and K["num_cams"] == 2
    for P in (api.default_params(), ol.default_params()):                 # the HIP library's and the oracle's
        for name in ("icp_skip", "f2f_iterations", "icp_iterations", "weight_3D2D", "weight_2D2D", "weight_3DPD", "loss_thresh_3D2D",
                     "loss_thresh_2D2D", "loss_thresh_3DPD", "loss_thresh_3D3D", "outlier_reject", "correspondence_thresh_icp", "icp_norm_condition"):
            assert float(getattr(P, name)) == K[name], name
    # the rows widened into: depth association threshold (kitti.h:28), agreement / loop-closure thresholds (kitti.h:33-35), ndiagonal
    assert odometry.AGREEMENT_T_THRESH == K["agreement_t_thresh"] and odometry.AGREEMENT_R_THRESH == K["agreement_r_thresh"]
    assert odometry.LOOP_CLOSE_THRESH == K["loop_close_thresh"]
    assert K["depth_assoc_thresh"] == 0.015 and K["ndiagonal"] == 4


def test_start_up_guess_and_cache_capacity_are_the_references():
    main = open(os.path.join(REF, "main.cpp")).read()
    m = re.search(r"double\s+transform\[6\]\s*=\s*\{([^}]*)\}", main)
    assert m, "main.cpp no longer declares the start-up transform"
    guess = [float(v) for v in m.group(1).split(",")]
    assert guess == list(synth.INITIAL_GUESS) == list(odometry.FIRST_GUESS)        # main.cpp:170
    lru = open(os.path.join(REF, "lru.h")).read()
    m = re.search(r"class\s+ScansLRU\s*\{.*?const\s+int\s+size\s*=\s*(\d+)", lru, flags=re.S)      # lru.h:33
    assert m, "lru.h no longer declares ScansLRU::size"
    import inspect
    assert inspect.signature(api.ScanCache.__init__).parameters["capacity"].default == int(m.group(1)) == 50
    assert inspect.signature(odometry.LidarOdometer.__init__).parameters["cache_capacity"].default == int(m.group(1))


def test_velo_h_gates_the_squared_distance_by_iter_to_the_fourth():
    """velo.h:829: the correspondence gate is correspondence_thresh_icp / iter^4 on the SQUARED distance -- the line itself."""
    velo = open(os.path.join(REF, "velo.h")).read()
    assert re.search(r"correspondence_thresh_icp\s*/\s*iter\s*/\s*iter\s*/\s*iter\s*/\s*iter", velo)
    d = synth.scan_pair(n_beams=8, n_azimuth=64)
    o = ol.Oracle(icp_skip=1)
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    o.associate(d["x_true"], 2)
    c = o.correspondences()
    has = c["ring_j"] >= 0
    assert np.all(c["dist_j"][has] <= 0.5 / 16 + 1e-7) and np.all(c["dist_i"][c["ring_i"] >= 0] <= 0.5 / 16 + 1e-7)
