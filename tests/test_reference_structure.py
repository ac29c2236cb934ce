"""CPU, in the authoring container only: the CHOICES the oracle restates from velo.h, checked against the reference's own text at test
time (nothing of it is kept here; skipped where /root/reference is absent).  Not a substitute for running the reference -- it cannot be
built here -- but it turns "as far as I can read it" into assertions a change of either side would trip:
which residual functor gets which loss, threshold and weight (velo.h:683-689,710-718,744-752,777-785,875-891), the strict '<' of the
ring scan and of the neighbour choice (velo.h:836,843,859), the query stride and the enable_icp multiplier (velo.h:806-807), and that
ceres::Solve runs on defaults except for the linear solver type (velo.h:897-902)."""
import os
import re

import numpy as np
import pytest

import helpers as H
import oracle_lib as ol
import velo_amd  # noqa: F401
from velo_amd import synth

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "velo.h")), reason="the reference checkout is not on this box")
KIND = {"cost3D3D": 0, "cost3D2D": 1, "cost2D3D": 2, "cost2D2D": 3, "cost3DPD": 4}          # ResidualType order + the point-to-plane block


def _velo():
    return open(os.path.join(REF, "velo.h")).read()


def _loss_sites(text):
    """{functor: (dim, loss class, threshold name, weight name or None)} from every AddResidualBlock of frameToFrame."""
    body = text[text.index("Eigen::Matrix4d frameToFrame("):text.rindex("void residualStats(")]      # the function's own body
    out = {}
    for m in re.finditer(r"AutoDiffCostFunction<\s*(cost\w+)\s*,\s*(\d)\s*,\s*6\s*>", body):
        tail = body[m.end():m.end() + 1200]
        tail = tail[tail.index("AddResidualBlock("):]
        tail = tail[:tail.index("transform)")]
        scaled = re.search(r"ScaledLoss\(\s*new\s+ceres::(\w+Loss)\((\w+)\)\s*,\s*(\w+)", tail)
        plain = re.search(r"new\s+ceres::(\w+Loss)\((\w+)\)", tail)
        assert scaled or plain, (m.group(1), tail)
        out[m.group(1)] = (int(m.group(2)),) + ((scaled.group(1), scaled.group(2), scaled.group(3)) if scaled else (plain.group(1), plain.group(2), None))
    return out


def test_every_block_kind_carries_the_references_loss():
    sites = _loss_sites(_velo())
    assert set(sites) == set(KIND)
    assert sites["cost3D3D"] == (3, "ArctanLoss", "loss_thresh_3D3D", None)
    assert sites["cost3DPD"][1] == "CauchyLoss" and sites["cost2D2D"][1] == "ArctanLoss"
    # ... and the oracle's blocks say the same, kind by kind
    d = H.small_pair()
    vis = synth.stereo_matches(80, mix="all", outlier_frac=0.0, x_true=d["x_true"])
    o = ol.Oracle(icp_skip=1)
    o.set_target(d["tgt_xyz"], d["tgt_off"])
    o.set_source(d["src_xyz"], d["src_off"])
    o.set_visual(vis)
    o.build_visual(d["x0"], 1)
    o.associate(d["x0"], 1)
    blocks = o.blocks()
    P = o.params
    dims = {0: 3, 1: 2, 2: 2, 3: 1, 4: 1}
    for name, kind in KIND.items():
        dim, loss, thresh, weight = sites[name]
        b = blocks[blocks["kind"] == kind]
        assert len(b) > 0, name
        assert dim == dims[kind]
        assert np.all(b["loss_type"] == {"CauchyLoss": 1, "ArctanLoss": 2}[loss]), name
        assert np.all(b["loss_a"] == float(getattr(P, thresh))), (name, thresh)
        assert np.all(b["loss_w"] == (float(getattr(P, weight)) if weight else 1.0)), (name, weight)


def test_ring_scan_neighbour_choice_and_query_stride_as_written():
    v = _velo()
    # velo.h:836,843: strict '<' twice -- ties keep the lower ring; the second test is an else-if
    assert re.search(r"if\s*\(\s*d\s*<\s*np_dist_i\s*\)", v) and re.search(r"else\s+if\s*\(\s*d\s*<\s*np_dist_j\s*\)", v)
    # velo.h:859: the +1 neighbour only when STRICTLY closer
    assert re.search(r"if\s*\(\s*util::norm2\(np_k_1\)\s*<\s*util::norm2\(np_k_2\)\s*\)", v)
    # velo.h:806-807: the ring loop bound is multiplied by enable_icp, the point loop strides by icp_skip from 0
    assert re.search(r"sm\s*<\s*scans_M\.size\(\)\s*\*\s*enable_icp", v)
    assert re.search(r"smi\s*=\s*0\s*;\s*smi\s*<\s*scans_M\[sm\]->size\(\)\s*;\s*smi\s*\+=\s*icp_skip", v)
    # velo.h:828-829: a ring counts when the search found something and the SQUARED distance is within the gate
    assert re.search(r"nearestKSearch\(\s*pointM\s*,\s*1\s*,\s*id\s*,\s*dist2\s*\)\s*<=\s*0", v)
    # velo.h:873: degenerate planes are skipped on the norm of the un-normalised normal
    assert re.search(r"icp_norm_condition", v)


def test_the_solve_runs_on_ceres_defaults_except_the_linear_solver():
    v = _velo()
    body = v[v.index("Eigen::Matrix4d frameToFrame("):]
    body = body[:body.index("ceres::Solve(options")]
    opts = re.findall(r"^\s*options\.(\w+)\s*=\s*([^;]+);", body, flags=re.M)
    assert dict(opts) == {"linear_solver_type": "ceres::DENSE_SCHUR", "minimizer_progress_to_stdout": "false"}, opts
    P = ol.default_params()                                                      # what the oracle (and the library) take for "defaults"
    assert (P.max_num_iterations, P.function_tolerance, P.gradient_tolerance, P.parameter_tolerance) == (50, 1e-6, 1e-10, 1e-8)
    assert (P.initial_trust_region_radius, P.min_relative_decrease, P.min_lm_diagonal, P.max_lm_diagonal) == (1e4, 1e-3, 1e-6, 1e32)


def test_rounding_points_of_utility_h_as_written():
    """utility.h:35-39,51-53,97-103: where the reference computes in float and where in double -- the oracle's sub_norm2 and
    transform_point restate exactly these lines (and the HIP kernels follow the oracle bit for bit)."""
    u = open(os.path.join(REF, "utility.h")).read()
    # subtract_assign: component-wise float subtraction IN PLACE on a pcl::PointXYZ (float members)
    m = re.search(r"subtract_assign\(pcl::PointXYZ\s*&a,\s*const\s+pcl::PointXYZ\s*&b\)\s*\{(.*?)\}", u, flags=re.S)
    assert m and re.sub(r"\s", "", m.group(1)) == "a.x-=b.x;a.y-=b.y;a.z-=b.z;"
    # norm2: the FLOAT expression x*x + y*y + z*z, widened to double only by the return type
    m = re.search(r"static\s+inline\s+double\s+norm2\(const\s+pcl::PointXYZ\s*&p\)\s*\{(.*?)\}", u, flags=re.S)
    assert m and re.sub(r"\s", "", m.group(1)) == "returnp.x*p.x+p.y*p.y+p.z*p.z;"
    # transform_point: widen to double, rotate in double, add the translation in double, store back to the float members
    m = re.search(r"transform_point\(pcl::PointXYZ\s*&p,\s*const\s+double\s+transform\[6\]\)\s*\{(.*?)\n    \}", u, flags=re.S)
    body = re.sub(r"\s", "", m.group(1))
    assert body == "doublex[3]={p.x,p.y,p.z},y[3]={0,0,0};ceres::AngleAxisRotatePoint(transform,x,y);p.x=y[0]+transform[3];p.y=y[1]+transform[4];p.z=y[2]+transform[5];"
    # the oracle does the same on a point where float and double arithmetic differ visibly
    p = np.array([1.0000001, 2.0000002, -3.0000005], dtype=np.float32)
    x = np.array([0.3, -0.2, 0.1, 1e-3, 2e-3, 3e-3])
    got = ol.transform_point(p, x)
    rot = ol.rotate_point(x[:3], p.astype(np.float64))
    want = (rot + x[3:]).astype(np.float32)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # pose_mat2vec / pose_vec2mat: column-major R from Ceres, T(i, j) = R[j * 3 + i]
    assert re.search(r"T\(i,\s*j\)\s*=\s*R\[j\*3\s*\+\s*i\]", u) and re.search(r"R\[j\*3\s*\+\s*i\]\s*=\s*T\(i,\s*j\)", u)
    T = ol.pose_vec_to_mat(x)
    Rcol = np.stack([ol.rotate_point(x[:3], e) for e in np.eye(3)], axis=1)      # column j = R e_j
    assert np.allclose(T[:3, :3], Rcol, atol=1e-15) and np.allclose(T[:3, 3], x[3:])


def test_visual_gates_as_written():
    """velo.h:674-682,709,739-742,772-775: squared norm against thresh*reject/iter * thresh*reject/iter (evaluated left to right -- the
    oracle and the gate kernel multiply in that order), |r| against thresh*reject/iter for the epipolar block (written with `abs`,
    restated as fabs), `continue` on failure -- which also skips the match's remaining residual types; only from the second iteration on."""
    v = re.sub(r"\s+", "", _velo())
    for t, n in (("3D3D", 3), ("3D2D", 2)):
        sq = "+".join(f"residual_test[{k}]*residual_test[{k}]" for k in range(n))
        assert f"if(iter>1&&{sq}>loss_thresh_{t}*outlier_reject/iter*loss_thresh_{t}*outlier_reject/iter)" in v, t
    assert v.count("*loss_thresh_3D2D*outlier_reject/iter)continue;") == 2                      # the 3D2D and the 2D3D block share the threshold
    assert "if(iter>1&&abs(residual_test[0])>loss_thresh_2D2D*outlier_reject/iter)continue;" in v
    # the left-to-right product differs from (thresh*reject/iter)^2 in the last bit for some thresholds: both restatements take the written order
    P = ol.default_params()
    a, b, it = P.loss_thresh_3D2D, P.outlier_reject, 2
    assert a * b / it * a * b / it == ((((a * b) / it) * a) * b) / it


def test_ring_segmenter_as_written():
    """kitti.h:164-183: a new ring where x > 0 and the sign of y flips (strict comparisons, prev_y starts at 0); rings re-ordered by
    _i - 1 - (i + _i/2) % _i -- what tests/segmenter_ref.py transcribes."""
    k = re.sub(r"\s+", "", open(os.path.join(REF, "kitti.h")).read())
    assert "floatprev_y=0;" in k and "if(i>0&&p.x>0&&(p.y>0)!=(prev_y>0)){scan_id++;}" in k
    assert "cloud_tmp->at(scan_ids[s][_i-1-(i+_i/2)%_i])" in k and "pcl::transformPointCloud(*point_cloud,*cloud_tmp,velo_to_cam);" in k


def test_widened_rows_as_written():
    """The lines the oracle's restatements of SURVEY 8(f) rows 3-4 follow: linterpolate in float (utility.h:7-29), the projection, window
    test and occlusion-stack rules of projectLidarToCamera (velo.h:345-366), the bisection and acceptance test of featureDepthAssociation
    (velo.h:397-422), the losses of triangulatePoint (velo.h:1078-1122)."""
    u = re.sub(r"\s+", "", open(os.path.join(REF, "utility.h")).read())
    assert u.count("floata=(mid-start)/(end-start);floatb=1-a;") == 2
    assert "floatx=p1.x*b+p2.x*a;floaty=p1.y*b+p2.y*a;floatz=p1.z*b+p2.z*a;" in u and "returnp1*b+p2*a;" in u
    v = re.sub(r"\s+", "", _velo())
    assert "pcl::PointXYZpp(p.x+t(0),p.y+t(1),p.z+t(2));cv::Point2fc(pp.x/pp.z,pp.y/pp.z);" in v
    assert "if(pp.z>0&&c.x>=min_x[cam]&&c.x<max_x[cam]&&c.y>=min_y[cam]&&c.y<max_y[cam])" in v
    assert "while(projection[s].size()>0&&c.x<projection[s].back().x&&pp.z<projected_points.back().z)" in v       # pop what the new point occludes
    assert "if(projection[s].size()>0&&c.x<projection[s].back().x&&pp.z>projected_points.back().z){bad++;continue;}" in v   # or drop the new point
    assert "if(projection[s].size()<=1){last_interp=-1;continue;}" in v
    assert "intlo=0,hi=projection[s].size()-2,mid=0;while(lo<=hi){mid=(lo+hi)/2;if(projection[s][mid].x>kp.x){hi=mid-1;}elseif(projection[s][mid+1].x<=kp.x){lo=mid+1;}else{found=true;" in v
    assert ("if(last_interp!=-1&&(projection[s][mid].y>kp.y)!=(projection[s-1][last_interp].y>kp.y)&&abs(projection[s][mid].x-projection[s][mid+1].x)<depth_assoc_thresh"
            "&&abs(projection[s-1][last_interp].x-projection[s-1][last_interp+1].x)<depth_assoc_thresh)") in v
    tri = v[v.index("triangulatePoint("):]
    assert "newceres::TrivialLoss," in tri and "newceres::ScaledLoss(newceres::CauchyLoss(loss_thresh_3D2D),weight_3D2D,ceres::TAKE_OWNERSHIP)" in tri


def test_residual_stats_as_written():
    """velo.h:921-1025: evaluated WITHOUT loss functions; one number per block (sqrt of the squared rows for 3D3D / 3D2D / 2D3D, `abs` for
    2D2D and for everything behind the visual residuals = 3DPD); sums in block order, then sort; printed: element size/2 and sum/size."""
    v = re.sub(r"\s+", "", _velo())
    st = v[v.rindex("voidresidualStats("):v.index("Eigen::Vector3dtriangulatePoint(") if "Eigen::Vector3dtriangulatePoint(" in v else len(v)]
    assert "evaluate_options.apply_loss_function=false;" in st
    assert "residuals_3D3D.push_back(sqrt(residuals[ri]*residuals[ri]+residuals[ri+1]*residuals[ri+1]+residuals[ri+2]*residuals[ri+2]));ri+=3;" in st
    assert st.count("sqrt(residuals[ri]*residuals[ri]+residuals[ri+1]*residuals[ri+1]));ri+=2;") == 2
    assert "residuals_2D2D.push_back(abs(residuals[ri]));ri++;" in st and "for(;ri<residuals.size();ri++){residuals_3DPD.push_back(abs(residuals[ri]));}" in st
    assert st.index("for(autor:residuals_3DPD){sum_3DPD+=r;}") < st.index("std::sort(residuals_3D3D.begin(),residuals_3D3D.end());")
    for t in ("3D3D", "3D2D", "2D3D", "2D2D", "3DPD"):
        assert f"residuals_{t}[residuals_{t}.size()/2]" in st and f"sum_{t}/residuals_{t}.size()" in st, t
