"""CPU: pin the oracle's pcl restatement to outputs of the reference's own semantic_depth_lib/pcl.py
(fixtures made by tests/golden/make_golden.py).  Bit-exact comparisons."""
import json
import os

import numpy as np
import pytest

from oracle import fusion, pcl, pipeline
from helpers import checksum


@pytest.fixture(scope="module")
def mini(golden_dir):
    return np.load(os.path.join(golden_dir, "pcl_mini.npz"))


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a, b, equal_nan=True)


def test_mini_scene_regenerates(mini):
    """the committed inputs are what the oracle's scene generator + fusion produce (seeded)."""
    dp, road, fence, frame, cam = pipeline.synthetic_scene(64, 128, seed=7, f=125.0, fences=True)
    _eq(dp, mini["disp_pair"]); _eq(road, mini["road_mask"]); _eq(fence, mini["fence_mask"]); _eq(frame, mini["frame_bgr"])
    fz = fusion.fuse(dp, road, fence, frame, **cam)
    _eq(fz["road3d"], mini["road3d"]); _eq(fz["road_rgb"], mini["road_rgb"])
    _eq(fz["fence3d"], mini["fence3d"])


def test_road_chain_matches_reference(mini):
    p, c = pcl.remove_from_to(mini["road3d"], mini["road_rgb"], 2, 0.0, 7.0)
    _eq(p, mini["zcut_pts"]); _eq(c, mini["zcut_col"])
    p, c = pcl.remove_noise_by_mad(p, c, 1, 15.0)
    _eq(p, mini["mad_y_pts"]); _eq(c, mini["mad_y_col"])
    p, c = pcl.remove_noise_by_mad(p, c, 0, 2.0)
    _eq(p, mini["mad_x_pts"]); _eq(c, mini["mad_x_col"])
    p, c, coeff = pcl.remove_noise_by_fitting_plane(p, c, axis=1, threshold=5.0)
    _eq(p, mini["plane_pts"]); _eq(c, mini["plane_col"])
    _eq(np.array([coeff[k] for k in ("Cx", "Cy", "Cz", "C")], np.float64), mini["plane_coeff"])
    l, r = pcl.get_end_points_of_road(p.astype(np.float64), 10.0 - 0.02)
    _eq(l, mini["left_pts"]); _eq(r, mini["right_pts"])


@pytest.mark.parametrize("axis,thr", [(0, 1.0), (1, 2.0), (2, 0.8)])
def test_mad_tight(mini, axis, thr):
    p, c = pcl.remove_noise_by_mad(mini["road3d"], mini["road_rgb"], axis, thr)
    assert 0 < len(p) < len(mini["road3d"])
    _eq(p, mini[f"mad_a{axis}_pts"]); _eq(c, mini[f"mad_a{axis}_col"])


@pytest.mark.parametrize("axis,thr", [(0, 0.5), (1, 0.02), (2, 3.0)])
def test_plane_all_axes(mini, axis, thr):
    p, c, coeff = pcl.remove_noise_by_fitting_plane(mini["road3d"], mini["road_rgb"], axis=axis, threshold=thr)
    _eq(p, mini[f"plane_a{axis}_pts"]); _eq(c, mini[f"plane_a{axis}_col"])
    _eq(np.array([coeff[k] for k in ("Cx", "Cy", "Cz", "C")], np.float64), mini[f"plane_a{axis}_coeff"])


def test_empty_window_and_mad_zero(mini):
    assert mini["empty_window_is_none"].all()
    l, r = pcl.get_end_points_of_road(mini["plane_pts"].astype(np.float64), 500.0)
    assert l is None and r is None
    with np.errstate(all="ignore"):
        p, _ = pcl.remove_noise_by_mad(mini["mad0_in"], mini["road_rgb"], 1, 15.0)
    _eq(p, mini["mad0_pts"])
    assert len(p) == 0          # MAD == 0 -> 0/0 = nan -> nothing passes '<'


def test_empty_cloud_raises_like_reference():
    with pytest.raises(ValueError):
        pcl.remove_from_to(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), 2, 0.0, 7.0)


def test_fence_helpers(mini):
    p, c = pcl.threshold_complete(mini["fence3d"], mini["fence_rgb"], 2, 35.0)
    _eq(p, mini["thr_pts"]); _eq(c, mini["thr_col"])
    a, ac, b, bc = pcl.extract_pcls(p, c)
    _eq(a, mini["split_left"]); _eq(ac, mini["split_left_col"]); _eq(b, mini["split_right"]); _eq(bc, mini["split_right_col"])
    d = pcl.compute_distance_in_3D(np.array([[1.0, 2.0, 3.0]]), np.array([[-2.0, 0.5, 7.0]]))
    assert np.float64(d) == mini["dist3d"]
    lp, rp = mini["line_in_left"].copy(), mini["line_in_right"].copy()
    line, lcol = pcl.create_3Dline_from_3Dpoints(lp, rp, [250, 0, 0])
    _eq(line, mini["line"]); _eq(lcol, mini["line_col"])
    _eq(lp, mini["line_left_after"]); _eq(rp, mini["line_right_after"])   # in-place +0.01 on y
    assert line.shape == (1001, 3)


def test_planes_intersection_closed_form():
    c1 = {"Cx": 1e-3, "Cy": -1.0, "Cz": 2e-4, "C": -1.5}
    c2 = {"Cx": -1.0, "Cy": 0.02, "Cz": -0.01, "C": 3.9}
    pt = pcl.planes_intersection_at_certain_depth(c1, c2, 10.0)
    assert pt.shape == (1, 3) and pt[0, 2] == -10.0
    for c in (c1, c2):
        assert abs(c["Cx"] * pt[0, 0] + c["Cy"] * pt[0, 1] + c["Cz"] * pt[0, 2] + c["C"]) < 1e-12


def test_full_scene_digests(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "pcl_full.json")))
    sc = g["scene"]
    dp, road, fence, frame, cam = pipeline.synthetic_scene(sc["h"], sc["w"], seed=sc["seed"], f=sc["f"])
    fz = fusion.fuse(dp, road, fence, frame, **cam)
    assert fz["road3d"].shape[0] == g["n_road"]
    assert checksum(fz["road3d"]) == g["road3d_checksum"] and checksum(fz["road_rgb"]) == g["road_rgb_checksum"]
    p, c = pcl.remove_from_to(fz["road3d"], fz["road_rgb"], 2, 0.0, 7.0)
    assert len(p) == g["n_zcut"] and checksum(p) == g["zcut_checksum"]
    p, c = pcl.remove_noise_by_mad(p, c, 1, 15.0)
    assert len(p) == g["n_mad_y"] and checksum(p) == g["mad_y_checksum"]
    p, c = pcl.remove_noise_by_mad(p, c, 0, 2.0)
    assert len(p) == g["n_mad_x"] and checksum(p) == g["mad_x_checksum"]
    p, c, coeff = pcl.remove_noise_by_fitting_plane(p, c, axis=1, threshold=5.0)
    assert len(p) == g["n_plane"] and checksum(p) == g["plane_checksum"]
    for k in ("Cx", "Cy", "Cz", "C"):
        assert float(coeff[k]) == g["plane_coeff"][k]
    l, r = pcl.get_end_points_of_road(p.astype(np.float64), 9.98)
    assert float(l[0][0]) == g["x_left"] and float(r[0][0]) == g["x_right"]
    assert float(abs(l[0][0] - r[0][0])) == g["dist_rw"]
    # the mask half-width is 3.5 m -> ~7 m, plane offset = -camera height
    assert abs(g["dist_rw"] - 7.0) < 0.05 and abs(g["plane_coeff"]["C"] + 1.5) < 1e-3


def test_fence_chain_matches_reference(mini):
    """semantic_depth.py:273-309 with the oracle's pcl restatement vs the reference's functions (mini scene, full arrays)."""
    from oracle.pipeline import fence_tail
    road_plane = dict(zip(("Cx", "Cy", "Cz", "C"), mini["plane_coeff"]))
    ft = fence_tail(mini["fence3d"], mini["fence_rgb"], road_plane)
    assert ft["n_mad_y"] == len(mini["fc_mad_y"]) and ft["n_thr"] == len(mini["fc_thr"])
    assert (ft["n_left"], ft["n_right"]) == (len(mini["fc_left"]), len(mini["fc_right"]))
    _eq(ft["left"], mini["fc_left_final"]); _eq(ft["right"], mini["fc_right_final"])
    for side in ("left", "right"):
        got = np.array([ft[f"plane_{side}"][k] for k in ("Cx", "Cy", "Cz", "C")], np.float64)
        _eq(got, mini[f"fc_plane_{side}"])
    # both intersection points satisfy the road plane and their fence plane at z = -10
    for pt, plane in ((ft["left_pt"], ft["plane_left"]), (ft["right_pt"], ft["plane_right"])):
        for c in (road_plane, plane):
            assert abs(c["Cx"] * pt[0] + c["Cy"] * pt[1] + c["Cz"] * pt[2] + c["C"]) < 1e-9
    assert 7.5 < ft["dist"] < 9.0          # fence strips at |X| in [4, 4.3)


def test_numpy_mean_is_chunked_pairwise():
    """the documented summation order of np.mean on a float32 column (what the HIP extract_pcls reproduces):
    8192-element chunks, numpy's pairwise routine per chunk, chunk sums added in order, float32 division."""
    def pw(a):
        n = len(a)
        if n < 8:
            r = np.float32(0.)
            for v in a:
                r = np.float32(r + v)
            return r
        if n <= 128:
            r = [np.float32(a[j]) for j in range(8)]
            i = 8
            while i < n - (n % 8):
                for j in range(8):
                    r[j] = np.float32(r[j] + a[i + j])
                i += 8
            res = np.float32(np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3])) +
                             np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7])))
            while i < n:
                res = np.float32(res + a[i]); i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return np.float32(pw(a[:n2]) + pw(a[n2:]))
    rng = np.random.default_rng(3)
    for n in (1, 7, 8, 129, 8191, 8192, 8193, 20011, 70001):
        col = (rng.standard_normal((n, 3)) * 5).astype(np.float32)[:, 0]
        tot = None
        for i in range(0, n, 8192):
            s = pw(col[i:i + 8192])
            tot = s if tot is None else np.float32(tot + s)
        assert np.float32(tot / np.float32(n)) == np.mean(col), n


def test_threshold_edges_match_the_reference(mini):
    """cuts that float32 cannot represent, points at the float32 neighbours of a cut, plane residuals within rounding of the
    threshold: the oracle takes the reference's side of every comparison (goldens captured under numpy >= 2)"""
    e, ec = mini["edge_in"], mini["edge_col"]
    for tag, cut in (("70", 7.0), ("71", 7.1), ("69", 6.999999)):
        p, c = pcl.remove_from_to(e, ec, 2, 0.0, cut)
        _eq(p, mini[f"edge_rft_{tag}_pts"]); _eq(c, mini[f"edge_rft_{tag}_col"])
        assert 0 < len(p) < len(e)
    _eq(pcl.threshold_complete(e, ec, 2, 35.0)[0], mini["edge_thr35_pts"])
    _eq(pcl.threshold_complete(e, ec, 2, 0.1)[0], mini["edge_thr01_pts"])
    p, c, coeff = pcl.remove_noise_by_fitting_plane(mini["edge_plane_in"], mini["edge_plane_col"], axis=1, threshold=0.75)
    _eq(p, mini["edge_plane_pts"])
    assert len(mini["edge_plane_in"]) - 16 <= len(p) < len(mini["edge_plane_in"])      # some of the 16 edge points go, none of the inliers
    np.testing.assert_allclose([coeff[k] for k in ("Cx", "Cy", "Cz", "C")], mini["edge_plane_coeff"], rtol=1e-12, atol=1e-14)
