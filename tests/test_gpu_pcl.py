"""GPU parity: the pcl.py mirror and the road-width tail (csrc/pcl.hip) vs golden vectors captured from the
reference's pcl.py, and vs the oracle for the Open3D filters.  Bit-exact selections."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import o3d as oracle_o3d
from oracle import pipeline
from oracle import pcl as oracle_pcl
from helpers import checksum
from gpu_common import Camera, RoadWidthParams, dev, engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pcl():
    eng = engine(512, 1024, 2, "resnet50", load=())[0]
    from semantic_depth_amd import pcl as m
    m.set_engine(eng)
    return m


@pytest.fixture(scope="module")
def mini(golden_dir):
    return np.load(os.path.join(golden_dir, "pcl_mini.npz"))


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert np.array_equal(a, b, equal_nan=True)


def test_road_chain_golden(pcl, mini):
    p, c = pcl.remove_from_to(mini["road3d"], mini["road_rgb"], 2, 0.0, 7.0)
    _eq(p, mini["zcut_pts"]); _eq(c, mini["zcut_col"])
    p, c = pcl.remove_noise_by_mad(p, c, 1, 15.0)
    _eq(p, mini["mad_y_pts"]); _eq(c, mini["mad_y_col"])
    p, c = pcl.remove_noise_by_mad(p, c, 0, 2.0)
    _eq(p, mini["mad_x_pts"]); _eq(c, mini["mad_x_col"])
    p, c, _, _, coeff = pcl.remove_noise_by_fitting_plane(p, c, axis=1, threshold=5.0)
    _eq(p, mini["plane_pts"]); _eq(c, mini["plane_col"])
    got = np.array([coeff[k] for k in ("Cx", "Cy", "Cz", "C")])
    assert np.allclose(got, mini["plane_coeff"], rtol=1e-9, atol=1e-12), (got, mini["plane_coeff"])
    l, r = pcl.get_end_points_of_road(p.astype(np.float64), 10.0 - 0.02)
    _eq(l[0], mini["left_pts"][0]); _eq(r[0], mini["right_pts"][0])


@pytest.mark.parametrize("axis,thr", [(0, 1.0), (1, 2.0), (2, 0.8)])
def test_mad_tight_golden(pcl, mini, axis, thr):
    p, c = pcl.remove_noise_by_mad(mini["road3d"], mini["road_rgb"], axis, thr)
    _eq(p, mini[f"mad_a{axis}_pts"]); _eq(c, mini[f"mad_a{axis}_col"])


@pytest.mark.parametrize("axis,thr", [(0, 0.5), (1, 0.02), (2, 3.0)])
def test_plane_all_axes_golden(pcl, mini, axis, thr):
    p, c, _, _, coeff = pcl.remove_noise_by_fitting_plane(mini["road3d"], mini["road_rgb"], axis=axis, threshold=thr)
    _eq(p, mini[f"plane_a{axis}_pts"]); _eq(c, mini[f"plane_a{axis}_col"])
    got = np.array([coeff[k] for k in ("Cx", "Cy", "Cz", "C")])
    assert np.allclose(got, mini[f"plane_a{axis}_coeff"], rtol=1e-8, atol=1e-10)


def test_edge_cases_golden(pcl, mini):
    l, r = pcl.get_end_points_of_road(mini["plane_pts"].astype(np.float64), 500.0)
    assert l is None and r is None                      # empty depth window
    p, _ = pcl.remove_noise_by_mad(mini["mad0_in"], mini["road_rgb"], 1, 15.0)
    _eq(p, mini["mad0_pts"])                            # MAD == 0 -> nothing survives
    with pytest.raises(ValueError):
        pcl.remove_from_to(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8), 2, 0.0, 7.0)
    p, c = pcl.threshold_complete(mini["fence3d"], mini["fence_rgb"], 2, 35.0)
    _eq(p, mini["thr_pts"]); _eq(c, mini["thr_col"])
    # host-side O(1) helpers
    assert np.float64(pcl.compute_distance_in_3D(np.array([[1.0, 2.0, 3.0]]), np.array([[-2.0, 0.5, 7.0]]))) == mini["dist3d"]
    lp, rp = mini["line_in_left"].copy(), mini["line_in_right"].copy()
    line, lcol = pcl.create_3Dline_from_3Dpoints(lp, rp, [250, 0, 0])
    _eq(line, mini["line"]); _eq(lp, mini["line_left_after"])


@pytest.mark.parametrize("n", [1, 2, 3, 64, 65, 1000, 1023, 1024, 1025, 4096, 100001])
def test_median_matches_numpy(pcl, n):
    rng = np.random.default_rng(n)
    v = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
    if n > 10:
        v[rng.integers(0, n, n // 3)] = v[0]            # heavy ties
        v[1] = np.inf; v[2] = -np.inf
    dev_, m = pcl.mad(v)
    ref_dev, ref_m = oracle_pcl.mad(v)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(dev_, ref_dev, equal_nan=True)
    assert np.float32(m) == np.float32(ref_m) or (np.isnan(m) and np.isnan(ref_m))


@pytest.mark.parametrize("kind", ["random", "sorted", "ties", "periodic", "two_level", "even_gap"])
def test_median_large_sample_bracket_and_fallback(pcl, kind):
    """n >= 16384 takes the sample-bracketed median (one collecting pass); "periodic" and "two_level" are built so that
    the systematic sample misrepresents the column and the exact radix select has to take over.  Bit-exact either way."""
    rng = np.random.default_rng(7)
    n = 70_000 if kind != "even_gap" else 65_536
    v = rng.standard_normal(n).astype(np.float32)
    if kind == "sorted":
        v = np.sort(v)[::-1].copy()
    elif kind == "ties":
        v = np.round(v * 2).astype(np.float32)
    elif kind == "periodic":          # every sampled row (stride n/4096) is an outlier
        idx = (np.arange(4096, dtype=np.int64) * n) // 4096
        v[idx] = 1e6
    elif kind == "two_level":         # half the column is one value: the bracket overflows its buffer
        v[: n // 2 + 5] = 0.25
    elif kind == "even_gap":          # even n, the two middle elements far apart
        v = np.concatenate([np.full(n // 2, -3.0, np.float32), np.full(n // 2, 5.0, np.float32)])
        rng.shuffle(v)
    dev_, m = pcl.mad(v)
    ref_dev, ref_m = oracle_pcl.mad(v)
    assert np.array_equal(dev_, ref_dev)
    assert np.float32(m) == np.float32(ref_m)
    v[123] = np.nan
    _, m = pcl.mad(v)
    assert np.isnan(m)


def test_median_with_nan_is_nan(pcl):
    v = np.arange(100, dtype=np.float32)
    v[17] = np.nan
    _, m = pcl.mad(v)
    assert np.isnan(m)


def _o3d_compare(pcl, pts, col, k=10, ratio=0.5, nb=80, radius=0.5):
    keep, mean_d, _, _ = oracle_o3d.statistical_outlier_mask(pts, k, ratio)
    p, c = pcl.statistical_outlier_removal(pts, col, k, ratio)
    assert len(p) == int(keep.sum())
    _eq(p, pts[keep]); _eq(c, col[keep])
    keep2 = oracle_o3d.radius_outlier_mask(p, nb, radius)
    p2, c2 = pcl.radius_outlier_removal(p, c, nb, radius)
    assert len(p2) == int(keep2.sum())
    _eq(p2, p[keep2]); _eq(c2, c[keep2])
    return len(p), len(p2)


def test_o3d_filters_mini(pcl, mini):
    n1, n2 = _o3d_compare(pcl, mini["zcut_pts"], mini["zcut_col"], nb=8)
    assert 0 < n2 < n1 < len(mini["zcut_pts"])
    # duplicates (mean distance 0 -> dropped by the statistical filter) and an isolated far outlier (brute-force path)
    pts = np.concatenate([mini["zcut_pts"][:300], mini["zcut_pts"][:12], np.float32([[50, 40, -900]])])
    col = np.concatenate([mini["zcut_col"][:300], mini["zcut_col"][:12], np.uint8([[1, 2, 3]])])
    _o3d_compare(pcl, pts, col, nb=5)
    # fewer points than k
    _o3d_compare(pcl, mini["zcut_pts"][:7], mini["zcut_col"][:7], nb=2)


def test_o3d_sparse_cloud_deep_shells(pcl):
    """a dense sheet plus a sparse halo and far strays: most halo points need shells beyond the per-thread search (the
    wave-cooperative kernel), the strays walk every shell and end in the brute-force scan.  Mean kNN distance and both
    filters stay bit-exact vs the oracle."""
    rng = np.random.default_rng(11)
    sheet = np.stack([rng.uniform(-4, 4, 6000), rng.normal(-1.6, 0.01, 6000), rng.uniform(-20, -8, 6000)], 1)
    halo = np.stack([rng.uniform(-30, 30, 400), rng.uniform(-10, 10, 400), rng.uniform(-60, -2, 400)], 1)
    strays = np.float64([[400, 0, -10], [-350, 90, -700], [0, 0, 5000]])
    pts = np.concatenate([sheet, halo, strays]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    col = rng.integers(0, 256, (len(pts), 3), dtype=np.uint8)
    n1, n2 = _o3d_compare(pcl, pts, col, nb=20)
    assert 0 < n2 <= n1 < len(pts)
    _o3d_compare(pcl, pts, col, k=16, ratio=1.5, nb=3, radius=2.0)


def test_threshold_edges_golden(pcl, mini):
    """the reference's answers (captured by make_golden.py) for cuts float32 cannot represent, points at the float32 neighbours of a
    cut and plane residuals within rounding of the threshold: the HIP filters take the same side of every comparison"""
    e, ec = mini["edge_in"], mini["edge_col"]
    for tag, cut in (("70", 7.0), ("71", 7.1), ("69", 6.999999)):
        p, c = pcl.remove_from_to(e, ec, 2, 0.0, cut)
        _eq(p, mini[f"edge_rft_{tag}_pts"]); _eq(c, mini[f"edge_rft_{tag}_col"])
    _eq(pcl.threshold_complete(e, ec, 2, 35.0)[0], mini["edge_thr35_pts"])
    _eq(pcl.threshold_complete(e, ec, 2, 0.1)[0], mini["edge_thr01_pts"])
    p, c, _, _, coeff = pcl.remove_noise_by_fitting_plane(mini["edge_plane_in"], mini["edge_plane_col"], axis=1, threshold=0.75)
    _eq(p, mini["edge_plane_pts"])


def test_radius_filter_fast_accept_stays_exact(pcl):
    """the radius filter accepts a query from the counts of its 3 x 3 x 3 cells alone when those cells are bounded; clouds that
    break that premise -- a grid clamped in y (extent / cell > 64 layers), strays far outside, points AT the grid faces, dense
    clusters of just nb / nb + 1 points, and the radius filter called directly on a cloud with clamped layers -- must still
    match the oracle point for point"""
    rng = np.random.default_rng(5)
    sheet = np.stack([rng.uniform(-6, 6, 20000), rng.normal(-1.5, 0.02, 20000), rng.uniform(-14, -8, 20000)], 1)
    tall = np.stack([rng.uniform(-1, 1, 3000), rng.uniform(-12, 12, 3000), rng.uniform(-10, -9, 3000)], 1)      # 24 m of y: clamps
    strays = np.float64([[300, 0, -10], [0, 250, -10], [0, 0, -4000], [-300, -250, 2000]])
    # clusters of exactly nb and nb + 1 points inside one cell-sized blob (the count must EXCEED nb)
    c80 = np.float64([20.0, 5.0, -30.0]) + rng.uniform(-0.04, 0.04, (80, 3))
    c81 = np.float64([-20.0, 5.0, -30.0]) + rng.uniform(-0.04, 0.04, (81, 3))
    pts = np.concatenate([sheet, tall, strays, c80, c81]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    col = rng.integers(0, 256, (len(pts), 3), dtype=np.uint8)
    for nb, radius in ((80, 0.5), (79, 0.5), (10, 0.25), (300, 1.0)):
        keep = oracle_o3d.radius_outlier_mask(pts, nb, radius)
        p, c = pcl.radius_outlier_removal(pts, col, nb, radius)
        assert len(p) == int(keep.sum()), (nb, radius, len(p), int(keep.sum()))
        _eq(p, pts[keep]); _eq(c, col[keep])
        assert 0 < len(p) < len(pts)
    # the sheet alone: every face layer bounded -> the fast accept decides nearly every point; same answer
    sh = sheet.astype(np.float32)
    keep = oracle_o3d.radius_outlier_mask(sh, 80, 0.5)
    p, _ = pcl.radius_outlier_removal(sh, col[:len(sh)], 80, 0.5)
    _eq(p, sh[keep])


def test_o3d_knn_mean_distance_exact(pcl, mini):
    """the per-point mean kNN distance itself is bit-exact vs the oracle's canonical float64 definition."""
    from semantic_depth_amd.engine import _ptr
    from semantic_depth_amd import _lib as L
    e = pcl._eng()
    pts = mini["zcut_pts"]
    n = len(pts)
    d_pts = dev(pts)
    o = torch.empty_like(d_pts)
    n_out = torch.zeros(1, dtype=torch.int32, device="cuda")
    md = torch.empty(n, dtype=torch.float64, device="cuda")
    st = e.lib.sd_o3d_statistical_outlier_removal(e.h, _ptr(d_pts), None, n, 10, 0.5, _ptr(o), None, _ptr(n_out), _ptr(md), e._stream())
    L.check(e.lib, e.h, st, "sor")
    assert np.array_equal(md.cpu().numpy(), oracle_o3d.knn_mean_distance(pts, 10))


def test_tail_full_size_golden_and_oracle(pcl, golden_dir):
    """512x1024 Appendix-F scene through Engine.road_width: digests captured from the reference's pcl (no Open3D),
    then the full chain incl. the Open3D filters vs the oracle."""
    g = json.load(open(os.path.join(golden_dir, "pcl_full.json")))
    sc = g["scene"]
    dp, road, fence, frame, cam = pipeline.synthetic_scene(sc["h"], sc["w"], seed=sc["seed"], f=sc["f"])
    e = pcl._eng()
    pp = e.post_process(dev(dp[None]))
    fz = e.fuse_backproject(pp, dev(road[None].astype(np.uint8)), dev(fence[None].astype(np.uint8)), dev(frame[None]), [Camera(**cam)])
    res, fin, nfin = e.road_width(fz["road_xyz"], fz["n_road"], RoadWidthParams(use_o3d=False), want_final=True)
    rec = e.records(res)[0]
    assert (rec["n_road"], rec["n_zcut"], rec["n_mad_y"], rec["n_mad_x"], rec["n_plane"]) == \
        (g["n_road"], g["n_zcut"], g["n_mad_y"], g["n_mad_x"], g["n_plane"])
    assert checksum(fin[0, :int(nfin[0])].cpu().numpy()) == g["plane_checksum"]
    assert rec["found"] == 1 and float(rec["x_left"]) == g["x_left"] and float(rec["x_right"]) == g["x_right"]
    assert float(rec["width"]) == g["dist_rw"]
    assert [float(v) for v in rec["left_pt"]] == g["left_pt"] and [float(v) for v in rec["right_pt"]] == g["right_pt"]
    for i, k in enumerate(("Cx", "Cy", "Cz", "C")):
        assert abs(rec["plane"][i] - g["plane_coeff"][k]) <= 1e-9 * max(1.0, abs(g["plane_coeff"][k]))
    # full chain with the Open3D filters vs the oracle
    ref = pipeline.frame_tail(dp, road, fence, frame, cam)["rw"]
    res, fin, nfin = e.road_width(fz["road_xyz"], fz["n_road"], RoadWidthParams(), want_final=True)
    rec = e.records(res)[0]
    assert (rec["n_sor"], rec["n_ror"]) == (ref["n_sor"], ref["n_ror"])
    assert np.array_equal(fin[0, :int(nfin[0])].cpu().numpy().astype(np.float64), ref["points"])
    assert float(rec["width"]) == ref["width"] and float(rec["x_left"]) == ref["x_left"]


def test_tail_batch_and_degenerate_frames(pcl):
    """B=2: one normal frame and one frame with an empty road mask -> found=0, counts 0, no crash; and a frame whose
    depth window is empty."""
    e = pcl._eng()
    dp, road, fence, frame, cam = pipeline.synthetic_scene(512, 1024, seed=21, f=1000.0)
    pp = e.post_process(dev(np.stack([dp, dp])))
    masks = np.stack([road, np.zeros_like(road)]).astype(np.uint8)
    fz = e.fuse_backproject(pp, dev(masks), dev(masks), dev(np.stack([frame, frame])), [Camera(**cam)] * 2)
    rec = e.records(e.road_width(fz["road_xyz"], fz["n_road"], RoadWidthParams()))
    ref = pipeline.frame_tail(dp, road, fence, frame, cam)["rw"]
    assert rec[0]["found"] == 1 and float(rec[0]["width"]) == ref["width"] and rec[0]["n_ror"] == ref["n_ror"]
    assert rec[1]["found"] == 0 and rec[1]["n_road"] == 0 and rec[1]["n_ror"] == 0 and np.isnan(rec[1]["width"])
    rec = e.records(e.road_width(fz["road_xyz"], fz["n_road"], RoadWidthParams(depth=500.0)))
    assert rec[0]["found"] == 0 and rec[0]["n_ror"] == ref["n_ror"]


def test_extract_pcls_and_mean_bit_exact(pcl, mini):
    """pcl.extract_pcls: the split mean reproduces np.mean of the float32 column bit for bit (chunked pairwise sum)."""
    a, ac, b, bc = pcl.extract_pcls(mini["thr_pts"], mini["thr_col"])
    _eq(a, mini["split_left"]); _eq(ac, mini["split_left_col"]); _eq(b, mini["split_right"]); _eq(bc, mini["split_right_col"])
    from semantic_depth_amd.engine import _ptr
    from semantic_depth_amd import _lib as L
    e = pcl._eng()
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 129, 8191, 8192, 8193, 20011, 70001, 300007):
        pts = (rng.standard_normal((n, 3)) * 5).astype(np.float32)
        d = dev(pts)
        o1, o2 = torch.empty_like(d), torch.empty_like(d)
        n1 = torch.zeros(1, dtype=torch.int32, device="cuda"); n2 = torch.zeros(1, dtype=torch.int32, device="cuda")
        mean = torch.zeros(1, dtype=torch.float32, device="cuda")
        for axis in (0, 2):
            st = e.lib.sd_pcl_extract_pcls(e.h, _ptr(d), None, n, axis, _ptr(o1), None, _ptr(n1), _ptr(o2), None, _ptr(n2), _ptr(mean), e._stream())
            L.check(e.lib, e.h, st, "extract")
            ref = np.mean(pts[:, axis])
            assert np.float32(mean.item()) == ref, (n, axis, mean.item(), ref)
            assert int(n1) == int((pts[:, axis] < ref).sum()) and int(n2) == int((pts[:, axis] > ref).sum())


def test_fence_chain_golden_and_oracle(pcl, mini, golden_dir):
    """fence chain + fence-to-fence: mini scene vs the reference-captured arrays, 512x1024 fence scene vs digests."""
    from oracle import pipeline as op
    from semantic_depth_amd.engine import FenceParams
    e = pcl._eng()
    # --- mini: through the pcl mirror, stage by stage
    p, c = pcl.remove_noise_by_mad(mini["fence3d"], mini["fence_rgb"], 1, 5.0)
    _eq(p, mini["fc_mad_y"])
    p, c = pcl.threshold_complete(p, c, 2, 35.0)
    _eq(p, mini["fc_thr"])
    l, lc, r, rc = pcl.extract_pcls(p, c)
    _eq(l, mini["fc_left"]); _eq(r, mini["fc_right"])
    l, lc = pcl.remove_noise_by_mad(l, lc, 0, 5.0)
    l, lc, _, _, cl = pcl.remove_noise_by_fitting_plane(l, lc, axis=0, threshold=1.0)
    r, rc = pcl.remove_noise_by_mad(r, rc, 0, 1.0)
    r, rc, _, _, cr = pcl.remove_noise_by_fitting_plane(r, rc, axis=0, threshold=1.0)
    _eq(l, mini["fc_left_final"]); _eq(r, mini["fc_right_final"])
    assert np.allclose([cl[k] for k in ("Cx", "Cy", "Cz", "C")], mini["fc_plane_left"], rtol=1e-8, atol=1e-10)
    # --- full size, batched device path
    g = json.load(open(os.path.join(golden_dir, "pcl_full.json")))["fence"]
    sc = g["scene"]
    dp, road, fence, frame, cam = op.synthetic_scene(sc["h"], sc["w"], seed=sc["seed"], f=sc["f"], fences=True)
    pp = e.post_process(dev(np.stack([dp, dp])))
    masks_r = np.stack([road, road]).astype(np.uint8)
    masks_f = np.stack([fence, np.zeros_like(fence)]).astype(np.uint8)      # frame 1 has no fence at all
    fz = e.fuse_backproject(pp, dev(masks_r), dev(masks_f), dev(np.stack([frame, frame])), [Camera(**cam)] * 2)
    rw = e.road_width(fz["road_xyz"], fz["n_road"], RoadWidthParams())
    f2f = e.f2f_records(e.fence_to_fence(fz["fence_xyz"], fz["n_fence"], rw, FenceParams()))
    c = f2f[0]["counts"]
    assert tuple(c) == (g["n_fence"], g["n_mad_y"], g["n_thr"], g["n_left"], g["n_right"], g["n_left_final"], g["n_right_final"])
    assert np.allclose(f2f[0]["plane_left"], g["plane_left"], rtol=1e-8, atol=1e-10)
    assert np.allclose(f2f[0]["plane_right"], g["plane_right"], rtol=1e-8, atol=1e-10)
    # oracle for the distance (the reference's own plane-intersection cannot run on numpy 2: closed form)
    ref_fz = op.frame_tail(dp, road, fence, frame, cam)
    ft = op.fence_tail(ref_fz["fence3d"], ref_fz["fence_rgb"], ref_fz["rw"]["plane"])
    assert f2f[0]["ok"] == 1 and abs(f2f[0]["dist"] - ft["dist"]) <= 1e-9 * ft["dist"]
    assert np.allclose(f2f[0]["left_pt"], ft["left_pt"], rtol=1e-9, atol=1e-9)
    assert f2f[1]["ok"] == 0 and f2f[1]["counts"][0] == 0


def test_o3d_filters_ties_and_degenerate_clouds(pcl):
    """distance ties everywhere (a regular lattice: every point has 6 / 12 / 8 neighbours at exactly equal distances, so the k-th
    neighbour is one of several equidistant candidates), heavy duplication (k-th distance 0 for most points), a cloud on a line
    (one grid axis degenerate) and on a single point; both filters against the oracle, bit for bit"""
    rng = np.random.default_rng(3)
    g = np.arange(12, dtype=np.float32) * np.float32(0.125)               # exact in binary: the ties are exact ties
    lattice = np.stack(np.meshgrid(g, g - 3, -g - 9, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    lattice = lattice[rng.permutation(len(lattice))]
    col = rng.integers(0, 256, (len(lattice), 3), dtype=np.uint8)
    n1, n2 = _o3d_compare(pcl, lattice, col, k=10, ratio=0.5, nb=6, radius=0.13)       # radius just above one spacing
    assert n1 > 0
    _o3d_compare(pcl, lattice, col, k=7, ratio=0.1, nb=18, radius=0.1768)              # sqrt(2) * 0.125 = 0.17678: face diagonals on the boundary
    three = np.float32([[0, 0, -10], [0.5, 0, -10], [0, 0.25, -11]])
    dup = three[rng.integers(0, 3, 600)]
    _o3d_compare(pcl, dup, col[:600], k=10, ratio=0.5, nb=50, radius=0.3)
    line = np.stack([np.linspace(-5, 5, 900), np.zeros(900), np.full(900, -12.0)], 1).astype(np.float32)
    _o3d_compare(pcl, line, col[:900], k=10, ratio=1.0, nb=20, radius=0.2)
    point = np.repeat(np.float32([[1.5, -2.0, -30.0]]), 40, axis=0)
    _o3d_compare(pcl, point, col[:40], k=10, ratio=0.5, nb=5, radius=0.01)
