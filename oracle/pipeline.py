"""Oracle: the per-frame chain  masks x disparity -> road cloud -> road width.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows FrameProcessor.process_frame,
semantic_depth.py:98-268 (working copy: semantic_depth_cityscapes_sequence.py:117-238).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import fusion, o3d, pcl


@dataclass
class RoadWidthParams:
    """All literals of the reference's call sites in one place (SURVEY §5 'Config / flags')."""
    depth: float = 10.0          # --depth, semantic_depth.py:754-756
    z_cut: float = 7.0           # semantic_depth.py:206
    mad_y: float = 15.0          # :209
    mad_x: float = 2.0           # :212
    plane_thr: float = 5.0       # :215-219
    sor_k: int = 10              # :234-235
    sor_ratio: float = 0.5
    ror_n: int = 80              # :238-239
    ror_r: float = 0.5
    depth_offset: float = 0.02   # :254-255
    use_o3d: bool = True


def road_width_tail(road3d: np.ndarray, road_rgb: np.ndarray, p: RoadWidthParams = RoadWidthParams()):
    """semantic_depth.py:203-259 on an (N,3) f32 road cloud.  Returns a dict with the kept count after
    every stage, the plane coefficients and the road-width record.  An empty road cloud is reported as
    found=False (the reference would raise in remove_from_to; seq:232-234 is the only guard it has)."""
    out = dict(n_in=int(road3d.shape[0]), found=False, width=float("nan"), x_left=float("nan"), x_right=float("nan"))
    if road3d.shape[0] == 0:
        out.update(n_zcut=0, n_mad_y=0, n_mad_x=0, n_plane=0, n_sor=0, n_ror=0, plane=None)
        return out
    pts, col = pcl.remove_from_to(road3d, road_rgb, 2, 0.0, p.z_cut)
    out["n_zcut"] = len(pts)
    pts, col = pcl.remove_noise_by_mad(pts, col, 1, p.mad_y)
    out["n_mad_y"] = len(pts)
    pts, col = pcl.remove_noise_by_mad(pts, col, 0, p.mad_x)
    out["n_mad_x"] = len(pts)
    if len(pts) >= 1:
        pts, col, coeff = pcl.remove_noise_by_fitting_plane(pts, col, axis=1, threshold=p.plane_thr)
    else:
        coeff = None
    out["n_plane"] = len(pts)
    out["plane"] = coeff
    if p.use_o3d:
        pts, col = o3d.statistical_outlier_removal(pts, col, p.sor_k, p.sor_ratio)
        out["n_sor"] = len(pts)
        pts, col = o3d.radius_outlier_removal(pts, col, p.ror_n, p.ror_r)
        out["n_ror"] = len(pts)
    else:
        pts = pts.astype(np.float64)
        out["n_sor"] = out["n_ror"] = len(pts)
    out["points"] = pts
    out["colors"] = col
    left, right = pcl.get_end_points_of_road(pts, p.depth - p.depth_offset)
    if left is not None and right is not None:
        out["found"] = True
        out["x_left"] = float(left[0][0])
        out["x_right"] = float(right[0][0])
        out["left_pt"] = left[0].copy()
        out["right_pt"] = right[0].copy()
        out["width"] = float(abs(left[0][0] - right[0][0]))      # semantic_depth.py:259
    return out


@dataclass
class FenceParams:
    """literals of semantic_depth.py:273-334"""
    depth: float = 10.0
    mad_y: float = 5.0
    z_max: float = 35.0
    mad_left: float = 5.0
    mad_right: float = 1.0
    plane_thr: float = 1.0


def fence_tail(fence3d: np.ndarray, fence_rgb: np.ndarray, road_plane: dict, p: FenceParams = FenceParams()):
    """semantic_depth.py:273-334 (seq:245-298): fence chain + fence-to-fence distance."""
    out = dict(n_fence=len(fence3d))
    pts, col = pcl.remove_noise_by_mad(fence3d, fence_rgb, 1, p.mad_y)
    out["n_mad_y"] = len(pts)
    pts, col = pcl.threshold_complete(pts, col, 2, p.z_max)
    out["n_thr"] = len(pts)
    l, lc, r, rc = pcl.extract_pcls(pts, col)
    out["n_left"], out["n_right"] = len(l), len(r)
    l, lc = pcl.remove_noise_by_mad(l, lc, 0, p.mad_left)
    l, lc, cl = pcl.remove_noise_by_fitting_plane(l, lc, axis=0, threshold=p.plane_thr)
    r, rc = pcl.remove_noise_by_mad(r, rc, 0, p.mad_right)
    r, rc, cr = pcl.remove_noise_by_fitting_plane(r, rc, axis=0, threshold=p.plane_thr)
    out.update(n_left_final=len(l), n_right_final=len(r), plane_left=cl, plane_right=cr, left=l, right=r)
    lp = pcl.planes_intersection_at_certain_depth(road_plane, cl, p.depth)
    rp = pcl.planes_intersection_at_certain_depth(road_plane, cr, p.depth)
    out.update(left_pt=lp[0], right_pt=rp[0], dist=float(pcl.compute_distance_in_3D(lp, rp)))
    return out


def frame_tail(disp_pair, road, fence, frame_bgr, cam, p: RoadWidthParams = RoadWidthParams()):
    """fusion + road-width tail for one frame. cam = dict(cx, cy, f, b, disp_mult)."""
    fz = fusion.fuse(disp_pair, road, fence, frame_bgr, **cam)
    rw = road_width_tail(fz["road3d"], fz["road_rgb"], p)
    fz["rw"] = rw
    return fz


def synthetic_scene(h=512, w=1024, seed=1234, cam_h=1.5, f=1000.0, b=1.0, half_width=3.5, noise=0.01,
                    fences=False):
    """SURVEY Appendix F ground-plane scene.  Returns (disp_pair (2,H,W) f32 in fraction of width, road bool,
    fence bool, frame u8 BGR, cam dict)."""
    rng = np.random.default_rng(seed)
    cx = w / 2 - 0.5 + 1.3
    cy = h / 2 - 0.5 - 3.7
    mult = float(w)
    v = np.arange(h, dtype=np.float64)[:, None] * np.ones((1, w))
    u = np.ones((h, 1)) * np.arange(w, dtype=np.float64)[None, :]
    d = np.where(v - cy > 4, b * (v - cy) / cam_h, 4 * b / cam_h)
    dn = d / mult
    dl = (dn * (1 + noise * rng.standard_normal((h, w)))).astype(np.float32)
    dr = np.fliplr(dn * (1 + noise * rng.standard_normal((h, w)))).astype(np.float32)
    X = (u - cx) * b / d
    road = (v - cy > 8) & (np.abs(X) < half_width)
    fence = np.zeros((h, w), bool)
    if fences:
        fence = (np.abs(X) >= 4.0) & (np.abs(X) < 4.3) & (v - cy > 8)
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    cam = dict(cx=cx, cy=cy, f=f, b=b, disp_mult=mult)
    return np.stack([dl, np.ascontiguousarray(dr)]), road, fence, frame, cam
