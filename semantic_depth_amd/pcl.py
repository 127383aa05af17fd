"""Drop-in for the reference's ``semantic_depth_lib/pcl.py`` — same function names, argument meaning, return
shapes and error behaviour, executed by the HIP kernels of libsemdepth (csrc/pcl.hip).

Arrays go in and come out as numpy (like the reference); they are staged through device memory per call.
The batched, device-resident path the pipeline uses is ``Engine.road_width``.  Differences kept on purpose:
  * remove_noise_by_fitting_plane builds the two visualisation arrays (plane3D, colors_plane, pcl.py:104-110,121-124 ...)
    on the host from the GPU's plane coefficients — they feed only the PLY output.
  * get_end_points_of_road returns the FIRST min-x / max-x row as a (1,3) array (the reference returns all tied
    rows and then only ever reads [0]).
The O(1) helpers at the bottom are host arithmetic on one or two points, as in the reference.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .engine import Engine, RW_DTYPE, _ptr

_engine: Engine | None = None


def set_engine(engine: Engine):
    """use an existing Engine (its o3d scratch bounds the cloud size: max_batch*H*W points)."""
    global _engine
    _engine = engine


def _eng() -> Engine:
    """the Engine the filters run on: one set with set_engine(), else the LARGEST (most points per call) of the engines the api
    classes already share, else a 512 x 1024 engine of its own.  (Not pinned when it comes from the api registry: a larger engine
    registered later takes over.)"""
    global _engine
    if _engine is not None:
        return _engine
    from . import api
    shared = [e for per in api._engines.values() for e in per.values() if getattr(e, "h", None)]
    if shared:
        return max(shared, key=lambda e: e.max_batch * e.cap)
    _engine = api.shared_engine(512, 1024)
    return _engine


def _up(points3D, colors):
    e = _eng()
    pts = np.ascontiguousarray(points3D, dtype=np.float32)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError("points3D must be (N,3)")
    d_pts = torch.from_numpy(pts).to(e.device) if len(pts) else torch.empty((1, 3), dtype=torch.float32, device=e.device)
    d_col = None
    if colors is not None:
        col = np.ascontiguousarray(colors)
        if col.dtype != np.uint8:
            col = col.astype(np.uint8)
        d_col = torch.from_numpy(col).to(e.device) if len(col) else torch.empty((1, 3), dtype=torch.uint8, device=e.device)
    return e, pts, d_pts, d_col


def _down(e, n_dev, o_pts, o_col, like_pts, like_col):
    n = int(n_dev.item())
    pts = o_pts[:n].cpu().numpy().astype(np.asarray(like_pts).dtype, copy=False)
    col = None
    if o_col is not None:
        col = o_col[:n].cpu().numpy().astype(np.asarray(like_col).dtype, copy=False)
    return pts, col


def _filter(fn_name, points3D, colors, *args, extra=None):
    e, pts, d_pts, d_col = _up(points3D, colors)
    n = len(pts)
    o_pts = torch.empty_like(d_pts)
    o_col = torch.empty_like(d_col) if d_col is not None else None
    n_out = torch.zeros(1, dtype=torch.int32, device=e.device)
    fn = getattr(e.lib, fn_name)
    tail = [] if extra is None else [_ptr(extra)]
    st = fn(e.h, _ptr(d_pts), _ptr(d_col), n, *args, _ptr(o_pts), _ptr(o_col), _ptr(n_out), *tail, e._stream())
    L.check(e.lib, e.h, st, fn_name)
    return _down(e, n_out, o_pts, o_col, points3D, colors)


# ------------------------------------------------------------------------------------------------
def remove_from_to(points3D, colors, axis, from_meter, to_meter):
    """pcl.py:30-43.  Keeps rows with coord[axis] < -to_meter (from_meter is ignored, as in the reference).
    Raises ValueError on an empty cloud like the reference's min() does."""
    if np.asarray(points3D).shape[0] == 0:
        raise ValueError("min() arg is an empty sequence")
    return _filter("sd_pcl_remove_from_to", points3D, colors, int(axis), float(to_meter))


def remove_noise_by_mad(points3D, colors, axis, threshold=15.0):
    """pcl.py:46-73 (+ mad, :76-81)."""
    return _filter("sd_pcl_remove_noise_by_mad", points3D, colors, int(axis), float(threshold),
                   extra=torch.empty(2, dtype=torch.float32, device=_eng().device))


def mad(points1D):
    """pcl.py:76-81 -> (abs_diffs, mad).  The two medians run on the GPU."""
    v = np.ascontiguousarray(points1D, dtype=np.float32)
    pts = np.zeros((len(v), 3), np.float32)
    pts[:, 0] = v
    e, _, d_pts, _ = _up(pts, None)
    stats = torch.empty(2, dtype=torch.float32, device=e.device)
    o = torch.empty_like(d_pts)
    n_out = torch.zeros(1, dtype=torch.int32, device=e.device)
    st = e.lib.sd_pcl_remove_noise_by_mad(e.h, _ptr(d_pts), None, len(v), 0, 1e30, _ptr(o), None, _ptr(n_out), _ptr(stats), e._stream())
    L.check(e.lib, e.h, st, "sd_pcl_remove_noise_by_mad")
    med, m = stats.cpu().numpy()
    return abs(v - med), m


def plane_grid(points3D, coefficients, axis, plane_color=[255, 255, 255], grid_size=0.05):
    """the visualisation arrays of pcl.py:104-124 / :141-160 / :176-195: a ``grid_size`` lattice over the bounding box of the
    INPUT cloud in the two in-plane coordinates, lifted onto the fitted plane.  Host numpy (it feeds only the PLY output)."""
    points3D = np.asarray(points3D)
    ia, ib = [(1, 2), (0, 2), (0, 1)][axis]
    ca, cb = [("Cy", "Cz"), ("Cx", "Cz"), ("Cx", "Cy")][axis]
    A, Bv = np.meshgrid(np.arange(np.amin(points3D[:, ia]), np.amax(points3D[:, ia]), grid_size),
                        np.arange(np.amin(points3D[:, ib]), np.amax(points3D[:, ib]), grid_size))
    Cn = coefficients[ca] * A + coefficients[cb] * Bv + coefficients["C"]
    cols = [None, None, None]
    cols[axis], cols[ia], cols[ib] = Cn.flatten(), A.flatten(), Bv.flatten()
    plane3D = np.c_[cols[0], cols[1], cols[2]]
    return plane3D, np.ones(plane3D.shape) * plane_color


def remove_noise_by_fitting_plane(points3D, colors, axis=0, threshold=1.0, plane_color=[255, 255, 255], with_plane3D=True):
    """pcl.py:84-209.  Returns (points3D', colors', plane3D, colors_plane, coefficients); the two visualisation arrays are
    built on the host (``with_plane3D=False`` returns None for them)."""
    coeff = torch.empty(4, dtype=torch.float64, device=_eng().device)
    pts, col = _filter("sd_pcl_remove_noise_by_fitting_plane", points3D, colors, int(axis), float(threshold), extra=coeff)
    c = coeff.cpu().numpy()
    coefficients = {"Cx": c[0], "Cy": c[1], "Cz": c[2], "C": c[3]}
    plane3D = colors_plane = None
    if with_plane3D and len(np.asarray(points3D)):
        plane3D, colors_plane = plane_grid(points3D, coefficients, int(axis), plane_color)
    return pts, col, plane3D, colors_plane, coefficients


def threshold_complete(points3D, colors, axis, threshold=15.0):
    """pcl.py:240-250."""
    return _filter("sd_pcl_threshold_complete", points3D, colors, int(axis), float(threshold))


def extract_pcls(points3D, colors, axis=0):
    """pcl.py:253-268: (left, left_colors, right, right_colors), split at np.mean of the ``axis`` column."""
    e, pts, d_pts, d_col = _up(points3D, colors)
    n = len(pts)
    outs = [torch.empty_like(d_pts), torch.empty_like(d_pts)]
    cols = [torch.empty_like(d_col) if d_col is not None else None for _ in range(2)]
    ns = [torch.zeros(1, dtype=torch.int32, device=e.device) for _ in range(2)]
    st = e.lib.sd_pcl_extract_pcls(e.h, _ptr(d_pts), _ptr(d_col), n, int(axis), _ptr(outs[0]), _ptr(cols[0]), _ptr(ns[0]),
                                   _ptr(outs[1]), _ptr(cols[1]), _ptr(ns[1]), None, e._stream())
    L.check(e.lib, e.h, st, "sd_pcl_extract_pcls")
    l, lc = _down(e, ns[0], outs[0], cols[0], points3D, colors)
    r, rc = _down(e, ns[1], outs[1], cols[1], points3D, colors)
    return l, lc, r, rc


def get_end_points_of_road(points3D, depth):
    """pcl.py:271-313.  (None, None) when no point lies in the +-0.05 depth window."""
    e, pts, d_pts, _ = _up(points3D, None)
    res = torch.zeros(RW_DTYPE.itemsize, dtype=torch.uint8, device=e.device)
    st = e.lib.sd_pcl_get_end_points_of_road(e.h, _ptr(d_pts), len(pts), float(depth), 0.05, _ptr(res), e._stream())
    L.check(e.lib, e.h, st, "sd_pcl_get_end_points_of_road")
    r = res.cpu().numpy().view(RW_DTYPE)[0]
    if not r["found"]:
        return None, None
    dt = np.asarray(points3D).dtype
    return r["left_pt"].astype(dt)[None, :], r["right_pt"].astype(dt)[None, :]


def statistical_outlier_removal(points3D, colors, nb_neighbors=10, std_ratio=0.5):
    """Open3D legacy statistical_outlier_removal + select_down_sample (semantic_depth.py:234-236)."""
    return _sor(points3D, colors, nb_neighbors, std_ratio)


def _sor(points3D, colors, nb_neighbors, std_ratio):
    e, pts, d_pts, d_col = _up(points3D, colors)
    o_pts = torch.empty_like(d_pts)
    o_col = torch.empty_like(d_col) if d_col is not None else None
    n_out = torch.zeros(1, dtype=torch.int32, device=e.device)
    st = e.lib.sd_o3d_statistical_outlier_removal(e.h, _ptr(d_pts), _ptr(d_col), len(pts), int(nb_neighbors), float(std_ratio),
                                                  _ptr(o_pts), _ptr(o_col), _ptr(n_out), None, e._stream())
    L.check(e.lib, e.h, st, "sd_o3d_statistical_outlier_removal")
    return _down(e, n_out, o_pts, o_col, points3D, colors)


def radius_outlier_removal(points3D, colors, nb_points=80, radius=0.5):
    """Open3D legacy radius_outlier_removal + select_down_sample (semantic_depth.py:238-241)."""
    return _filter("sd_o3d_radius_outlier_removal", points3D, colors, int(nb_points), float(radius))


# ------------------------------------------------------------------------------------------------ O(1) helpers
def planes_intersection_at_certain_depth(C_p1, C_p2, z):
    """pcl.py:212-237 (2x2 solve at z = -depth; returns shape (1,3) float64)."""
    z = -z
    A = np.array([[C_p1["Cx"], C_p1["Cy"]], [C_p2["Cx"], C_p2["Cy"]]], np.float64)
    B = np.array([-(C_p1["Cz"] * z + C_p1["C"]), -(C_p2["Cz"] * z + C_p2["C"])], np.float64)
    X = np.linalg.inv(A) @ B
    return np.array([[X[0], X[1], z]], np.float64)


def compute_distance_in_3D(pt3D_A, pt3D_B):
    """pcl.py:316-318."""
    return np.linalg.norm(pt3D_A - pt3D_B)


def create_3Dline_from_3Dpoints(left_pt, right_pt, color):
    """pcl.py:321-331 (mutates both end points: y += 0.01)."""
    left_pt[0][1] += 0.01
    right_pt[0][1] += 0.01
    v = right_pt - left_pt
    t = np.arange(0.0, 1.0, 0.001)
    line = np.concatenate([left_pt] + [left_pt + ti * v for ti in t], axis=0)
    return line, np.ones(line.shape) * color
