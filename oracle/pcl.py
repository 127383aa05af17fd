"""Oracle: numpy restatement of the reference's hand-made point-cloud library.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED: tests/test_oracle_golden.py checks every
function here against outputs captured from the reference's own semantic_depth_lib/pcl.py
(tests/golden/make_golden.py, run in the build container where /root/reference is mounted).

Each function cites the reference lines it restates.  All of them return NEW arrays and keep the input
row order, like the reference.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg


def remove_from_to(points3d, colors, axis, from_meter, to_meter):
    """pcl.py:30-43.  Keeps rows whose ``axis`` coordinate is < -to_meter.  ``from_meter`` is ignored by
    the reference; the reference also calls builtin min() on the column first, which raises ValueError
    on an empty cloud — restated here so the edge case is the same."""
    if points3d.shape[0] == 0:
        raise ValueError("min() arg is an empty sequence")
    keep = points3d[:, axis] < -to_meter
    return points3d[keep], colors[keep]


def mad(values):
    """pcl.py:76-81: (|v - median(v)|, median of that)."""
    med = np.median(values)
    dev = abs(values - med)
    return dev, np.median(dev)


def mad_penalty(values):
    """pcl.py:61-63: 0.6745 * |v - med| / MAD (inf/nan when MAD == 0)."""
    dev, m = mad(values)
    with np.errstate(divide="ignore", invalid="ignore"):
        return 0.6745 * dev / m


def remove_noise_by_mad(points3d, colors, axis, threshold=15.0):
    """pcl.py:46-73: keep rows with penalty < threshold."""
    keep = mad_penalty(points3d[:, axis]) < threshold
    return points3d[keep], colors[keep]


_PLANE_COLS = {0: (1, 2, 0), 1: (0, 2, 1), 2: (0, 1, 2)}   # (u, v, dependent) per ``axis``


def fit_plane(points3d, axis):
    """pcl.py:113-115 / :151-154 / :187-190: least squares dep = C0*u + C1*v + C2 on float64 columns
    via scipy.linalg.lstsq."""
    u, v, dep = _PLANE_COLS[axis]
    A = np.c_[points3d[:, u], points3d[:, v], np.ones(points3d.shape[0])]
    C, _, _, _ = scipy.linalg.lstsq(A, points3d[:, dep])
    return C


def plane_coefficients(C, axis):
    """pcl.py:130, :168, :204: reorder so that Cx*x + Cy*y + Cz*z + C = 0."""
    if axis == 0:
        return {"Cx": -1.0, "Cy": C[0], "Cz": C[1], "C": C[2]}
    if axis == 1:
        return {"Cx": C[0], "Cy": -1.0, "Cz": C[1], "C": C[2]}
    return {"Cx": C[0], "Cy": C[1], "Cz": -1.0, "C": C[2]}


def remove_noise_by_fitting_plane(points3d, colors, axis=0, threshold=1.0):
    """pcl.py:84-209 without the visualisation grid (plane3D / colors_plane, :104-110, :121-124 and the
    two sibling branches), which does not feed the road-width scalar.
    Returns (points', colors', coefficients dict)."""
    u, v, dep = _PLANE_COLS[axis]
    C = fit_plane(points3d, axis)
    resid = C[0] * points3d[:, u] + C[1] * points3d[:, v] - points3d[:, dep] + C[2]
    keep = abs(resid) < threshold
    return points3d[keep], colors[keep], plane_coefficients(C, axis)


def planes_intersection_at_certain_depth(c1, c2, z):
    """pcl.py:212-237: solve the 2x2 system at z = -depth.  (The reference builds a ragged np.array at
    :235 that numpy >= 1.24 rejects; the value is the closed-form solution below, shape (1,3).)"""
    z = -z
    A = np.array([[c1["Cx"], c1["Cy"]], [c2["Cx"], c2["Cy"]]], np.float64)
    B = np.array([-(c1["Cz"] * z + c1["C"]), -(c2["Cz"] * z + c2["C"])], np.float64)
    X = np.linalg.inv(A) @ B
    return np.array([[X[0], X[1], z]], np.float64)


def threshold_complete(points3d, colors, axis, threshold=15.0):
    """pcl.py:240-250: keep |coord| < threshold."""
    keep = abs(points3d[:, axis]) < threshold
    return points3d[keep], colors[keep]


def extract_pcls(points3d, colors, axis=0):
    """pcl.py:253-268: split at the mean of ``axis`` (strict <, strict >; points equal to the mean drop)."""
    col = points3d[:, axis]
    mean = np.mean(col)
    lo, hi = col < mean, col > mean
    return points3d[lo], colors[lo], points3d[hi], colors[hi]


def get_end_points_of_segment(segment):
    """pcl.py:293-313: rows with the minimum / maximum x (all ties, in row order) or (None, None)."""
    xs = segment[:, 0]
    if xs.size == 0:
        return None, None
    return segment[xs == np.amin(xs)], segment[xs == np.amax(xs)]


def get_end_points_of_road(points3d, depth):
    """pcl.py:271-290: window -(depth+0.05) < z < -(depth-0.05), then the segment's end points."""
    z = points3d[:, 2]
    sel = (z < -(depth - 0.05)) & (z > -(depth + 0.05))
    return get_end_points_of_segment(points3d[sel])


def compute_distance_in_3D(a, b):
    """pcl.py:316-318."""
    return np.linalg.norm(a - b)


def create_3Dline_from_3Dpoints(left_pt, right_pt, color):
    """pcl.py:321-331: lifts both end points by 0.01 in y IN PLACE, then 1 + 1000 samples left + t*v,
    t = arange(0, 1, 0.001)."""
    left_pt[0][1] += 0.01
    right_pt[0][1] += 0.01
    v = right_pt - left_pt
    t = np.arange(0.0, 1.0, 0.001)
    line = np.concatenate([left_pt] + [left_pt + ti * v for ti in t], axis=0)
    return line, np.ones(line.shape) * color
