"""Oracle: Open3D legacy outlier filters used at semantic_depth.py:227-245 (seq:196-223).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
[UPSTREAM, PARITY UNPINNED]  Open3D is neither vendored nor listed in requirements.txt; this restates
the published algorithm of Open3D's legacy ``statistical_outlier_removal`` / ``radius_outlier_removal``
(PointCloud.cpp, RemoveStatisticalOutliers / RemoveRadiusOutliers, 0.4-0.7 era) from memory:

  statistical(nb_neighbors=k, std_ratio=r)
      for every point: kNN with k neighbours INCLUDING the point itself (FLANN returns squared
      distances, sorted ascending); mean_i = (sum of sqrt(d2)) / k.
      cloud_mean = (sum of the mean_i that are > 0) / N ;  std = sqrt(sum_{mean_i>0} (mean_i-cloud_mean)^2 / (N-1))
      keep i  iff  mean_i > 0  and  mean_i < cloud_mean + r*std
  radius(nb_points=n, radius=R)
      keep i  iff  #{j : d2(i,j) < R*R}  (self included) > n

Arithmetic is float64 (Vector3dVector stores doubles).  To make the oracle and the HIP path comparable
bit for bit the squared distance is DEFINED here as ((dx*dx + dy*dy) + dz*dz) with no fused multiply-add,
neighbour distances are summed in ascending order, and the two cloud-wide sums run in index order.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def _d2(a, b):
    dx, dy, dz = a[..., 0] - b[..., 0], a[..., 1] - b[..., 1], a[..., 2] - b[..., 2]
    return (dx * dx + dy * dy) + dz * dz


def knn_mean_distance(points: np.ndarray, k: int) -> np.ndarray:
    """mean of the k smallest Euclidean distances (self included) for every point; float64 (N,)."""
    pts = np.asarray(points, np.float64)
    n = pts.shape[0]
    if n == 0:
        return np.zeros(0)
    kk = min(n, k)
    if n <= 2048:
        d2 = _d2(pts[:, None, :], pts[None, :, :])
        d2 = np.sort(d2, axis=1)[:, :kk]
    else:
        # candidates from the tree (a few extra to be immune to last-ulp ordering), exact recompute here
        kq = min(n, k + 4)
        _, idx = cKDTree(pts).query(pts, k=kq, workers=-1)
        d2 = np.sort(_d2(pts[:, None, :], pts[idx]), axis=1)[:, :kk]
    d = np.sqrt(d2)
    acc = np.zeros(n)
    for j in range(kk):          # ascending order, sequential adds
        acc = acc + d[:, j]
    return acc / kk


def statistical_outlier_mask(points: np.ndarray, nb_neighbors: int = 10, std_ratio: float = 0.5):
    mean_d = knn_mean_distance(points, nb_neighbors)
    n = mean_d.shape[0]
    if n == 0:
        return np.zeros(0, bool), mean_d, 0.0, 0.0
    pos = mean_d > 0
    cloud_mean = (np.cumsum(np.where(pos, mean_d, 0.0))[-1]) / n
    dev = np.where(pos, (mean_d - cloud_mean) * (mean_d - cloud_mean), 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        std = np.sqrt(np.float64(np.cumsum(dev)[-1]) / np.float64(n - 1))
    thr = cloud_mean + std_ratio * std
    return pos & (mean_d < thr), mean_d, cloud_mean, std


def radius_count(points: np.ndarray, radius: float) -> np.ndarray:
    pts = np.asarray(points, np.float64)
    n = pts.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    r2 = radius * radius
    if n <= 2048:
        return (_d2(pts[:, None, :], pts[None, :, :]) < r2).sum(1)
    tree = cKDTree(pts)
    # The tree's own rounding of d2 may differ from the canonical one in the last ulp, so count with a
    # slightly deflated and a slightly inflated ball; only where the two disagree is the strict test
    # re-evaluated with the canonical d2.
    lo = tree.query_ball_point(pts, radius * (1 - 1e-9), return_length=True, workers=-1)
    hi = tree.query_ball_point(pts, radius * (1 + 1e-9), return_length=True, workers=-1)
    out = np.asarray(lo, np.int64)
    for i in np.nonzero(lo != hi)[0]:
        nb = tree.query_ball_point(pts[i], radius * (1 + 1e-9))
        out[i] = int((_d2(pts[i][None, :], pts[nb]) < r2).sum())
    return out


def radius_outlier_mask(points: np.ndarray, nb_points: int = 80, radius: float = 0.5):
    return radius_count(points, radius) > nb_points


def statistical_outlier_removal(points, colors, nb_neighbors=10, std_ratio=0.5):
    keep = statistical_outlier_mask(points, nb_neighbors, std_ratio)[0]
    return np.asarray(points, np.float64)[keep], np.asarray(colors, np.float64)[keep]


def radius_outlier_removal(points, colors, nb_points=80, radius=0.5):
    keep = radius_outlier_mask(points, nb_points, radius)
    return np.asarray(points, np.float64)[keep], np.asarray(colors, np.float64)[keep]
