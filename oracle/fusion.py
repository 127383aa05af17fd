"""Oracle: flip-pair post-processing, disparity scaling, back-projection, mask gather.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Followed sources:
  * post_processing   semantic_depth.py:656-664 (numpy f64 because linspace is f64; m_disp is f32), cast f32 at :676
  * disparity scaling semantic_depth.py:109,145 (x original_width) / semantic_depth_cityscapes_sequence.py:105,146 (x 3800)
  * Q matrix          semantic_depth.py:691-694 (np.float32 entries)
  * reproject         cv2.reprojectImageTo3D at semantic_depth.py:696  [UPSTREAM OpenCV 4.0.0.21, requirements.txt:21]
  * mask gather       semantic_depth.py:183-187 (boolean indexing, row-major order), colours BGR->RGB at :161
"""
from __future__ import annotations

import numpy as np


def ramp_masks(w: int):
    """l_mask / r_mask of semantic_depth.py:660-662 for one image row (they do not depend on the row)."""
    l = np.linspace(0, 1, w)
    l_mask = 1.0 - np.clip(20 * (l - 0.05), 0, 1)
    r_mask = l_mask[::-1].copy()
    return l_mask, r_mask


def post_processing(disp: np.ndarray) -> np.ndarray:
    """disp: (2,H,W) f32 = net output for (frame, flipped frame), channel 0.  Returns f64 (H,W);
    the caller casts to f32 (semantic_depth.py:676)."""
    assert disp.ndim == 3 and disp.shape[0] == 2
    _, h, w = disp.shape
    left = disp[0]
    right = disp[1][:, ::-1]                    # fliplr of the flipped frame's disparity
    mean = 0.5 * (left + right)                 # stays in the input dtype (f32) like the reference
    l_mask, r_mask = ramp_masks(w)
    return r_mask[None, :] * left + l_mask[None, :] * right + (1.0 - l_mask - r_mask)[None, :] * mean


def make_Q(cx: float, cy: float, f: float, b: float) -> np.ndarray:
    """semantic_depth.py:691-694: y axis up, z axis toward the viewer (scene has negative z)."""
    return np.float32([[1, 0, 0, -cx],
                       [0, -1, 0, cy],
                       [0, 0, 0, -f],
                       [0, 0, 1 / b, 0]])


def reproject(disp: np.ndarray, Q: np.ndarray) -> np.ndarray:
    """cv2.reprojectImageTo3D(disp f32, Q) with handleMissingValues=False, ddepth=-1 (f32 out).

    [UPSTREAM, parity unpinned]  Restated from OpenCV 4.x calibration.cpp: Q is promoted to double;
    per pixel homg = Q * (x, y, d, 1) accumulated left to right in double; the three numerators are
    narrowed to float (``dptr[x] = Vec3d(homg)`` into a Vec3f), then ``dptr[x] /= homg[3]``: core/matx.hpp implements
    ``Vec<_Tp,cn> /= double`` as a multiply by ``ialpha = 1./alpha`` with a saturate_cast back to float, so each numerator
    is multiplied by the double reciprocal of W and narrowed again (NOT divided: the two differ by 1 ulp on rare pixels).
    d = 0 gives W = 0, ialpha = +-inf and +-inf / nan results exactly as IEEE does."""
    assert disp.dtype == np.float32 and disp.ndim == 2
    h, w = disp.shape
    q = Q.astype(np.float64)
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    d = disp.astype(np.float64)
    out = np.empty((h, w, 3), np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        W = ((q[3, 0] * x + q[3, 1] * y) + q[3, 2] * d) + q[3, 3]
        for i in range(3):
            num = ((q[i, 0] * x + q[i, 1] * y) + q[i, 2] * d) + q[i, 3]
            out[..., i] = (num.astype(np.float32).astype(np.float64) * (1.0 / W)).astype(np.float32)
    return out


def bgr_to_rgb(frame_u8: np.ndarray) -> np.ndarray:
    return frame_u8[..., ::-1].copy()


def mask_gather(points3d: np.ndarray, colors: np.ndarray, mask: np.ndarray):
    """points3D[mask], colors[mask] — row-major order of the True pixels."""
    return points3d[mask], colors[mask]


def fuse(disp_pair: np.ndarray, road: np.ndarray, fence: np.ndarray, frame_bgr: np.ndarray,
         cx: float, cy: float, f: float, b: float, disp_mult: float):
    """Steps 4(post)-8 of the per-frame pipeline.  Returns a dict."""
    disp_pp = post_processing(disp_pair).astype(np.float32)
    disparity = disp_pp * np.float32(disp_mult)          # f32 array x python scalar stays f32 in the reference
    pts = reproject(disparity, make_Q(cx, cy, f, b))
    rgb = bgr_to_rgb(frame_bgr)
    road3d, road_rgb = mask_gather(pts, rgb, road)
    fence3d, fence_rgb = mask_gather(pts, rgb, fence)
    return dict(disp_pp=disp_pp, disparity=disparity, points3d=pts, road3d=road3d, road_rgb=road_rgb,
                fence3d=fence3d, fence_rgb=fence_rgb)
