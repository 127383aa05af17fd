"""CPU restatement of ``cv2.resize(img, (W, H), interpolation=cv2.INTER_CUBIC)`` for uint8 images -- the reference's
input stage (semantic_depth.py:111, semantic_depth_cityscapes_sequence.py:128).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: OpenCV (opencv-contrib-python 4.0.0.21 in the reference's requirements.txt) is not installable here and
the reference holds no vectors for it.  This follows the published scalar algorithm of imgproc/resize.cpp:
  * source coordinate of destination pixel d: f = (d + 0.5) * scale - 0.5, s = floor(f), frac = f - s (float32)
  * bicubic weights with A = -0.75 (interpolateCubic), converted to fixed point with cvRound(w * 2048) (int16)
  * taps s-1 .. s+2, indices replicated at the borders
  * horizontal pass in int32, vertical pass in int32, result = saturate_u8((sum + 2^21) >> 22)
OpenCV's SIMD vertical pass (VResizeCubicVec_32s8u) evaluates the same sum in float32 and may differ by one level on a small
fraction of pixels; the scalar path is the definition taken here.
"""
import numpy as np


def _coeffs(dst: int, src: int):
    inv = np.float64(dst) / np.float64(src)
    scale = np.float64(1.0) / inv
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    x = (f - s.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    xm = one - x
    c2 = ((A + np.float32(2)) * xm - (A + np.float32(3))) * xm * xm + one
    c3 = one - c0 - c1 - c2
    c = np.stack([c0, c1, c2, c3], 1).astype(np.float32)
    ic = np.rint(c * np.float32(2048.0)).astype(np.int64)           # cvRound: round half to even
    ic = np.clip(ic, -32768, 32767).astype(np.int32)
    idx = np.clip(s[:, None] - 1 + np.arange(4)[None, :], 0, src - 1)
    return idx, ic


def resize_cubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img u8 [H, W, C] -> u8 [out_h, out_w, C]"""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    if (H, W) == (out_h, out_w):
        return img.copy()                                            # cv2.resize returns a copy for equal sizes
    xi, xa = _coeffs(out_w, W)
    yi, ya = _coeffs(out_h, H)
    s = img.astype(np.int64)
    hor = np.zeros((H, out_w, img.shape[2]), np.int64)
    for k in range(4):
        hor += s[:, xi[:, k], :] * xa[None, :, k, None]
    ver = np.zeros((out_h, out_w, img.shape[2]), np.int64)
    for k in range(4):
        ver += hor[yi[:, k], :, :] * ya[:, k, None, None]
    out = (ver + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)
