"""Shared fixtures/helpers for the -m gpu tests (parity tests proper; they call through the C ABI)."""
import numpy as np
import torch

from semantic_depth_amd import _lib as L
from semantic_depth_amd import weights as W
from semantic_depth_amd.engine import Camera, Engine, RoadWidthParams  # noqa: F401

_cache = {}


def engine(H, W_, max_batch=1, encoder="resnet50", fcn_kw=None, mono_kw=None, load=("fcn", "mono"), precision="f32"):
    """engines are cached per configuration; weights are seeded (seed 1 FCN, seed 2 monodepth)."""
    key = (H, W_, max_batch, encoder, tuple(sorted((fcn_kw or {}).items())), tuple(sorted((mono_kw or {}).items())), load, precision)
    if key in _cache:
        return _cache[key]
    eng = Engine(H, W_, max_batch, encoder, precision=precision)
    wf = wm = None
    if "fcn" in load:
        wf = W.make_fcn8s_weights(1, **(fcn_kw or {}))
        eng.load_weights(L.SD_NET_FCN8S, wf)
    if "mono" in load:
        wm = W.make_monodepth_weights(encoder, 2, **(mono_kw or {}))
        eng.load_weights(L.SD_NET_MONODEPTH, wm)
    _cache[key] = (eng, wf, wm)
    return _cache[key]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def err_report(a, b):
    """max-normalised error (the north-star figure, = relerr) plus per-element statistics, |delta| / (|ref| + 1e-3 max|ref|):
    a max-normalised bound alone is lenient for tensors with a wide range (logits)."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    d = np.abs(a - b)
    scale = float(np.abs(b).max()) + 1e-30
    per = d / (np.abs(b) + 1e-3 * scale)
    return {"max_rel": float(d.max() / scale), "p99_elem_rel": float(np.quantile(per, 0.99)), "max_elem_rel": float(per.max()),
            "rms_rel": float(np.sqrt((d * d).mean()) / scale)}
