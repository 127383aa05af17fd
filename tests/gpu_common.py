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
    """max-normalised error (the north-star figure, = relerr) plus per-element statistics: a max-normalised bound alone is lenient for
    tensors with a wide range (logits).  strict_p99 / strict_max: |delta| / (|ref| + 1e-2 max|ref|), the strict figure the parity tests
    ASSERT beside the max-normalised one (STRICT_P99 / STRICT_MAX below); p99_elem_rel / max_elem_rel: the same with 1e-3 max|ref| in the
    denominator (round 2's figure: dominated by the elements nearest zero, reported only)."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    d = np.abs(a - b)
    scale = float(np.abs(b).max()) + 1e-30
    per = d / (np.abs(b) + 1e-3 * scale)
    strict = d / (np.abs(b) + 1e-2 * scale)
    return {"max_rel": float(d.max() / scale), "strict_p99": float(np.quantile(strict, 0.99)), "strict_max": float(strict.max()),
            "p99_elem_rel": float(np.quantile(per, 0.99)), "max_elem_rel": float(per.max()),
            "rms_rel": float(np.sqrt((d * d).mean()) / scale)}


# FROZEN at the round-3 values: these are not re-fitted when a kernel changes (round 4's folded upconvs / fused decoder tail had to pass them
# as they stood) -- a change that needs them loosened is a numerical regression to be explained, not absorbed.
# Bounds on the strict per-element figure |delta| / (|ref| + 1e-2 max|ref|), by engine and tensor kind, = 2 x the worst value measured at
# 512 x 1024 (profiles/r03_strict_error_*.txt: the plan and bf16x2 over 10 / 4 weight + frame seeds against the f32 engine; the f32 engine
# against the CPU oracle in the bench run and in test_gpu_pipeline): an element of magnitude >= 1 % of the tensor's maximum is then within
# STRICT_MAX relative error of the oracle's, 99 % of the elements within STRICT_P99.  The logits' figures are an order of magnitude above the
# disparities': the three logits planes cross zero everywhere, the disparities are 0.3 * sigmoid.  Two f32 implementations with
# different summation orders (the f32 engine vs the torch-CPU oracle) already differ by 3e-5 / 1.7e-4 on the logits.
STRICT_P99 = {("f32", "logits"): 7e-5, ("bf16x3", "logits"): 7e-5, ("bf16x2", "logits"): 6e-4, ("mixed", "logits"): 6e-4, ("plan", "logits"): 1.4e-2,
              ("f32", "disp"): 4e-6, ("bf16x3", "disp"): 4e-6, ("bf16x2", "disp"): 1.3e-5, ("mixed", "disp"): 9e-4, ("plan", "disp"): 9e-4}
STRICT_MAX = {("f32", "logits"): 4e-4, ("bf16x3", "logits"): 4e-4, ("bf16x2", "logits"): 4e-3, ("mixed", "logits"): 4e-3, ("plan", "logits"): 9e-2,
              ("f32", "disp"): 1e-5, ("bf16x3", "disp"): 1e-5, ("bf16x2", "disp"): 3e-5, ("mixed", "disp"): 2e-3, ("plan", "disp"): 2e-3}


# "f16x2" (round 5: fp32-grade on three fp16 MFMA products, split_fmt.hpp "HS") is held to the FROZEN bounds of the exact-f32 engine: it claims the
# reference's own precision, so it gets no bound of its own
for _k in ("logits", "disp"):
    STRICT_P99[("f16x2", _k)] = STRICT_P99[("f32", _k)]
    STRICT_MAX[("f16x2", _k)] = STRICT_MAX[("f32", _k)]


def assert_close(got, ref, precision, tol=1e-3, what="", kind="logits"):
    """the parity bar: max-normalised error within north_star's 1e-3 AND the strict per-element figure within the engine's bound
    (``kind``: 'logits' or 'disp')"""
    rep = err_report(got, ref)
    assert rep["max_rel"] < tol, (what, precision, rep)
    assert rep["strict_p99"] < STRICT_P99[(precision, kind)] and rep["strict_max"] < STRICT_MAX[(precision, kind)], (what, precision, rep)
    return rep
