"""GPU: other frame geometries through the whole path -- the reference's own 256 x 512 input (semantic_depth.py:105-112), a
KITTI-like 384 x 1280, a full-resolution Cityscapes frame 1024 x 2048 and an odd batch: the split engines against the exact-f32
engine (the layer routing -- direct / LDS-DMA / register-staged kernels, tile shapes, the ':x' closure -- depends on the geometry),
process_batch records included."""
import numpy as np
import pytest
import torch

from semantic_depth_amd import _lib as L
from semantic_depth_amd import weights as Wt
from semantic_depth_amd.engine import Camera, Engine, RoadWidthParams

pytestmark = pytest.mark.gpu
TOL = 1e-3          # north_star: "within 1e-3 relative fp32 tolerance"


def _frames(B, H, W, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (B, H // 8, W // 8, 3), dtype=np.uint8)
    fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    return torch.from_numpy((fr.astype(np.int16) + rng.integers(-16, 17, fr.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)).cuda()


@pytest.mark.parametrize("H,W,B", [(256, 512, 5), (384, 1280, 2), (1024, 2048, 1)])
def test_other_geometries_track_the_f32_engine(H, W, B):
    wf = Wt.make_fcn8s_weights(1, decoder_std=0.05)
    wm = Wt.make_monodepth_weights("resnet50", 2)
    fr = _frames(B, H, W, seed=H + W)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    got = {}
    for prec in ("f32", "plan", "bf16x2", "f16x2"):
        e = Engine(H, W, B, "resnet50", precision=prec)
        e.load_weights(L.SD_NET_FCN8S, wf)
        e.load_weights(L.SD_NET_MONODEPTH, wm)
        lg = e.fcn8s_forward(fr, want_logits=True)["logits"].clone()
        pp = e.monodepth_forward(fr).clone()
        cam = Camera(W / 2 - 0.5, H / 2 - 0.5, 10.0 * float(pp.median()) * W, 1.0, float(W))
        rec = Engine.records(e.process_batch(fr, [cam] * B, RoadWidthParams())["records"])
        got[prec] = (lg, pp, rec["n_road"].astype(np.int64), rec["found"].copy(), rec["width"].copy())
        e.close()
    assert got["f32"][2].min() > 1000                      # the masks are not empty
    for prec, tol in (("plan", TOL), ("bf16x2", 1e-4), ("f16x2", 1e-5)):       # (f16x2: fp32-grade, an order of magnitude inside bf16x2's budget)
        lg, pp, n, found, width = got[prec]
        print(H, W, B, prec, rel(lg, got["f32"][0]), rel(pp, got["f32"][1]))
        assert rel(lg, got["f32"][0]) < tol and rel(pp, got["f32"][1]) < tol
        assert np.abs(n - got["f32"][2]).max() <= max(4, int(2e-3 * got["f32"][2].max()))
        assert np.array_equal(found, got["f32"][3])
        both = found.astype(bool)
        assert np.abs(width[both] - got["f32"][4][both]).max() < 0.05 if both.any() else True
