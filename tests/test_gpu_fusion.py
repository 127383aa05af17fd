"""GPU parity: post-processing, back-projection and ordered mask gather (csrc/fuse.hip) vs the oracle — bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fusion, pipeline
from helpers import checksum
from gpu_common import Camera, dev, engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_small():
    return engine(128, 256, 4, "resnet50", load=())[0]


@pytest.fixture(scope="module")
def eng_full():
    return engine(512, 1024, 2, "resnet50", load=())[0]


def test_post_processing_bit_exact(eng_small):
    rng = np.random.default_rng(5)
    raw = (0.3 * rng.random((3, 2, 128, 256))).astype(np.float32)
    got = eng_small.post_process(dev(raw)).cpu().numpy()
    for b in range(3):
        ref = fusion.post_processing(raw[b]).astype(np.float32)
        assert np.array_equal(got[b], ref)
    # closed form: identical left / re-flipped right -> output equals the input
    same = np.stack([raw[0, 0], raw[0, 0][:, ::-1]])[None]
    got = eng_small.post_process(dev(same)).cpu().numpy()[0]
    assert np.abs(got - raw[0, 0]).max() <= 2e-8 * 0.3 * 4


def _fuse_ref(dp, road, fence, frame, cam):
    return fusion.fuse(dp, road, fence, frame, **cam)


@pytest.mark.parametrize("one_pixel_per_thread", [False, True])
def test_fuse_matches_oracle_small(eng_small, one_pixel_per_thread, monkeypatch):
    """both forms of the gather: four pixels per thread (widths that are multiples of 4) and the generic one behind it"""
    e = eng_small
    if one_pixel_per_thread:
        monkeypatch.setenv("SEMDEPTH_DISABLE", "fuse4")            # switches are latched when the handle is created
        from semantic_depth_amd.engine import Engine
        e = Engine(128, 256, 4, "resnet50")
    scenes = [pipeline.synthetic_scene(128, 256, seed=s, f=250.0, fences=True) for s in (3, 4, 5)]
    cams = [Camera(**s[4]) for s in scenes]
    cams[1] = Camera(cams[1].cx + 2.25, cams[1].cy - 1.5, 300.0, 0.6, 3800.0)      # per-frame cameras differ
    pp = e.post_process(dev(np.stack([s[0] for s in scenes])))
    out = e.fuse_backproject(pp, dev(np.stack([s[1] for s in scenes]).astype(np.uint8)),
                             dev(np.stack([s[2] for s in scenes]).astype(np.uint8)), dev(np.stack([s[3] for s in scenes])),
                             cams, dense=True)
    for b, s in enumerate(scenes):
        cam = dict(cx=cams[b].cx, cy=cams[b].cy, f=cams[b].f, b=cams[b].b, disp_mult=cams[b].disp_mult)
        ref = _fuse_ref(s[0], s[1], s[2], s[3], cam)
        assert np.array_equal(pp[b].cpu().numpy(), ref["disp_pp"])
        assert np.array_equal(out["dense"][b].cpu().numpy(), ref["points3d"])
        nr, nf = int(out["n_road"][b]), int(out["n_fence"][b])
        assert nr == len(ref["road3d"]) and nf == len(ref["fence3d"]) and nf > 0
        assert np.array_equal(out["road_xyz"][b, :nr].cpu().numpy(), ref["road3d"])
        assert np.array_equal(out["road_rgb"][b, :nr].cpu().numpy(), ref["road_rgb"])
        assert np.array_equal(out["fence_xyz"][b, :nf].cpu().numpy(), ref["fence3d"])
        assert np.array_equal(out["fence_rgb"][b, :nf].cpu().numpy(), ref["fence_rgb"])


def test_fuse_edge_cases(eng_small):
    e = eng_small
    dp, road, fence, frame, cam = pipeline.synthetic_scene(128, 256, seed=9, f=250.0)
    pp = e.post_process(dev(dp[None]))
    # empty masks, full masks
    zeros = np.zeros((1, 128, 256), np.uint8)
    ones = np.ones((1, 128, 256), np.uint8)
    out = e.fuse_backproject(pp, dev(zeros), dev(ones), dev(frame[None]), [Camera(**cam)])
    assert int(out["n_road"][0]) == 0 and int(out["n_fence"][0]) == 128 * 256
    ref = fusion.fuse(dp, zeros[0].astype(bool), ones[0].astype(bool), frame, **cam)
    assert np.array_equal(out["fence_xyz"][0].cpu().numpy(), ref["fence3d"])
    # zero disparity -> +-inf / nan exactly like IEEE division in the oracle
    pp0 = pp.clone()
    pp0[0, 5, :] = 0.0
    out = e.fuse_backproject(pp0, dev(ones), dev(zeros), dev(frame[None]), [Camera(**cam)], dense=True)
    d0 = pp0[0].cpu().numpy() * np.float32(cam["disp_mult"])
    with np.errstate(all="ignore"):
        ref = fusion.reproject(d0, fusion.make_Q(cam["cx"], cam["cy"], cam["f"], cam["b"]))
    assert np.array_equal(out["dense"][0].cpu().numpy(), ref, equal_nan=True)
    assert not np.isfinite(ref[5]).all()
    # capacity smaller than the cloud: count is the true count, rows beyond cap are dropped, rows below intact
    out2 = e.fuse_backproject(pp, dev(ones), dev(zeros), dev(frame[None]), [Camera(**cam)], cap=1000)
    assert int(out2["n_road"][0]) == 128 * 256
    full = e.fuse_backproject(pp, dev(ones), dev(zeros), dev(frame[None]), [Camera(**cam)])
    assert torch.equal(out2["road_xyz"][0], full["road_xyz"][0, :1000])


def test_full_size_golden_digest(eng_full, golden_dir):
    """512x1024 Appendix-F scene: clouds must reproduce the digests captured with the reference's pcl inputs."""
    g = json.load(open(os.path.join(golden_dir, "pcl_full.json")))
    sc = g["scene"]
    dp, road, fence, frame, cam = pipeline.synthetic_scene(sc["h"], sc["w"], seed=sc["seed"], f=sc["f"])
    e = eng_full
    pp = e.post_process(dev(dp[None]))
    out = e.fuse_backproject(pp, dev(road[None].astype(np.uint8)), dev(fence[None].astype(np.uint8)), dev(frame[None]), [Camera(**cam)])
    n = int(out["n_road"][0])
    assert n == g["n_road"] == int(road.sum())
    assert checksum(out["road_xyz"][0, :n].cpu().numpy()) == g["road3d_checksum"]
    assert checksum(out["road_rgb"][0, :n].cpu().numpy()) == g["road_rgb_checksum"]


def test_full_size_properties(eng_full):
    """size-independent properties at 512x1024, B=2: counts == mask sums, order == row-major, batch independence."""
    rng = np.random.default_rng(11)
    H, W = 512, 1024
    raw = (0.05 + 0.25 * rng.random((2, 2, H, W))).astype(np.float32)
    road = (rng.random((2, H, W)) < 0.3).astype(np.uint8)
    fence = (rng.random((2, H, W)) < 0.01).astype(np.uint8)
    frames = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    cams = [Camera(W / 2, H / 2, 1000.0, 1.0, float(W)), Camera(1048.64 / 2, 519.277 / 2, 1000.0, 1.0, 3800.0)]
    e = eng_full
    pp = e.post_process(dev(raw))
    out = e.fuse_backproject(pp, dev(road), dev(fence), dev(frames), cams, dense=True)
    for b in range(2):
        nr, nf = int(out["n_road"][b]), int(out["n_fence"][b])
        assert nr == int(road[b].sum()) and nf == int(fence[b].sum())
        dense = out["dense"][b].cpu().numpy()
        assert np.array_equal(out["road_xyz"][b, :nr].cpu().numpy(), dense[road[b].astype(bool)])
        assert np.array_equal(out["fence_xyz"][b, :nf].cpu().numpy(), dense[fence[b].astype(bool)])
        assert np.array_equal(out["road_rgb"][b, :nr].cpu().numpy(), frames[b][..., ::-1][road[b].astype(bool)])
        # same frame alone gives the same bits
        solo = e.fuse_backproject(pp[b:b + 1], dev(road[b:b + 1]), dev(fence[b:b + 1]), dev(frames[b:b + 1]), [cams[b]])
        assert torch.equal(solo["road_xyz"][0, :nr], out["road_xyz"][b, :nr])


# ------------------------------------------------------------------------------------------------ the one-pass form
def _scene_batch(h, w, seeds, f):
    scenes = [pipeline.synthetic_scene(h, w, seed=s, f=f, fences=True) for s in seeds]
    cams = [Camera(**s[4]) for s in scenes]
    return scenes, cams, dev(np.stack([s[0] for s in scenes])), dev(np.stack([s[1] for s in scenes]).astype(np.uint8)), \
        dev(np.stack([s[2] for s in scenes]).astype(np.uint8)), dev(np.stack([s[3] for s in scenes]))


def _assert_fuse_equals_oracle(out, scenes, cams, pp=None):
    for b, s in enumerate(scenes):
        cam = dict(cx=cams[b].cx, cy=cams[b].cy, f=cams[b].f, b=cams[b].b, disp_mult=cams[b].disp_mult)
        ref = _fuse_ref(s[0], s[1], s[2], s[3], cam)
        if pp is not None:
            assert np.array_equal(pp[b].cpu().numpy(), ref["disp_pp"])
        nr, nf = int(out["n_road"][b]), int(out["n_fence"][b])
        assert nr == len(ref["road3d"]) and nf == len(ref["fence3d"])
        assert np.array_equal(out["road_xyz"][b, :nr].cpu().numpy(), ref["road3d"])
        assert np.array_equal(out["road_rgb"][b, :nr].cpu().numpy(), ref["road_rgb"])
        assert np.array_equal(out["fence_xyz"][b, :nf].cpu().numpy(), ref["fence3d"])
        assert np.array_equal(out["fence_rgb"][b, :nf].cpu().numpy(), ref["fence_rgb"])


@pytest.mark.parametrize("switch", [None, "fuse1", "fuse1,fuse4"])
def test_one_pass_and_three_launch_forms_agree_with_the_oracle(switch, monkeypatch):
    """sd_fuse_backproject (post-processed map in) and sd_postprocess_fuse_backproject (raw pair in, post-processing folded into
    the same pass) in the one-pass look-back form (default) and the three-launch forms behind it: all bit-exact vs the oracle"""
    from semantic_depth_amd.engine import Engine
    if switch:
        monkeypatch.setenv("SEMDEPTH_DISABLE", switch)
    e = Engine(128, 256, 4, "resnet50")
    scenes, cams, raw, road, fence, frames = _scene_batch(128, 256, (3, 4, 5), 250.0)
    cams[1] = Camera(cams[1].cx + 2.25, cams[1].cy - 1.5, 300.0, 0.6, 3800.0)
    pp = e.post_process(raw)
    _assert_fuse_equals_oracle(e.fuse_backproject(pp, road, fence, frames, cams), scenes, cams, pp)
    out = e.fuse_from_raw(road, fence, frames, cams, disp_raw=raw)
    _assert_fuse_equals_oracle(out, scenes, cams, out["disp_pp"])
    # one class only, no colours, capacity clamp
    solo = e.fuse_from_raw(road, None, None, cams, disp_raw=raw, cap=777)
    assert solo["fence_xyz"] is None and solo["road_rgb"] is None
    assert torch.equal(solo["n_road"], out["n_road"]) and torch.equal(solo["road_xyz"][:, :777], out["road_xyz"][:, :777])


def test_one_pass_full_size_many_launches_and_changing_batch():
    """512 x 1024 (512 look-back blocks per frame), random masks: order == row-major, counts == mask sums; then 1100 launches on
    a small handle so that the epoch of the look-back words wraps (the scratch is re-zeroed) with the batch size changing"""
    from semantic_depth_amd.engine import Engine
    H, W = 512, 1024
    e = engine(H, W, 2, "resnet50", load=())[0]
    rng = np.random.default_rng(21)
    raw = (0.05 + 0.25 * rng.random((2, 2, H, W))).astype(np.float32)
    road = (rng.random((2, H, W)) < 0.4).astype(np.uint8)
    fence = ((rng.random((2, H, W)) < 0.3) & (road == 0)).astype(np.uint8)
    road[1, :100] = 0                                   # long runs of empty and of full blocks
    fence[1, 300:] = 1
    frames = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
    cams = [Camera(W / 2, H / 2, 1000.0, 1.0, float(W)), Camera(1048.64 / 2, 519.277 / 2, 1000.0, 1.0, 3800.0)]
    pp = e.post_process(dev(raw))
    dense = e.fuse_backproject(pp, None, None, None, cams, dense=True, want_fence=False)["dense"].cpu().numpy()
    for _ in range(3):
        out = e.fuse_from_raw(dev(road), dev(fence), dev(frames), cams, disp_raw=dev(raw))
        assert torch.equal(out["disp_pp"], pp)
        for b in range(2):
            nr, nf = int(out["n_road"][b]), int(out["n_fence"][b])
            assert nr == int(road[b].sum()) and nf == int(fence[b].sum())
            assert np.array_equal(out["road_xyz"][b, :nr].cpu().numpy(), dense[b][road[b].astype(bool)])
            assert np.array_equal(out["fence_xyz"][b, :nf].cpu().numpy(), dense[b][fence[b].astype(bool)])
            assert np.array_equal(out["fence_rgb"][b, :nf].cpu().numpy(), frames[b][..., ::-1][fence[b].astype(bool)])
    s = Engine(64, 128, 3, "resnet50")
    scenes, cams3, raw3, road3, fence3, frames3 = _scene_batch(64, 128, (1, 2, 3), 125.0)
    want = s.fuse_from_raw(road3, fence3, frames3, cams3, disp_raw=raw3)
    for it in range(1100):
        nb = 1 + it % 3
        got = s.fuse_from_raw(road3[:nb], fence3[:nb], frames3[:nb], cams3[:nb], disp_raw=raw3[:nb])
        if it % 97 == 0 or it > 1015:
            assert torch.equal(got["n_road"], want["n_road"][:nb]) and torch.equal(got["n_fence"], want["n_fence"][:nb])
            for b in range(nb):
                n = int(got["n_road"][b])
                assert torch.equal(got["road_xyz"][b, :n], want["road_xyz"][b, :n]) and torch.equal(got["road_rgb"][b, :n], want["road_rgb"][b, :n])
