"""GPU parity at BASELINE.json's frame size (512 x 1024) and through the reference-shaped boundary:

  * FCN-8s logits + taps and monodepth-resnet50 disparities of ONE full-size frame vs the CPU oracle, every precision
    (configs[1], configs[2]: the 512-channel direct-conv passes, 256x256 LDS-DMA blocks, fc6 at K = 25088 and the
    source-resolution upconv tiles only exist at this size);
  * Engine.process_batch at B = 8 and B = 32 (configs[3]): the records must equal oracle.pipeline.frame_tail fed the GPU's
    OWN raw disparities and masks, bit for bit ("same disparity + mask -> identical selected pixels");
  * approach='both': fence chain + fence-to-fence vs oracle.pipeline.fence_tail;
  * records of the split engine vs the exact-f32 engine with an explicit tolerance;
  * api.SegmentFrame / DepthFrame / FrameProcessor built the way the reference's main() builds them.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fusion, nets, pipeline
from semantic_depth_amd import _lib as L
from semantic_depth_amd import weights as Wt
from semantic_depth_amd.engine import Camera, Engine, FenceParams, RoadWidthParams
from gpu_common import assert_close, dev, engine, err_report, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-3          # north_star: "within 1e-3 relative fp32 tolerance"
H, W = 512, 1024


def _smooth_frames(B, h=H, w=W, seed=0):
    """the bench's frames: low-pass of uniform noise + a little noise, so that the random-weight masks form regions"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (B, h // 8, w // 8, 3), dtype=np.uint8)
    fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    return (fr.astype(np.int16) + rng.integers(-16, 17, fr.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def keep_taps():
    os.environ["SEMDEPTH_KEEP_ACTIVATIONS"] = "1"
    yield
    os.environ.pop("SEMDEPTH_KEEP_ACTIVATIONS", None)


@pytest.fixture(scope="module")
def oracle_full():
    """one full-size frame through the CPU oracle (~6 s on the GPU box), shared by the precision cases"""
    wf = Wt.make_fcn8s_weights(1, decoder_std=0.05, bias_std=0.1)
    wm = Wt.make_monodepth_weights("resnet50", 2, bias_std=0.05)
    fr = _smooth_frames(1, seed=41)
    logits, taps = nets.fcn8s_forward(fr, wf, return_taps=True)
    f = fr[0].astype(np.float32) / 255
    scales = nets.monodepth_forward(np.stack((f, np.fliplr(f)), 0), wm, "resnet50", all_scales=True)
    return dict(wf=wf, wm=wm, frames=fr, logits=logits, taps=taps, scales=scales)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2", "bf16x2", "plan"])
def test_fcn8s_full_size_matches_oracle(precision, oracle_full, keep_taps):
    o = oracle_full
    eng = Engine(H, W, 1, "resnet50", precision=precision)
    eng.load_weights(L.SD_NET_FCN8S, o["wf"])
    eng.load_weights(L.SD_NET_MONODEPTH, o["wm"])
    out = eng.fcn8s_forward(dev(o["frames"]), want_logits=True)
    for name, key in (("layer3_out", "layer3"), ("layer4_out", "layer4"), ("layer7_out", "layer7"), ("first_skip", "first_skip"),
                      ("second_skip", "second_skip")):
        got = eng.net_tensor(L.SD_NET_FCN8S, name).cpu().numpy()
        assert got.shape == o["taps"][key].shape
        assert relerr(got, o["taps"][key]) < TOL, (name, err_report(got, o["taps"][key]))
    lg = out["logits"].cpu().numpy()
    rep = assert_close(lg, o["logits"], precision, TOL, "logits")       # max-normalised < 1e-3 AND the strict per-element bound
    print("fcn8s 512x1024 logits", precision, rep)
    _, road_r, fence_r, am_r = nets.softmax_masks(o["logits"])
    assert float((out["road"].cpu().numpy().astype(bool) != road_r).mean()) < 2e-3
    assert float((out["argmax"].cpu().numpy() != am_r).mean()) < 2e-3
    assert 0.02 < road_r.mean() < 0.98


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2", "bf16x2", "mixed", "plan"])
def test_monodepth_full_size_matches_oracle(precision, oracle_full, keep_taps):
    o = oracle_full
    eng = Engine(H, W, 1, "resnet50", precision=precision)
    eng.load_weights(L.SD_NET_FCN8S, o["wf"])
    eng.load_weights(L.SD_NET_MONODEPTH, o["wm"])
    pp, raw = eng.monodepth_forward(dev(o["frames"]), want_raw=True)
    raw = raw.cpu().numpy()[0]
    ref_raw = o["scales"][1][..., 0]
    rep = assert_close(raw, ref_raw, precision, TOL, "raw disparity pair", kind="disp")
    print("monodepth-resnet50 512x1024 disparity", precision, rep)
    for lvl in (4, 3, 2):
        got = eng.net_tensor(L.SD_NET_MONODEPTH, f"dec/disp{lvl}").cpu().numpy()
        assert relerr(got[:2], o["scales"][lvl]) < TOL, lvl
    ref_pp = fusion.post_processing(ref_raw.astype(np.float32)).astype(np.float32)
    assert relerr(pp.cpu().numpy()[0], ref_pp) < TOL
    assert np.array_equal(pp.cpu().numpy()[0], fusion.post_processing(raw).astype(np.float32))


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2", "bf16x2", "plan"])
def test_monodepth_vgg_full_size_matches_oracle(precision):
    """the reference's DEFAULT monodepth encoder (--monodepth_encoder vgg, semantic_depth.py:721-722) at BASELINE's frame size against
    the CPU oracle: raw disparity pair and the post-processed map (this encoder had only a 128 x 256 oracle test).
    Under SD_PREC_PLAN the vgg encoder keeps three products (the built-in monodepth plan names ResNet-50 layers)."""
    wm = Wt.make_monodepth_weights("vgg", 2, gain=1.0, bias_std=0.05)
    fr = _smooth_frames(1, seed=45)
    f = fr[0].astype(np.float32) / 255
    ref_raw = nets.monodepth_forward(np.stack((f, np.fliplr(f)), 0), wm, "vgg")[..., 0]
    eng = Engine(H, W, 1, "vgg", precision=precision)
    eng.load_weights(L.SD_NET_MONODEPTH, wm)
    eng.load_weights(L.SD_NET_FCN8S, Wt.make_fcn8s_weights(1, decoder_std=0.05))
    pp, raw = eng.monodepth_forward(dev(fr), want_raw=True)
    raw = raw.cpu().numpy()[0]
    rep = assert_close(raw, ref_raw, "bf16x2" if precision == "plan" else precision, TOL, "raw disparity pair (vgg encoder)", kind="disp")
    print("monodepth-vgg 512x1024 disparity", precision, rep)
    assert np.array_equal(pp.cpu().numpy()[0], fusion.post_processing(raw).astype(np.float32))
    assert relerr(pp.cpu().numpy()[0], fusion.post_processing(ref_raw.astype(np.float32)).astype(np.float32)) < TOL


@pytest.fixture(scope="module")
def oracle_b8():
    """BASELINE.json configs[1] / configs[2] as written: B = 8 frames of 512 x 1024 through the CPU oracle (once per module)"""
    wf = Wt.make_fcn8s_weights(3, decoder_std=0.05, bias_std=0.1)
    wm = Wt.make_monodepth_weights("resnet50", 4, bias_std=0.05)
    fr = _smooth_frames(8, seed=43)
    logits = np.concatenate([nets.fcn8s_forward(fr[b:b + 1], wf) for b in range(8)], 0)
    disp = []
    for b in range(8):
        f = fr[b].astype(np.float32) / 255
        disp.append(nets.monodepth_forward(np.stack((f, np.fliplr(f)), 0), wm, "resnet50", all_scales=True)[1][..., 0])
    return dict(wf=wf, wm=wm, frames=fr, logits=logits, disp=np.stack(disp))


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2", "plan"])
def test_nets_b8_full_size_match_oracle(precision, oracle_b8):
    """configs[1] (FCN-8s forward, B = 8) and configs[2] (monodepth-resnet50 forward on frame + flip, B = 8) in one network pass
    each, every frame against the oracle: logits, masks, raw disparity pair, post-processed disparity"""
    o = oracle_b8
    eng = Engine(H, W, 8, "resnet50", precision=precision)
    eng.load_weights(L.SD_NET_FCN8S, o["wf"])
    eng.load_weights(L.SD_NET_MONODEPTH, o["wm"])
    fr = dev(o["frames"])
    out = eng.fcn8s_forward(fr, want_logits=True)
    pp, raw = eng.monodepth_forward(fr, want_raw=True)
    lg, raw, pp = out["logits"].cpu().numpy(), raw.cpu().numpy(), pp.cpu().numpy()
    worst_l = worst_d = 0.0
    for b in range(8):
        worst_l = max(worst_l, relerr(lg[b], o["logits"][b]))
        worst_d = max(worst_d, relerr(raw[b], o["disp"][b]))
        _, road_r, _, am_r = nets.softmax_masks(o["logits"][b:b + 1])
        assert float((out["road"][b].cpu().numpy().astype(bool) != road_r[0]).mean()) < 2e-3, b
        assert float((out["argmax"][b].cpu().numpy() != am_r[0]).mean()) < 2e-3, b
        assert np.array_equal(pp[b], fusion.post_processing(raw[b]).astype(np.float32)), b
        assert relerr(pp[b], fusion.post_processing(o["disp"][b].astype(np.float32)).astype(np.float32)) < TOL, b
        assert_close(lg[b], o["logits"][b], precision, TOL, f"logits of frame {b}")
        assert_close(raw[b], o["disp"][b], precision, TOL, f"raw disparity of frame {b}", kind="disp")
    print("B=8 512x1024", precision, "worst logits", worst_l, "worst disparity", worst_d)
    assert worst_l < TOL and worst_d < TOL


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2", "plan"])
def test_nets_b32_as_benchmarked_match_oracle(precision, oracle_b8):
    """configs[3] as bench.py runs it: ONE pass of B = 32 frames per network on the benchmarked engines; frames 0, 9, 22 and 31 of the
    batch (= frames 0..3 of the oracle batch, placed there) against the CPU oracle: logits, masks, raw and post-processed disparity.
    The other 28 frames are different images, so every checked frame sits between foreign neighbours in the 32-frame tensors."""
    o = oracle_b8
    eng, _, _ = engine(H, W, 32, "resnet50", fcn_kw=dict(decoder_std=0.05), precision=precision)
    eng.load_weights(L.SD_NET_FCN8S, o["wf"])
    eng.load_weights(L.SD_NET_MONODEPTH, o["wm"])
    try:
        frames_np = _smooth_frames(32, seed=77)
        slots = (0, 9, 22, 31)
        for i, b in enumerate(slots):
            frames_np[b] = o["frames"][i]
        fr = dev(frames_np)
        out = eng.fcn8s_forward(fr, want_logits=True)
        pp, raw = eng.monodepth_forward(fr, want_raw=True)
        for i, b in enumerate(slots):
            rl = assert_close(out["logits"][b].cpu().numpy(), o["logits"][i], precision, TOL, f"logits of frame {b}")
            rd = assert_close(raw[b].cpu().numpy(), o["disp"][i], precision, TOL, f"raw disparity of frame {b}", kind="disp")
            _, road_r, _, am_r = nets.softmax_masks(o["logits"][i:i + 1])
            assert float((out["road"][b].cpu().numpy().astype(bool) != road_r[0]).mean()) < 2e-3, b
            assert float((out["argmax"][b].cpu().numpy() != am_r[0]).mean()) < 2e-3, b
            assert np.array_equal(pp[b].cpu().numpy(), fusion.post_processing(raw[b].cpu().numpy()).astype(np.float32)), b
            print(f"B=32 {precision} frame {b}: logits {rl['max_rel']:.2e} (strict p99 {rl['strict_p99']:.2e}, max {rl['strict_max']:.2e}); "
                  f"disparity {rd['max_rel']:.2e} (strict p99 {rd['strict_p99']:.2e}, max {rd['strict_max']:.2e})")
    finally:      # the cached engine goes back to the weights the other tests expect
        eng.load_weights(L.SD_NET_FCN8S, Wt.make_fcn8s_weights(1, decoder_std=0.05))
        eng.load_weights(L.SD_NET_MONODEPTH, Wt.make_monodepth_weights("resnet50", 2))


def _camera_at_10m(disp_pp, mult=float(W)):
    """a camera that puts the median disparity at 10 m, so that the z-cut / depth window / Open3D filters all see points"""
    d0 = float(disp_pp.median().item()) * mult
    return Camera(W / 2 - 0.5, H / 2 - 0.5, 10.0 * d0, 1.0, mult)


def _check_records_against_oracle(eng, frames_np, out, raw, cam, prm, colours=True):
    """oracle.pipeline.frame_tail fed the GPU's own raw disparity pairs and masks must reproduce every field of the records"""
    recs = Engine.records(out["records"])
    cam_d = dict(cx=cam.cx, cy=cam.cy, f=cam.f, b=cam.b, disp_mult=cam.disp_mult)
    found = 0
    for b in range(frames_np.shape[0]):
        road = out["seg"]["road"][b].cpu().numpy().astype(bool)
        fence = out["seg"]["fence"][b].cpu().numpy().astype(bool)
        ref = pipeline.frame_tail(raw[b].cpu().numpy(), road, fence, frames_np[b], cam_d, pipeline.RoadWidthParams(**{
            k: getattr(prm, k) for k in ("depth", "z_cut", "mad_y", "mad_x", "plane_thr", "sor_k", "sor_ratio", "ror_n", "ror_r", "depth_offset", "use_o3d")}))
        assert np.array_equal(out["disp_pp"][b].cpu().numpy(), ref["disp_pp"]), b
        n = int(out["fuse"]["n_road"][b])
        assert n == ref["road3d"].shape[0], b
        assert np.array_equal(out["fuse"]["road_xyz"][b, :n].cpu().numpy(), ref["road3d"]), b
        if colours:
            assert np.array_equal(out["fuse"]["road_rgb"][b, :n].cpu().numpy(), ref["road_rgb"]), b
        nf = int(out["fuse"]["n_fence"][b])
        assert nf == ref["fence3d"].shape[0] and np.array_equal(out["fuse"]["fence_xyz"][b, :nf].cpu().numpy(), ref["fence3d"]), b
        rw, r = ref["rw"], recs[b]
        got = [int(r[k]) for k in ("n_road", "n_zcut", "n_mad_y", "n_mad_x", "n_plane", "n_sor", "n_ror")]
        want = [rw[k] for k in ("n_in", "n_zcut", "n_mad_y", "n_mad_x", "n_plane", "n_sor", "n_ror")]
        assert got == want, (b, got, want)
        assert bool(r["found"]) == rw["found"], b
        if rw["found"]:
            found += 1
            assert float(r["width"]) == rw["width"] and float(r["x_left"]) == rw["x_left"] and float(r["x_right"]) == rw["x_right"], b
        if rw["plane"] is not None:
            np.testing.assert_allclose(r["plane"], [rw["plane"][k] for k in ("Cx", "Cy", "Cz", "C")], rtol=1e-8, atol=1e-10)
    return recs, found


# (the benchmarked batch of 32 on the headline engine and on the plan leg; the other engines on 8: the CPU oracle tail costs ~1 s per frame,
#  and the tail kernels are the same whatever engine produced the masks and the disparity)
@pytest.mark.parametrize("B,precision", [(8, "bf16x2"), (8, "f32"), (32, "plan"), (8, "bf16x3"), (32, "f16x2")])
def test_process_batch_records_equal_oracle_tail(B, precision):
    """configs[3] (B = 32: the benchmarked configuration, on the engines bench.py times -- f32 headline, plan leg) and the B = 8 batch
    of configs[1]/[2] through Engine.process_batch"""
    eng, wf, wm = engine(H, W, 32, "resnet50", fcn_kw=dict(decoder_std=0.05), precision=precision)
    frames_np = _smooth_frames(B, seed=100 + B)
    fr = dev(frames_np)
    pp, raw = eng.monodepth_forward(fr, want_raw=True)
    cam = _camera_at_10m(pp)
    prm = RoadWidthParams()
    out = eng.process_batch(fr, [cam] * B, prm, want_final=True)
    assert torch.equal(out["disp_pp"], pp)
    recs, found = _check_records_against_oracle(eng, frames_np, out, raw, cam, prm)
    print(f"B={B} {precision}: n_road mean {recs['n_road'].mean():.0f}, after chain {recs['n_ror'].mean():.0f}, found {found}/{B}")
    assert recs["n_road"].mean() > 10000 and recs["n_ror"].mean() > 100 and found >= B // 2      # the chain had real work
    # the denoised cloud and its colours (carried through every filter like the reference's road_colors)
    for b in (0, B - 1):
        n = int(out["road_final"]["n"][b])
        assert n == int(recs["n_ror"][b])
        road = out["seg"]["road"][b].cpu().numpy().astype(bool)
        ref = pipeline.frame_tail(raw[b].cpu().numpy(), road, np.zeros_like(road), frames_np[b],
                                  dict(cx=cam.cx, cy=cam.cy, f=cam.f, b=cam.b, disp_mult=cam.disp_mult))["rw"]
        assert np.array_equal(out["road_final"]["xyz"][b, :n].cpu().numpy().astype(np.float64), ref["points"])
        assert np.array_equal(out["road_final"]["rgb"][b, :n].cpu().numpy(), ref["colors"])


def test_process_batch_both_matches_oracle_fence_tail():
    """approach='both' (semantic_depth.py:273-334) on the synthetic two-fence scene at 512 x 1024, B = 2: the scene's masks and
    disparities go through fuse -> road chain -> fence chain -> f2f on the GPU and through oracle.pipeline on the CPU"""
    eng, _, _ = engine(H, W, 32, "resnet50", fcn_kw=dict(decoder_std=0.05), precision="bf16x2")
    B = 2
    scenes = [pipeline.synthetic_scene(H, W, seed=77 + i, f=1000.0, fences=True) for i in range(B)]
    dp = np.stack([s[0] for s in scenes])
    pp = eng.post_process(dev(dp))
    road = dev(np.stack([s[1] for s in scenes]).astype(np.uint8))
    fence = dev(np.stack([s[2] for s in scenes]).astype(np.uint8))
    frames = dev(np.stack([s[3] for s in scenes]))
    cams = [Camera(**s[4]) for s in scenes]
    fz = eng.fuse_backproject(pp, road, fence, frames, cams)
    rw, fin, frgb, nfin = eng.road_width(fz["road_xyz"], fz["n_road"], RoadWidthParams(), want_final=True, road_rgb=fz["road_rgb"])
    f2, cl = eng.fence_to_fence(fz["fence_xyz"], fz["n_fence"], rw, FenceParams(), fence_rgb=fz["fence_rgb"], want_clouds=True)
    recs, f2r = Engine.records(rw), Engine.f2f_records(f2)
    for b in range(B):
        ref = pipeline.frame_tail(dp[b], scenes[b][1], scenes[b][2], scenes[b][3], scenes[b][4])
        ft = pipeline.fence_tail(ref["fence3d"], ref["fence_rgb"], ref["rw"]["plane"])
        assert [int(c) for c in f2r[b]["counts"]] == [ft["n_fence"], ft["n_mad_y"], ft["n_thr"], ft["n_left"], ft["n_right"],
                                                       ft["n_left_final"], ft["n_right_final"]]
        assert bool(f2r[b]["ok"]) and float(recs[b]["width"]) == ref["rw"]["width"]
        np.testing.assert_allclose(f2r[b]["dist"], ft["dist"], rtol=1e-9)
        np.testing.assert_allclose(f2r[b]["left_pt"], ft["left_pt"], rtol=1e-9, atol=1e-12)
        nl, nr = int(f2r[b]["counts"][5]), int(f2r[b]["counts"][6])
        assert np.array_equal(cl["left_xyz"][b, :nl].cpu().numpy(), ft["left"]) and np.array_equal(cl["right_xyz"][b, :nr].cpu().numpy(), ft["right"])
        # colours follow their points: every kept fence point still carries the colour of its source pixel
        src = {tuple(p): tuple(c) for p, c in zip(ref["fence3d"].tolist(), ref["fence_rgb"].tolist())}
        got = cl["left_rgb"][b, :nl].cpu().numpy()
        pts = cl["left_xyz"][b, :nl].cpu().numpy()
        assert all(src[tuple(p)] == tuple(c) for p, c in zip(pts[::97].tolist(), got[::97].tolist()))


@pytest.mark.parametrize("prec,flip_tol,disp_tol,n_tol,w_tol", [("bf16x2", 1e-4, 1e-4, 1e-4, 0.05), ("plan", 5e-4, 1e-3, 1e-3, 0.05)])
def test_split_engine_records_track_the_exact_f32_engine(prec, flip_tol, disp_tol, n_tol, w_tol):
    """mask pixels near softmax = 0.5 can flip between precisions and move n_road / the width.  Same frames through the
    f32 engine and the split engine / the built-in precision plan (what bench.py measures), explicit tolerances on the records."""
    B = 4
    frames_np = _smooth_frames(B, seed=7)
    fr = dev(frames_np)
    res = {}
    cam = None
    for p_ in ("f32", prec):
        eng, _, _ = engine(H, W, 32, "resnet50", fcn_kw=dict(decoder_std=0.05), precision=p_)
        if cam is None:
            cam = _camera_at_10m(eng.monodepth_forward(fr))
        out = eng.process_batch(fr, [cam] * B, RoadWidthParams())
        res[p_] = (Engine.records(out["records"]), out["seg"]["road"].clone(), out["disp_pp"].clone())
    a, b = res["f32"][0], res[prec][0]
    flips = float((res["f32"][1] != res[prec][1]).float().mean())
    print(prec, "mask flips", flips, "n_road", a["n_road"], b["n_road"], "n_ror", a["n_ror"], b["n_ror"], "width", a["width"], b["width"])
    assert flips < flip_tol                                                    # fraction of pixels whose road mask differs
    assert relerr(res[prec][2].cpu().numpy(), res["f32"][2].cpu().numpy()) < disp_tol
    assert (np.abs(a["n_road"] - b["n_road"]) <= np.maximum(3, n_tol * a["n_road"])).all()
    assert (np.abs(a["n_ror"] - b["n_ror"]) <= np.maximum(20, 50 * n_tol * a["n_ror"])).all()
    assert (a["found"] == b["found"]).all()
    ok = a["found"] != 0
    assert ok.any() and np.abs(a["width"][ok] - b["width"][ok]).max() < w_tol  # metres, at ~10 m depth


# ------------------------------------------------------------------------------------------------ the reference-shaped boundary
@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_process_frame_full_size_against_the_oracle(precision):
    """BASELINE configs[0]'s substitute at the size it names: ONE 512 x 1024 frame through api.FrameProcessor.process_frame (the
    reference's per-frame operator, semantic_depth.py:98-334) on the headline engine.  Networks against the CPU oracle (logits -> masks
    as a mismatch fraction, disparity within 1e-3 and the strict bound); the exact tail -- back-projection, mask gather, road chain,
    width -- bit for bit against oracle.pipeline.frame_tail fed the engine's own raw disparity pair and masks."""
    from semantic_depth_amd import api
    wf = Wt.make_fcn8s_weights(11, decoder_std=0.05, bias_std=0.1)
    wm = Wt.make_monodepth_weights("resnet50", 12, bias_std=0.05)
    depther = api.DepthFrame(False, "resnet50", H, W, wm, None, precision=precision)
    segmenter = api.SegmentFrame((H, W), wf, True, False, "0", precision=precision)
    assert segmenter.engine is depther.engine and depther.engine.precision == precision
    frame = _smooth_frames(1, seed=77)[0]
    road, fence, _ = segmenter.segment_frame(frame)
    lg = nets.fcn8s_forward(frame[None], wf)
    _, road_r, fence_r, _ = nets.softmax_masks(lg[0])
    assert float((road[..., 0] != road_r).mean()) < 2e-3 and float((fence[..., 0] != fence_r).mean()) < 2e-3
    disp = depther.compute_disparity(frame)
    ref_disp = nets.compute_disparity(frame, wm, "resnet50")
    assert_close(disp, ref_disp, precision, TOL, "disparity of the frame", kind="disp")
    depther.f = 10.0 * float(np.median(disp)) * W                                   # the driver reassigns .f (:859): points at the measuring depth
    fp = api.FrameProcessor(segmenter, depther, depth=10.0, approach="rw")
    res = fp.process_frame(frame, want_clouds=True)
    assert np.array_equal(res["road_mask"], road[..., 0]) and np.array_equal(res["disparity"], disp * np.float32(W))
    cam_d = dict(cx=depther.cx, cy=depther.cy, f=depther.f, b=depther.b, disp_mult=float(W))
    _, raw = depther.engine.monodepth_forward(dev(frame[None]), want_raw=True)
    ref = pipeline.frame_tail(raw[0].cpu().numpy(), road[..., 0], fence[..., 0], frame, cam_d, pipeline.RoadWidthParams())
    assert np.array_equal(res["road3D"], ref["road3d"]) and np.array_equal(res["road_colors"], ref["road_rgb"])
    rec = res["record"]
    assert int(rec["n_road"]) == ref["road3d"].shape[0]
    for k in ("n_zcut", "n_mad_y", "n_mad_x", "n_plane", "n_sor", "n_ror"):
        assert int(rec[k]) == int(ref["rw"][k]), k
    assert bool(rec["found"]) == ref["rw"]["found"]
    if ref["rw"]["found"]:
        assert res["dist_rw"] == ref["rw"]["width"]
    assert np.array_equal(res["road3D_final"].astype(np.float64), ref["rw"]["points"])
    print("512x1024 process_frame", precision, "road points", int(rec["n_road"]), "after the chain", int(rec["n_ror"]), "width", res.get("dist_rw"))


def test_api_classes_built_like_the_reference_main(tmp_path):
    """semantic_depth.py:773-789: DepthFrame and SegmentFrame are constructed independently (no shared-engine argument), then
    handed to FrameProcessor; process_frame on a frame that is NOT the network size (cubic resize on the GPU, :111)."""
    from semantic_depth_amd import api, outputs, pcl
    from oracle import resize as oresize
    h, w = 128, 256
    wf = Wt.make_fcn8s_weights(1, decoder_std=0.05, bias_std=0.1)
    wm = Wt.make_monodepth_weights("vgg", 2, gain=1.0, bias_std=0.05)
    frame_depther = api.DepthFrame(False, "vgg", h, w, wm, None)                       # positional, like the reference
    frame_segmenter = api.SegmentFrame((h, w), wf, True, False, "0")
    assert frame_depther.f == 380 and frame_depther.cx == 314.05519001                 # :600-607
    fp = api.FrameProcessor(frame_segmenter, frame_depther, depth=10.0, approach="both")
    assert frame_segmenter.engine is frame_depther.engine                              # ONE engine, not two
    big = _smooth_frames(1, 2 * h, 2 * w, seed=3)[0]
    small = oresize.resize_cubic_u8(big, h, w)
    # operator by operator (semantic_depth.py:121,144,160)
    road, fence, overlay = frame_segmenter.segment_frame(small)
    assert road.shape == (h, w, 1) and road.dtype == bool and overlay.shape == (h, w, 3)
    lg = nets.fcn8s_forward(small[None], wf)
    _, road_r, fence_r, _ = nets.softmax_masks(lg[0])
    assert float((road[..., 0] != road_r).mean()) < 2e-3 and float((fence[..., 0] != fence_r).mean()) < 2e-3
    disp = frame_depther.compute_disparity(small)
    ref_disp = nets.compute_disparity(small, wm, "vgg")
    assert disp.dtype == np.float32 and relerr(disp, ref_disp) < TOL
    pts = frame_depther.compute_3D_points(disp * np.float32(2 * w))
    Q = fusion.make_Q(frame_depther.cx, frame_depther.cy, frame_depther.f, frame_depther.b)
    assert np.array_equal(pts, fusion.reproject(disp * np.float32(2 * w), Q), equal_nan=True)
    # the whole frame: focal length from the net's own median disparity so that the chain has points at 10 m
    frame_depther.f = 10.0 * float(np.median(disp)) * 2 * w                            # the driver reassigns .f (:859)
    res = fp.process_frame(big, want_clouds=True)                                      # original width = 2w scales the disparity (:109)
    assert np.array_equal(res["road_mask"], road[..., 0]) and np.array_equal(res["disparity"], disp * np.float32(2 * w))
    cam_d = dict(cx=frame_depther.cx, cy=frame_depther.cy, f=frame_depther.f, b=frame_depther.b, disp_mult=2.0 * w)
    e = frame_depther.engine
    _, raw = e.monodepth_forward(dev(small[None]), want_raw=True)
    ref = pipeline.frame_tail(raw[0].cpu().numpy(), road[..., 0], fence[..., 0], small, cam_d, pipeline.RoadWidthParams(ror_n=8, sor_k=10))
    # (the default ror_n = 80 empties a 128 x 256 cloud; the FrameProcessor was built with the defaults, so rebuild with ror_n = 8)
    fp8 = api.FrameProcessor(frame_segmenter, frame_depther, depth=10.0, approach="both", params=api.RoadWidthParams(ror_n=8))
    res = fp8.process_frame(big, want_clouds=True)
    assert np.array_equal(res["road3D"], ref["road3d"]) and np.array_equal(res["road_colors"], ref["road_rgb"])
    assert int(res["record"]["n_ror"]) == ref["rw"]["n_ror"] and bool(res["record"]["found"]) == ref["rw"]["found"]
    if ref["rw"]["found"]:
        assert res["dist_rw"] == ref["rw"]["width"]
    assert np.array_equal(res["road3D_final"].astype(np.float64), ref["rw"]["points"])
    assert res["f2f_record"] is not None and int(res["f2f_record"]["counts"][0]) == ref["fence3d"].shape[0]
    # the file outputs of --save_data (semantic_depth.py:339-458) from that result
    if res["record"]["found"]:
        files = outputs.save_frame_outputs(str(tmp_path / "frame_output"), res, 10.0, approach="both", segmented_frame=overlay,
                                           times={k: 0.0 for k in outputs.TIME_KEYS})
        names = {os.path.basename(f) for f in files}
        assert {"frame_output_ROAD.ply", "frame_output.ply", "frame_output_times.txt", "frame_output_distances.txt", "frame_output.png",
                "frame_output_only_segmentation.png"} <= names
        # the combined cloud in the reference's order (:421-434): road, road plane, rw line, left / right fence, their two planes, f2f line
        from semantic_depth_amd.point_cloud_2_ply import PointCloud2Ply
        rp, rcp = outputs.road_plane_grid(res["road3D"], res["road_colors"])
        assert rp is not None and (rcp == 200).all()
        rec = res["record"]
        line_rw, cl_rw = pcl.create_3Dline_from_3Dpoints(rec["left_pt"].astype(np.float64)[None, :].copy(), rec["right_pt"].astype(np.float64)[None, :].copy(), [250, 0, 0])
        line_rw[:, 2] += 0.2
        want = PointCloud2Ply(res["road3D_final"].astype(np.float64), res["road_colors_final"], str(tmp_path / "want"))
        want.add_extra_point_cloud(rp, rcp)
        want.add_extra_point_cloud(line_rw, cl_rw)
        if "fence3D_left" in res:
            want.add_extra_point_cloud(res["fence3D_left"], res["fence_left_colors"])
            want.add_extra_point_cloud(res["fence3D_right"], res["fence_right_colors"])
            for g, gc in zip(*[iter(outputs.fence_plane_grids(res["fence3D"], res["fence_colors"]))] * 2):
                if g is not None:
                    assert (gc == np.array([40, 70, 40])).all()
                    want.add_extra_point_cloud(g, gc)
            f2 = res["f2f_record"]
            want.add_extra_point_cloud(*pcl.create_3Dline_from_3Dpoints(f2["left_pt"][None, :].copy(), f2["right_pt"][None, :].copy(), [0, 255, 0]))
        want.prepare_and_save_point_cloud()
        assert open(tmp_path / "want.ply").read() == open(tmp_path / "frame_output.ply").read()
        # the overlay resized back to the original frame size before the banner is drawn (:341-345)
        from semantic_depth_amd import frame_io
        files2 = outputs.save_frame_outputs(str(tmp_path / "big"), res, 10.0, approach="rw", segmented_frame=overlay, original_size=(2 * h, 2 * w))
        seg_big = frame_io.imread(str(tmp_path / "big_only_segmentation.png"))
        assert seg_big.shape == (2 * h, 2 * w, 3) and np.array_equal(seg_big, oresize.resize_cubic_u8(overlay, 2 * h, 2 * w))
        assert str(tmp_path / "big.png") in files2 and frame_io.imread(str(tmp_path / "big.png")).shape == (2 * h, 2 * w, 3)
        assert open(tmp_path / "frame_output_distances.txt").read().startswith("rw distance:    {}\n".format(res["dist_rw"]))
        import json as _json
        outputs.append_metrics_jsonl(str(tmp_path / "metrics.jsonl"), "frame_output", res, times={k: 0.0 for k in outputs.TIME_KEYS})
        outputs.append_metrics_jsonl(str(tmp_path / "metrics.jsonl"), "frame_output_again", res)
        lines = [_json.loads(l) for l in open(tmp_path / "metrics.jsonl")]
        assert len(lines) == 2 and lines[0]["dist_rw"] == res["dist_rw"] and lines[0]["points"]["n_ror"] == int(res["record"]["n_ror"])
        assert lines[0]["fence"]["counts"][0] == int(res["f2f_record"]["counts"][0]) and "times_s" in lines[0] and "times_s" not in lines[1]
        head = open(tmp_path / "frame_output_ROAD.ply").read().split("\n")
        assert head[0] == "ply" and head[2].strip().startswith("element vertex")
    # DepthFrame.disp_to_image (semantic_depth.py:681-683): a gray PNG of the original frame size, min -> 0, max -> 255
    from semantic_depth_amd import frame_io
    pth = frame_depther.disp_to_image(disp, str(tmp_path / "frame_output"), 2 * h, 2 * w)
    dimg = frame_io.imread(pth)
    assert pth.endswith("frame_output_disp.png") and dimg.shape == (2 * h, 2 * w, 3) and dimg.min() == 0 and dimg.max() == 255
    iy, ix = np.unravel_index(int(np.argmax(disp)), disp.shape)
    assert dimg[2 * iy:2 * iy + 2, 2 * ix:2 * ix + 2, 0].max() >= 250
    # two DepthFrames of one geometry share the Engine; each computes with ITS OWN checkpoint whatever was loaded last
    wm2 = Wt.make_monodepth_weights("vgg", 5, gain=1.0, bias_std=0.05)
    other = api.DepthFrame(False, "vgg", h, w, wm2, None)
    assert other._engine is frame_depther._engine
    d_other = other.compute_disparity(small)
    assert relerr(d_other, nets.compute_disparity(small, wm2, "vgg")) < TOL
    assert np.array_equal(frame_depther.compute_disparity(small), disp)
    # the pcl drop-in shares that engine and returns the visualisation plane like the reference
    pcl._engine = None                      # (another test module may have pinned its own engine with pcl.set_engine)
    assert pcl._eng() in [x for per in api._engines.values() for x in per.values()]
    p2, c2, plane3D, colors_plane, coeff = pcl.remove_noise_by_fitting_plane(ref["road3d"], ref["road_rgb"], axis=1, threshold=5.0,
                                                                             plane_color=[200, 200, 200])
    assert plane3D is not None and plane3D.shape[1] == 3 and colors_plane.shape == plane3D.shape and (colors_plane == 200).all()


def test_run_sequence_on_one_gpu_equals_process_batch():
    """distributed.run_sequence + make_engine_step (world 1): 1024x2048 frames -> GPU cubic resize -> whole path, in chunks of 3,
    equals process_batch on the resized frames (config 5's driver; the 2-rank form is covered on CPU with gloo)"""
    from semantic_depth_amd.distributed import make_engine_step, run_sequence
    eng, _, _ = engine(H, W, 32, "resnet50", fcn_kw=dict(decoder_std=0.05), precision="bf16x2")
    big = _smooth_frames(5, 2 * H, 2 * W, seed=9)
    small = eng.resize_cubic(dev(big))
    cam = Camera(1048.64 / 2, 519.277 / 2, 1000.0, 1.0, 3800.0)
    want = eng.process_batch(small, [cam] * 5, RoadWidthParams())["records"].clone()
    got = run_sequence(lambda lo, hi: big[lo:hi], 5, make_engine_step(eng, lambda i: cam), batch=3, device="cuda")
    assert torch.equal(got, want)
