"""dev tool: error of a reduced-precision engine (default: the built-in precision plan) against the exact-f32 engine over several
weight / frame seeds (4 frames of 512x1024 per seed).  Per seed: max |delta| / max |ref| (north_star's figure) of the logits and of the
raw / post-processed disparities, and the strict per-element figure |delta| / (|ref| + 1e-2 max|ref|) (p99 and max).
    python scripts/plan_error_sweep.py [n_seeds | s0,s1,...] ["fcn layers|monodepth layers"] [--precision plan|bf16x2|bf16x3|mixed]
                                       [--decoder-std 0.05] [--encoder resnet50|vgg]  > profiles/r03_plan_error_sweep.txt
--decoder-std 0.01 = the reference's own initialiser scale for the six decoder layers (fcn8s/fcn.py:161)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
ap = argparse.ArgumentParser()
ap.add_argument("seeds", nargs="?", default="8")
ap.add_argument("plan", nargs="?", default=None)
ap.add_argument("--precision", default="plan")
ap.add_argument("--decoder-std", type=float, default=0.05)
ap.add_argument("--encoder", default="resnet50")
a = ap.parse_args()
H, W, B = 512, 1024, 4
seeds = [int(t) for t in a.seeds.split(",")] if "," in a.seeds else list(range(int(a.seeds)))
n = len(seeds)
plan = tuple(a.plan.split("|")) if a.plan else None
def frames(seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (B, H // 8, W // 8, 3), dtype=np.uint8)
    fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    return torch.from_numpy((fr.astype(np.int16) + rng.integers(-16, 17, fr.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)).cuda()
e32 = Engine(H, W, B, a.encoder, precision="f32")
ep = Engine(H, W, B, a.encoder, precision=a.precision, plan=plan if a.precision == "plan" else None)
print(f"{a.precision} vs the exact-f32 engine, monodepth-{a.encoder}, decoder_std {a.decoder_std}, {n} weight / frame seeds, {B} frames of {H}x{W} each")
if a.precision == "plan":
    print("plan:", ep.precision_plan())
def rel(x, r):
    return float((x - r).abs().max() / r.abs().max())
def strict(x, r):
    d = (x.double() - r.double()).abs().flatten()
    q = d / (r.double().abs().flatten() + 1e-2 * float(r.abs().max()))
    k = max(1, int(0.99 * q.numel()))
    return float(q.kthvalue(k).values), float(q.max())
worst = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
for s in seeds:
    kw = dict(decoder_std=a.decoder_std, bias_std=0.1) if s % 2 else dict(decoder_std=a.decoder_std)
    wf = Wt.make_fcn8s_weights(100 + s, **kw)
    wm = Wt.make_monodepth_weights(a.encoder, 200 + s, **({"bias_std": 0.05} if s % 2 else {}))
    fr = frames(300 + s)
    out = []
    for e in (e32, ep):
        e.load_weights(L.SD_NET_FCN8S, wf); e.load_weights(L.SD_NET_MONODEPTH, wm)
        lg = e.fcn8s_forward(fr, want_logits=True)["logits"].clone()
        pp, raw = e.monodepth_forward(fr, want_raw=True)
        out.append((lg, raw.clone(), pp.clone()))
    el, ed, ep_ = rel(out[1][0], out[0][0]), rel(out[1][1], out[0][1]), rel(out[1][2], out[0][2])
    sl, sd_ = strict(out[1][0], out[0][0]), strict(out[1][1], out[0][1])
    worst = [max(worst[0], el), max(worst[1], max(ed, ep_)), max(worst[2], sl[0]), max(worst[3], sl[1]), max(worst[4], sd_[0]), max(worst[5], sd_[1])]
    print(f"seed {s} ({'biases' if s % 2 else 'zero biases'}): logits {el:.3e} (strict p99 {sl[0]:.2e} max {sl[1]:.2e})  raw disparity {ed:.3e} "
          f"(strict p99 {sd_[0]:.2e} max {sd_[1]:.2e})  post-processed {ep_:.3e}", flush=True)
sat = ep.saturation_count() if hasattr(ep, "saturation_count") else None
print(f"worst of {n}: logits {worst[0]:.3e}  disparity {worst[1]:.3e}  (tolerance 1e-3); strict p99 / max: logits {worst[2]:.2e} / {worst[3]:.2e}, "
      f"disparity {worst[4]:.2e} / {worst[5]:.2e}" + (f"; fp16-saturated values {sat}" if sat is not None else ""))
