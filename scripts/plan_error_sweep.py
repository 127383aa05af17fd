"""dev tool: error of the built-in precision plan against the exact-f32 engine over several weight / frame seeds
(4 frames of 512x1024 per seed; max |delta| / max |ref| of logits and raw disparity pairs)
    python scripts/plan_error_sweep.py [n_seeds | s0,s1,...] ["fcn layers|monodepth layers"] > profiles/r02_plan_error_sweep.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
H, W, B = 512, 1024, 4
arg = sys.argv[1] if len(sys.argv) > 1 else "8"
seeds = [int(t) for t in arg.split(",")] if "," in arg else list(range(int(arg)))
n = len(seeds)
plan = tuple(sys.argv[2].split("|")) if len(sys.argv) > 2 else None
def frames(seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (B, H // 8, W // 8, 3), dtype=np.uint8)
    fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    return torch.from_numpy((fr.astype(np.int16) + rng.integers(-16, 17, fr.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)).cuda()
e32 = Engine(H, W, B, "resnet50", precision="f32")
ep = Engine(H, W, B, "resnet50", precision="plan", plan=plan)
print("plan:", ep.precision_plan())
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
worst = [0.0, 0.0]
for s in seeds:
    kw = dict(decoder_std=0.05, bias_std=0.1) if s % 2 else dict(decoder_std=0.05)
    wf = Wt.make_fcn8s_weights(100 + s, **kw)
    wm = Wt.make_monodepth_weights("resnet50", 200 + s, **({"bias_std": 0.05} if s % 2 else {}))
    fr = frames(300 + s)
    out = []
    for e in (e32, ep):
        e.load_weights(L.SD_NET_FCN8S, wf); e.load_weights(L.SD_NET_MONODEPTH, wm)
        lg = e.fcn8s_forward(fr, want_logits=True)["logits"].clone()
        pp, raw = e.monodepth_forward(fr, want_raw=True)
        out.append((lg, raw.clone(), pp.clone()))
    el, ed, ep_ = rel(out[1][0], out[0][0]), rel(out[1][1], out[0][1]), rel(out[1][2], out[0][2])
    worst = [max(worst[0], el), max(worst[1], max(ed, ep_))]
    print(f"seed {s} ({'biases' if s % 2 else 'zero biases'}): logits {el:.3e}  raw disparity {ed:.3e}  post-processed {ep_:.3e}", flush=True)
print(f"worst of {n}: logits {worst[0]:.3e}  disparity {worst[1]:.3e}  (tolerance 1e-3)")
