"""dev tool: error (vs the exact-f32 engine, 4 frames of 512x1024) and conv time (32 frames) of a precision plan
    python scripts/try_plan.py "<fcn layers>" "<monodepth layers>" """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
H, W = 512, 1024
fcn, mono = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(123)
def frames(B):
    base = rng.integers(0, 256, (B, H // 8, W // 8, 3), dtype=np.uint8)
    fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    return torch.from_numpy((fr.astype(np.int16) + rng.integers(-16, 17, fr.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)).cuda()
wf = Wt.make_fcn8s_weights(1, decoder_std=0.05); wm = Wt.make_monodepth_weights("resnet50", 2)
fr4 = frames(4)
def outs(precision, plan=None, B=4):
    e = Engine(H, W, B, "resnet50", precision=precision, plan=plan)
    e.load_weights(L.SD_NET_FCN8S, wf); e.load_weights(L.SD_NET_MONODEPTH, wm)
    return e
e32 = outs("f32"); lg32 = e32.fcn8s_forward(fr4, want_logits=True)["logits"].clone(); d32 = e32.monodepth_forward(fr4).clone(); e32.close()
ep = outs("plan", (fcn, mono)); lg = ep.fcn8s_forward(fr4, want_logits=True)["logits"]; d = ep.monodepth_forward(fr4)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
print("plan:", ep.precision_plan())
print(f"error vs f32 (4 frames): logits {rel(lg, lg32):.3e}  disparity {rel(d, d32):.3e}")
ep.close()
e = outs("plan", (fcn, mono), B=32); fr = frames(32)
for _ in range(2): e.fcn8s_forward(fr); e.monodepth_forward(fr)
torch.cuda.synchronize(); e.profile(True)
e.fcn8s_forward(fr); b1 = e.profile_read(); e.monodepth_forward(fr); b2 = e.profile_read()
print("conv ms per 32 frames: fcn %.2f  mono %.2f" % (sum(b["ms"] for b in b1), sum(b["ms"] for b in b2)))
