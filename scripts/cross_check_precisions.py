"""The split-bf16 engine against the exact f32 engine (conv_igemm: plain f32 MFMA, no specialised kernel) on frame shapes the
CPU oracle is too slow for: every kernel choice the planner makes at these shapes (direct conv passes, source-resolution
upconv tiles, sub-plane hand-off, 256x256 DMA blocks, fallbacks where a width is not a multiple of 32) must agree with the
f32 engine to the split format's precision.  usage: python scripts/cross_check_precisions.py [precision]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine

def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x2"
worst = 0.0
for (H, W, B, enc) in [(512, 1024, 3, "resnet50"), (512, 1024, 2, "vgg"), (384, 1280, 2, "resnet50"), (256, 512, 5, "resnet50"),
                       (192, 640, 2, "resnet50"), (384, 1024, 1, "vgg"), (128, 2048, 2, "resnet50")]:
    wf = Wt.make_fcn8s_weights(1, decoder_std=0.05, bias_std=0.1)
    wm = Wt.make_monodepth_weights(enc, 2, bias_std=0.05)
    fr = torch.from_numpy(np.random.default_rng(H + W).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
    res = {}
    for p in ("f32", prec):
        eng = Engine(H, W, B, enc, precision=p)
        eng.load_weights(L.SD_NET_FCN8S, wf)
        eng.load_weights(L.SD_NET_MONODEPTH, wm)
        lg = eng.fcn8s_forward(fr, want_logits=True)["logits"].clone()
        _, raw = eng.monodepth_forward(fr, want_raw=True)
        res[p] = (lg, raw.clone())
        del eng
    e_fcn, e_mono = rel(res[prec][0], res["f32"][0]), rel(res[prec][1], res["f32"][1])
    worst = max(worst, e_fcn, e_mono)
    print(f"{H}x{W} B={B} {enc}: {prec} vs f32  logits {e_fcn:.2e}  disparities {e_mono:.2e}", flush=True)
print("worst", f"{worst:.2e}")
sys.exit(0 if worst < 1e-3 else 1)
