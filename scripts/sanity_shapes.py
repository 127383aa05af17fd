import sys, numpy as np, torch
sys.path.insert(0, '.')
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Camera, Engine, RoadWidthParams
for (H, W, B, enc, prec) in [(256, 512, 3, "resnet50", "bf16x2"), (256, 512, 2, "vgg", "bf16x2"), (384, 1280, 2, "resnet50", "mixed"), (512, 1024, 8, "resnet50", "f32")]:
    eng = Engine(H, W, B, enc, precision=prec)
    eng.load_weights(L.SD_NET_FCN8S, Wt.make_fcn8s_weights(1, decoder_std=0.05))
    eng.load_weights(L.SD_NET_MONODEPTH, Wt.make_monodepth_weights(enc, 2))
    fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
    out = eng.process_batch(fr, [Camera(W / 2, H / 2, 2000.0, 1.0, float(W))] * B, RoadWidthParams())
    torch.cuda.synchronize()
    rec = Engine.records(out["records"])
    print(H, W, B, enc, prec, "disp range", float(out["disp_pp"].min()), float(out["disp_pp"].max()), "road frac", float(out["seg"]["road"].float().mean()), "n_road", rec["n_road"].tolist(), "finite", bool(torch.isfinite(out["disp_pp"]).all()))
    del eng
