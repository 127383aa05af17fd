"""per-layer conv timings (dev tool): SEMDEPTH_PROFILE_VERBOSE=1 python scripts/layer_times.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SEMDEPTH_PROFILE_VERBOSE"] = "1"
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
enc = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
H, W = 512, 1024
eng = Engine(H, W, B, enc, precision=prec)
eng.load_weights(L.SD_NET_FCN8S, Wt.make_fcn8s_weights(1, decoder_std=0.05))
eng.load_weights(L.SD_NET_MONODEPTH, Wt.make_monodepth_weights(enc, 2))
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
for _ in range(2):
    eng.fcn8s_forward(fr); eng.monodepth_forward(fr)
torch.cuda.synchronize()
eng.profile(True)
eng.fcn8s_forward(fr)
print("=== FCN-8s", file=sys.stderr)
b1 = eng.profile_read()
eng.monodepth_forward(fr)
print("=== monodepth", file=sys.stderr)
b2 = eng.profile_read()
for name, bs in (("fcn", b1), ("mono", b2)):
    ms = sum(b["ms"] for b in bs); fl = sum(b["flops"] for b in bs)
    print(name, "conv ms", round(ms, 2), "TF/s", round(fl / ms / 1e9, 2), file=sys.stderr)
