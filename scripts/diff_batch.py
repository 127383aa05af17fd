"""dev tool: first monodepth layer whose output for image 0 differs between a 1-frame and a 4-frame network pass"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SEMDEPTH_PROFILE_VERBOSE"] = "1"
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
H, W = 128, 256
prec = sys.argv[1] if len(sys.argv) > 1 else "plan"
rng = np.random.default_rng(8)
fr = torch.from_numpy(rng.integers(0, 256, (4, H, W, 3), dtype=np.uint8)).cuda()
wm = Wt.make_monodepth_weights("resnet50", 2)
names = ["enc/conv1"]
for s, nb in zip((2, 3, 4, 5), (3, 4, 6, 3)):
    for b in range(1, nb + 1):
        names += [f"enc/res{s}_{b}/conv{i}" for i in (1, 2, 3)]
for l in (6, 5, 4, 3, 2, 1):
    names += [f"dec/upconv{l}", f"dec/iconv{l}"] + ([f"dec/disp{l}"] if l <= 4 else [])
def run(B):
    e = Engine(H, W, B, "resnet50", precision=prec)
    e.load_weights(L.SD_NET_FCN8S, Wt.make_fcn8s_weights(1, decoder_std=0.05)); e.load_weights(L.SD_NET_MONODEPTH, wm)
    e.profile(True)
    e.monodepth_forward(fr[:B].contiguous())
    out = {}
    for n in names:
        try: out[n] = e.net_tensor(L.SD_NET_MONODEPTH, n).cpu().numpy()
        except Exception as ex: out[n] = None
    e.close()
    return out
a, b = run(1), run(4)
for n in names:
    if a[n] is None or b[n] is None: print(n, "n/a"); continue
    # images of frame 0: index 0 and B (the flipped half follows the straight half?) -> compare image 0 only
    same = np.array_equal(a[n][0], b[n][0])
    print(f"{n:24s} {'same' if same else 'DIFF  max|d| %.3e' % np.abs(a[n][0] - b[n][0]).max()}")
