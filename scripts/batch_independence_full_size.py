"""Does a frame's result depend on the batch it is computed in AT FULL SIZE?  (tests/test_gpu_nets.py::test_batch_and_chunk_independence runs 128 x 256, where no
layer reaches the 256 x 256 phased block.)  512 x 1024, engine of 8: frames 0..7 in one call against frame 0 alone and frames 0..1 -- bit equality per network.
    python scripts/batch_independence_full_size.py [precision ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
H, W, B = 512, 1024, 8
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
wf = Wt.make_fcn8s_weights(1, decoder_std=0.05)
wm = Wt.make_monodepth_weights("resnet50", 2)
for prec in (sys.argv[1:] or ["bf16x3", "f16x2", "f32", "plan"]):
    eng = Engine(H, W, B, "resnet50", precision=prec)
    eng.load_weights(L.SD_NET_FCN8S, wf)
    eng.load_weights(L.SD_NET_MONODEPTH, wm)
    lg8 = eng.fcn8s_forward(fr, want_logits=True)["logits"].clone()
    _, raw8 = eng.monodepth_forward(fr, want_raw=True)
    raw8 = raw8.clone()
    for n in (1, 2):
        lg = eng.fcn8s_forward(fr[:n].contiguous(), want_logits=True)["logits"]
        _, raw = eng.monodepth_forward(fr[:n].contiguous(), want_raw=True)
        dl = float((lg[0] - lg8[0]).abs().max() / lg8[0].abs().max())
        dd = float((raw[0] - raw8[0]).abs().max() / raw8[0].abs().max())
        print(f"{prec}: frame 0 in a call of {n} against a call of 8: logits {'bit-equal' if torch.equal(lg[0], lg8[0]) else f'differ by {dl:.2e}'}, "
              f"raw disparity {'bit-equal' if torch.equal(raw[0], raw8[0]) else f'differ by {dd:.2e}'}", flush=True)
    del eng
