"""[needs a dev build of the library: SEMDEPTH_DEV_BUILD=1 python -m semantic_depth_amd.build --force -- the shipped library carries no decomposition copies]
Where does a (tile, pass) item of conv_direct3 go?  SEMDEPTH_X3_DIAG=4 launches the TIMED copy of the dominant form (conv_direct3_kernel<2, false, 2, 2, false, true>):
s_memtime stamps around every phase's wait + barrier, the barrier in front of the epilogue and the three parts of the epilogue, summed per wave over the items of a
workgroup and printed by waves 0 and 4 (the older and the younger wave of SIMD 0) of the middle workgroup of every launch (dev tool, round 5).
    python scripts/direct3_timed.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SEMDEPTH_X3_DIAG"] = "4"
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = 512, 1024
eng = Engine(H, W, B, "resnet50", precision="bf16x3")
eng.load_weights(L.SD_NET_FCN8S, Wt.make_fcn8s_weights(1, decoder_std=0.05))
eng.load_weights(L.SD_NET_MONODEPTH, Wt.make_monodepth_weights("resnet50", 2))
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
for _ in range(2):
    eng.fcn8s_forward(fr); eng.monodepth_forward(fr)
torch.cuda.synchronize()
