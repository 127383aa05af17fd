"""[needs a dev build of the library: SEMDEPTH_DEV_BUILD=1 python -m semantic_depth_amd.build --force -- the shipped library carries no decomposition copies]
Where does a phase of conv_dma3 go?  SEMDEPTH_X3_DIAG=3 launches the TIMED copy of the 1x1 form (conv_dma3_kernel<1, HS, true>): s_memtime stamps around
the counted s_waitcnt, the s_barrier and the body of every phase, printed by waves 0 and 4 of the middle workgroup of every launch (dev tool, round 5).
    python scripts/dma3_timed.py [bf16x3|f16x2] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SEMDEPTH_X3_DIAG", "3")        # (7: the timed copies without their DMA pieces -- what the hooks cost)
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H, W = 512, 1024
eng = Engine(H, W, B, "resnet50", precision=prec)
eng.load_weights(L.SD_NET_FCN8S, Wt.make_fcn8s_weights(1, decoder_std=0.05))
eng.load_weights(L.SD_NET_MONODEPTH, Wt.make_monodepth_weights("resnet50", 2))
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
for _ in range(2):
    eng.fcn8s_forward(fr); eng.monodepth_forward(fr)
torch.cuda.synchronize()
