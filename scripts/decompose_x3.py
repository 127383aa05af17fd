"""[needs a dev build of the library: SEMDEPTH_DEV_BUILD=1 python -m semantic_depth_amd.build --force -- the shipped library carries no decomposition copies]
Decomposition of the bf16x3 conv kernels WITHOUT disturbing their inputs (dev tool, round 5).

SEMDEPTH_X3_DIAG (1 = no output stores, 2 = conv_dma3: no epilogue at all / conv_direct3: no MFMAs) is latched per handle, and a handle
that does not store its outputs feeds zeros to every later layer -- zeros draw less MFMA power, the chip clocks up, and the "no stores"
column of scripts/layer_times.py then mixes the store cost with a data effect (monodepth -22 %, FCN-8s 0 % in gpurun_out/r05a).
Here engine A runs normally and fills its arenas; engines B1 / B2 (the switch set) are bound to A's arenas, so every layer of their
profiled pass reads the real activations A left behind and writes nothing.

    python scripts/decompose_x3.py [B [bf16x3|f16x2]] 2> table"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SEMDEPTH_PROFILE_VERBOSE"] = "1"
import numpy as np
import torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine, _ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"       # "f16x2": SEMDEPTH_X3_DIAG=2 = the H2 direct kernel without its epilogue
DIAGS = (1, 2) if PREC == "bf16x3" else (2,)
H, W, enc = 512, 1024, "resnet50"
wf = Wt.make_fcn8s_weights(1, decoder_std=0.05)
wm = Wt.make_monodepth_weights(enc, 2)
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()


def make(diag):
    if diag:
        os.environ["SEMDEPTH_X3_DIAG"] = str(diag)
    try:
        e = Engine(H, W, B, enc, precision=PREC)
    finally:
        os.environ.pop("SEMDEPTH_X3_DIAG", None)
    return e


A = make(0)
A.load_weights(L.SD_NET_FCN8S, wf)
A.load_weights(L.SD_NET_MONODEPTH, wm)


def layers(eng, fill):
    """per-layer (name, ms) of one profiled pass of each network; `fill` runs A first so that the arenas hold real activations"""
    import io, contextlib
    out = {}
    for net, fwd in (("fcn", lambda e: e.fcn8s_forward(fr)), ("mono", lambda e: e.monodepth_forward(fr))):
        fill and fwd(A)
        torch.cuda.synchronize()
        eng.profile(True)
        fwd(eng)
        eng.profile_read()          # (prints the per-layer lines to stderr under SEMDEPTH_PROFILE_VERBOSE)
        eng.profile(False)
    return out


for _ in range(2):
    A.fcn8s_forward(fr); A.monodepth_forward(fr)
torch.cuda.synchronize()
print("=== diag 0 (engine A)", file=sys.stderr)
layers(A, False)
for d in DIAGS:
    Bn = make(d)
    # bind B to A's arenas (same plan, same layout; the tables it uploads are the ones already there) and mark its weights loaded
    L.check(Bn.lib, Bn.h, Bn.lib.sd_bind_memory(Bn.h, _ptr(A._wf), _ptr(A._wm), _ptr(A._ws)), "sd_bind_memory")
    Bn.load_weights(L.SD_NET_FCN8S, wf)
    Bn.load_weights(L.SD_NET_MONODEPTH, wm)
    torch.cuda.synchronize()
    print(f"=== diag {d} on A's arenas", file=sys.stderr)
    layers(Bn, True)
    del Bn
