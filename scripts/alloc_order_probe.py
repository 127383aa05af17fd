"""dev probe: does the ORDER of the large arena allocations change the speed of HBM-bound kernels?  (round 3: the plan engine built
after the f32 engine ran its stem / ResNet 1x1 layers 2-3x slower than when it is built first)
    python scripts/alloc_order_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
H, W, B = 512, 1024, 32
wf = Wt.make_fcn8s_weights(1, decoder_std=0.05)
wm = Wt.make_monodepth_weights("resnet50", 2)
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()

def stream_gbs(t):
    """GB/s of a device-to-device copy over a [n] uint8 tensor half -> half"""
    n = t.numel() // 2
    a, b = t[:n], t[n:2 * n]
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    return 2 * n * 5 / (time.perf_counter() - t0) / 1e9

def time_plan(eng, label):
    for _ in range(2):
        eng.fcn8s_forward(fr); eng.monodepth_forward(fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.fcn8s_forward(fr); eng.monodepth_forward(fr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    eng.profile(True); eng.fcn8s_forward(fr); eng.monodepth_forward(fr)
    bk = {b["kernel"]: b["ms"] for b in eng.profile_read()}
    eng.profile(False)
    print(f"{label}: nets {ms:.2f} ms per {B} frames; ws ptr {eng._ws.data_ptr():#x} ({eng._ws.data_ptr() % (1 << 21)} mod 2 MiB); stream copy over the workspace "
          f"{stream_gbs(eng._ws):.0f} GB/s; conv_stem {bk.get('conv_stem_kernel', 0):.3f} ms, dma<2,4,2,2> {bk.get('conv_dma_f16x1_kernel<2,4,2,2>', 0):.3f} ms", flush=True)

def mk():
    e = Engine(H, W, B, "resnet50", precision="plan")
    e.load_weights(L.SD_NET_FCN8S, wf); e.load_weights(L.SD_NET_MONODEPTH, wm)
    return e

e1 = mk(); time_plan(e1, "plan engine, first allocation")
big = torch.zeros(19 * 2**30, dtype=torch.uint8, device="cuda")
print(f"19 GiB tensor at {big.data_ptr():#x}; stream copy {stream_gbs(big):.0f} GB/s")
e2 = mk(); time_plan(e2, "plan engine, allocated after a 19 GiB tensor")
time_plan(e1, "first engine again")
del e2, big; torch.cuda.empty_cache()
e3 = mk(); time_plan(e3, "plan engine, after freeing both")
e32 = Engine(H, W, B, "resnet50", precision="f32"); e32.load_weights(L.SD_NET_FCN8S, wf); e32.load_weights(L.SD_NET_MONODEPTH, wm)
e32.fcn8s_forward(fr); e32.monodepth_forward(fr); torch.cuda.synchronize()
e4 = mk(); time_plan(e4, "plan engine, after an f32 engine ran")
time_plan(e3, "third engine again")
