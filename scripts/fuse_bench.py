"""fusion-stage timing (dev tool): one-pass look-back kernel vs the three-launch form, 32 frames of 512x1024, bench-like masks"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semantic_depth_amd.engine import Camera, Engine
H, W, B = 512, 1024, 32
rng = np.random.default_rng(0)
raw = torch.from_numpy((0.05 + 0.1 * rng.random((B, 2, H, W))).astype(np.float32)).cuda()
blob = rng.random((B, H // 16, W // 16))
road = torch.from_numpy(np.repeat(np.repeat(blob < 0.33, 16, 1), 16, 2).astype(np.uint8)).cuda()
fence = torch.from_numpy(np.repeat(np.repeat((blob > 0.4) & (blob < 0.73), 16, 1), 16, 2).astype(np.uint8)).cuda()
frames = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
cams = [Camera(W / 2, H / 2, 1000.0, 1.0, float(W))] * B
res = {}
for name, env in (("onepass", {}), ("three_launch", {"SEMDEPTH_DISABLE": "fuse1"})):
    os.environ.pop("SEMDEPTH_DISABLE", None)
    os.environ.update(env)
    e = Engine(H, W, B, "resnet50", precision="bf16x2")
    for mode in ("from_raw", "from_pp"):
        def run():
            if mode == "from_raw":
                return e.fuse_from_raw(road, fence, frames, cams, disp_raw=raw)
            return e.fuse_backproject(e.post_process(raw), road, fence, frames, cams)
        out = run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): out = run()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        npts = float(out["n_road"].sum() + out["n_fence"].sum())
        byts = B * H * W * 17.0 + 15.0 * npts
        res[(name, mode)] = (us, byts / us / 1e6, out)
        print(f"{name:13s} {mode:9s} {us:8.1f} us  {byts / us / 1e6:7.2f} TB/s algorithmic ({byts / 1e6:.0f} MB)", flush=True)
    e.close()
a, b = res[("onepass", "from_raw")][2], res[("three_launch", "from_pp")][2]
print("identical:", all(torch.equal(a[k], b[k]) for k in ("n_road", "n_fence")) and
      all(torch.equal(a[k][i, :int(a["n_" + k.split("_")[0]][i])], b[k][i, :int(a["n_" + k.split("_")[0]][i])]) for k in ("road_xyz", "road_rgb", "fence_xyz", "fence_rgb") for i in range(B)))
