"""Input-stage feed rate (dev tool, run on the GPU box): can the host keep the GPU supplied at the benchmarked
frames/s?  Measures, for Cityscapes-sized 1024 x 2048 frames:
  decode    frame_io.imread (zlib inflate + sd_png_unfilter_bgr) per core and with a thread pool
  upload    pinned host -> HBM copy of decoded frames
  resize    Engine.resize_cubic 1024x2048 -> 512x1024 on the GPU
  feeder    frame_io.FrameFeeder end to end (decode + pinned upload, one batch ahead)
    python scripts/feed_rate.py [--frames 256] [--workers 128] [--out profiles/r03_feed_rate.json]
"""
import argparse
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from semantic_depth_amd import frame_io, outputs            # noqa: E402
from semantic_depth_amd.engine import Engine                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--workers", type=int, default=min(128, os.cpu_count() or 8))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_feed_rate.json"))
    a = ap.parse_args()
    H, W = 1024, 2048
    rng = np.random.default_rng(0)
    res = {"frame": [H, W, 3], "host_cpus": os.cpu_count(), "workers": a.workers}
    with tempfile.TemporaryDirectory() as td:
        # street-like content: smooth gradients + texture noise (PNG of pure noise does not compress; real frames do, ~2.2 MB each)
        yy, xx = np.mgrid[0:H, 0:W]
        paths = []
        for i in range(a.frames):
            base = np.stack([(yy // 3 + xx // 5 + 7 * i) % 256, (xx // 4 + 3 * i) % 256, (yy // 2 + xx // 7) % 256], -1).astype(np.uint8)
            img = base ^ rng.integers(0, 4, (H, W, 3), dtype=np.uint8)
            paths.append(outputs.write_png(os.path.join(td, f"f{i:05d}.png"), img, level=6))
        res["png_bytes_mean"] = float(np.mean([os.path.getsize(p) for p in paths]))
        t0 = time.perf_counter()
        for p in paths[:8]:
            frame_io.imread(p)
        res["decode_ms_per_frame_1_thread"] = (time.perf_counter() - t0) / 8 * 1e3
        with ThreadPoolExecutor(a.workers) as ex:
            list(ex.map(frame_io.imread, paths[:a.workers]))            # warm
            t0 = time.perf_counter()
            list(ex.map(frame_io.imread, paths))
            res["decode_fps_python_thread_pool"] = a.frames / (time.perf_counter() - t0)
        # the native batch reader (what FrameFeeder calls): all files -> one host buffer, `workers` C++ threads
        import ctypes as C
        from semantic_depth_amd import _lib as L
        lib = L.load()
        hostbuf = np.empty((a.frames, H, W, 3), np.uint8)
        arr = (C.c_char_p * a.frames)(*[p.encode() for p in paths])
        for _ in range(2):
            t0 = time.perf_counter()
            st = lib.sd_decode_files_bgr(arr, a.frames, H, W, hostbuf.ctypes.data_as(C.c_void_p), H * W * 3, a.workers, None)
            dt = time.perf_counter() - t0
        assert st == 0
        res["decode_fps_native_batch"] = a.frames / dt
        if torch.cuda.is_available():
            host = torch.empty((32, H, W, 3), dtype=torch.uint8, pin_memory=True)
            dev = torch.empty((32, H, W, 3), dtype=torch.uint8, device="cuda")
            dev.copy_(host, non_blocking=True); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                dev.copy_(host, non_blocking=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            res["upload_gb_per_s"] = host.numel() / dt / 1e9
            res["upload_fps"] = 32 / dt
            eng = Engine(512, 1024, 32, "resnet50", precision="bf16x2")
            for fr, lo in frame_io.FrameFeeder(paths[:64], 32, "cuda", a.workers):        # warm (pinned staging allocation)
                pass
            eng.resize_cubic(dev); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                eng.resize_cubic(dev)
            torch.cuda.synchronize()
            res["resize_fps"] = 32 * 5 / (time.perf_counter() - t0)
            t0 = time.perf_counter()
            n = 0
            for fr, lo in frame_io.FrameFeeder(paths, 32, "cuda", a.workers):
                eng.resize_cubic(fr)
                n += fr.shape[0]
            torch.cuda.synchronize()
            res["feeder_fps_decode_upload_resize"] = n / (time.perf_counter() - t0)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
