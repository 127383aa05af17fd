"""Host-side mirror of the reference's operator interface for the hot path.

    SegmentFrame  <- semantic_depth.py:464-571   (seq:378-485)
    DepthFrame    <- semantic_depth.py:575-697   (seq:488-589)
    FrameProcessor.process_frame <- semantic_depth.py:98-268 (seq:117-238), compute steps only

Same constructor arguments / method names / return types, so the reference's FrameProcessor body reads the same
against these classes.  Differences: weights come from a dict / .npz of TF-layout arrays (the reference restores TF
checkpoints; INTEGRATION.md has the name map), both classes can share one Engine, and every method also has a
batched, device-resident form on the Engine.  No file I/O, drawing or PLY writing here (out of scope).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from .engine import Camera, Engine, RoadWidthParams


def _load_weight_arg(w):
    if isinstance(w, dict):
        return w
    if isinstance(w, str):
        z = np.load(w)
        return {k: z[k] for k in z.files}
    raise TypeError("weights must be a dict name->array or a path to an .npz")


class SegmentFrame:
    """SegmentFrame(input_shape, model_var_dir, use_frozen, use_xla, CUDA_DEVICE_NUMBER) — semantic_depth.py:464-469.
    ``model_var_dir``: dict / .npz of FCN-8s weights (weights.fcn8s_weight_shapes).  use_frozen / use_xla are accepted
    and ignored (TF graph details)."""

    def __init__(self, input_shape, model_var_dir, use_frozen=True, use_xla=False, CUDA_DEVICE_NUMBER="0", engine: Engine | None = None,
                 encoder: str = "resnet50"):
        self.input_shape = tuple(input_shape)
        self.engine = engine or Engine(self.input_shape[0], self.input_shape[1], 1, encoder, int(CUDA_DEVICE_NUMBER))
        self.engine.load_weights(L.SD_NET_FCN8S, _load_weight_arg(model_var_dir))

    def segment_frame(self, frame: np.ndarray):
        """semantic_depth.py:544-571: (road bool (H,W,1), fence bool (H,W,1), overlay u8 (H,W,3))."""
        e = self.engine
        fr = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))[None].to(e.device)
        out = e.fcn8s_forward(fr)
        road = out["road"][0].cpu().numpy().astype(bool)[..., None]
        fence = out["fence"][0].cpu().numpy().astype(bool)[..., None]
        return road, fence, self._overlay(frame, road, fence)

    @staticmethod
    def _overlay(frame, road, fence):
        # cosmetic (semantic_depth.py:557-569): RGBA paste of (128,64,128,64) on road, (160,10,10,64) on fence
        img = frame.astype(np.float32)
        for mask, colour in ((road, (128, 64, 128)), (fence, (160, 10, 10))):
            a = 64.0 / 255.0
            img = np.where(mask, img * (1 - a) + np.array(colour, np.float32) * a, img)
        return img.round().astype(np.uint8)


class DepthFrame:
    """DepthFrame(is_city, encoder, input_height, input_width, checkpoint_path, f) — semantic_depth.py:575-624.
    ``checkpoint_path``: dict / .npz of monodepth weights."""

    def __init__(self, is_city=False, encoder="vgg", input_height=256, input_width=512, checkpoint_path=None, f=None,
                 engine: Engine | None = None):
        self.is_city, self.encoder = is_city, encoder
        self.input_height, self.input_width = input_height, input_width
        self.f = float(f) if f is not None else None
        if is_city:           # semantic_depth.py:592-599
            self.cx, self.cy, self.b = 1048.64 / 4, 519.277 / 4, 0.6
            if self.f is None:
                self.f = 500
        else:                 # :600-607
            self.cx, self.cy, self.b = 314.05519001, 124.09658151, 1
            if self.f is None:
                self.f = 380
        self.engine = engine or Engine(input_height, input_width, 1, encoder)
        assert self.engine.encoder == encoder and (self.engine.H, self.engine.W) == (input_height, input_width)
        self.engine.load_weights(L.SD_NET_MONODEPTH, _load_weight_arg(checkpoint_path))

    def compute_disparity(self, frame: np.ndarray) -> np.ndarray:
        """semantic_depth.py:667-678 -> float32 (H,W), fraction of image width."""
        e = self.engine
        fr = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))[None].to(e.device)
        return e.monodepth_forward(fr)[0].cpu().numpy()

    def compute_3D_points(self, disp: np.ndarray) -> np.ndarray:
        """semantic_depth.py:686-697: cv2.reprojectImageTo3D(disp, Q) -> float32 (H,W,3).  ``disp`` in pixels."""
        e = self.engine
        d = torch.from_numpy(np.ascontiguousarray(disp, dtype=np.float32))[None].to(e.device)
        cam = Camera(self.cx, self.cy, self.f, self.b, 1.0)       # the caller has already applied the multiplier
        out = e.fuse_backproject(d, None, None, None, [cam], dense=True, want_fence=False)
        return out["dense"][0].cpu().numpy()


class FrameProcessor:
    """Compute steps of FrameProcessor.process_frame (semantic_depth.py:98-268) on an already-resized BGR frame.
    ``disp_multiplier``: original_width (semantic_depth.py:109) or 3800 (seq:105)."""

    def __init__(self, frame_segmenter: SegmentFrame, frame_depther: DepthFrame, depth: float = 10.0,
                 disp_multiplier: float | None = None, params: RoadWidthParams | None = None):
        self.frame_segmenter, self.frame_depther = frame_segmenter, frame_depther
        self.depth = depth
        self.disp_multiplier = disp_multiplier
        self.params = params or RoadWidthParams(depth=depth)
        assert frame_segmenter.engine is frame_depther.engine, "share one Engine between the two operators"

    def process_frame(self, frame: np.ndarray, original_width: int | None = None):
        e = self.frame_depther.engine
        d = self.frame_depther
        # semantic_depth.py:105-112: the frame as read from disk is cubic-resized to the network shape and its ORIGINAL width
        # scales the disparities; a frame that is not H x W takes the same route here (on the GPU)
        mult = self.disp_multiplier if self.disp_multiplier is not None else (original_width or frame.shape[1])
        fr = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))[None].to(e.device)
        if tuple(fr.shape[1:3]) != (e.H, e.W):
            fr = e.resize_cubic(fr)
        out = e.process_batch(fr, [Camera(d.cx, d.cy, d.f, d.b, float(mult))], self.params)
        rec = Engine.records(out["records"])[0]
        n = int(out["fuse"]["n_road"][0].item())
        return dict(record=rec, dist_rw=float(rec["width"]) if rec["found"] else None,
                    road_mask=out["seg"]["road"][0].cpu().numpy().astype(bool),
                    fence_mask=out["seg"]["fence"][0].cpu().numpy().astype(bool),
                    disparity=out["disp_pp"][0].cpu().numpy() * np.float32(mult),
                    road3D=out["fuse"]["road_xyz"][0, :n].cpu().numpy(), road_colors=out["fuse"]["road_rgb"][0, :n].cpu().numpy())
