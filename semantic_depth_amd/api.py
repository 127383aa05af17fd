"""Host-side mirror of the reference's operator interface for the hot path.

    SegmentFrame  <- semantic_depth.py:464-571   (seq:378-485)
    DepthFrame    <- semantic_depth.py:575-697   (seq:488-589)
    FrameProcessor.process_frame <- semantic_depth.py:98-334 (seq:117-298), compute steps

Same constructor arguments / method names / return types, so the reference's FrameProcessor body reads the same
against these classes, and they are built the way the reference's main() builds them (DepthFrame and SegmentFrame
independently, semantic_depth.py:773-789): both resolve to ONE shared Engine per (H, W, device, precision) through the
registry below.  Differences: weights come from a dict / .npz of TF-layout arrays or a TF checkpoint / frozen graph read
by ``tf_import`` (the reference restores TF checkpoints; INTEGRATION.md has the name map), and every method also has a
batched, device-resident form on the Engine.  File outputs live in ``outputs.py``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from .engine import Camera, Engine, FenceParams, RoadWidthParams

# ---------------------------------------------------------------------------------------------------------------------
# one Engine per (H, W, device, precision): the reference builds DepthFrame and SegmentFrame independently and both then
# live in one process / on one GPU; here they share the handle, the activation workspace and the stream
# ---------------------------------------------------------------------------------------------------------------------
_engines: dict = {}
# The reference computes in float32 (semantic_depth.py:550-552, 675).  Three engines here do, to the same measured error against a float64
# oracle (profiles/r06_f32_grade_check.txt): "f16x2" (every f32 operand carried to 22 bits as fp16 hi + scaled lo planes, three fp16 MFMA products, f32
# accumulation: the engine bench.py headlines; a value beyond the fp16 range of its planes raises engine.RangeError), "bf16x3" (three exact bf16 planes,
# six products, no range limit) and "f32" (the f32 MFMA, a third of the speed).  The classes below default to the fastest; precision= selects the others.
DEFAULT_PRECISION = "f16x2"


def shared_engine(H: int, W: int, device: int = 0, precision: str = DEFAULT_PRECISION, encoder: str | None = None, max_batch: int = 1) -> Engine:
    """the registered Engine for this geometry; created on first use.  ``encoder`` None = whatever is registered (or 'vgg',
    the reference's default --monodepth_encoder, semantic_depth.py:721-722)."""
    key = (int(H), int(W), int(device), precision)
    per = _engines.setdefault(key, {})
    if encoder is None:
        if per:
            return next(iter(per.values()))
        encoder = "vgg"
    eng = per.get(encoder)
    if eng is None or eng.max_batch < max_batch:
        eng = Engine(H, W, max_batch, encoder, device, precision=precision)
        eng._api_loaded = {}
        per[encoder] = eng
    return eng


def register_engine(engine: Engine):
    """make an existing Engine the shared one for its geometry (bench / batched drivers build theirs with max_batch > 1)."""
    key = (engine.H, engine.W, engine.device.index or 0, engine.precision)
    if not hasattr(engine, "_api_loaded"):
        engine._api_loaded = {}
    _engines.setdefault(key, {})[engine.encoder] = engine


def release_engines():
    """drop the shared Engines (their arenas are freed once the operator objects that hold them are gone too)"""
    for per in _engines.values():
        for e in per.values():
            e.close()
    _engines.clear()


def any_engine() -> Engine | None:
    for per in _engines.values():
        for e in per.values():
            return e
    return None


def _load_weight_arg(w, net="fcn8s", encoder=None):
    if isinstance(w, dict):
        return w
    if isinstance(w, str):
        if w.endswith(".npz"):
            z = np.load(w)
            return {k: z[k] for k in z.files}
        from . import tf_import           # TF checkpoint prefix / frozen .pb (semantic_depth.py:498-541, :627-653)
        return tf_import.load_any(w, net, encoder)
    raise TypeError("weights must be a dict name->array, a path to an .npz, a TF checkpoint prefix or a frozen .pb")


def _ensure_loaded(engine: Engine, net: int, weights: dict):
    loaded = getattr(engine, "_api_loaded", None)
    if loaded is None:
        loaded = engine._api_loaded = {}
    if loaded.get(net) is not weights:
        engine.load_weights(net, weights)
        loaded[net] = weights


class SegmentFrame:
    """SegmentFrame(input_shape, model_var_dir, use_frozen, use_xla, CUDA_DEVICE_NUMBER) — semantic_depth.py:464-469.
    ``model_var_dir``: dict / .npz of FCN-8s weights (weights.fcn8s_weight_shapes).  use_frozen / use_xla are accepted
    and ignored (TF graph details)."""

    def __init__(self, input_shape, model_var_dir, use_frozen=True, use_xla=False, CUDA_DEVICE_NUMBER="0", engine: Engine | None = None,
                 precision: str = DEFAULT_PRECISION):
        self.input_shape = tuple(input_shape)
        self.model_var_dir = model_var_dir
        self.CUDA_DEVICE_NUMBER = CUDA_DEVICE_NUMBER
        self.precision = precision
        self._weights = _load_weight_arg(model_var_dir)
        self._engine = engine

    @property
    def engine(self) -> Engine:
        """resolved on first use, so that a DepthFrame built before OR after this object decides the monodepth encoder of the
        shared Engine"""
        if self._engine is None:
            self._engine = shared_engine(self.input_shape[0], self.input_shape[1], int(self.CUDA_DEVICE_NUMBER), self.precision)
        _ensure_loaded(self._engine, L.SD_NET_FCN8S, self._weights)
        return self._engine

    def segment_frame(self, frame: np.ndarray):
        """semantic_depth.py:544-571: (road bool (H,W,1), fence bool (H,W,1), overlay u8 (H,W,3))."""
        e = self.engine
        fr = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))[None].to(e.device)
        out = e.fcn8s_forward(fr)
        road = out["road"][0].cpu().numpy().astype(bool)[..., None]
        fence = out["fence"][0].cpu().numpy().astype(bool)[..., None]
        e.check_range()                       # (engines with fp16 planes: a value beyond their range is an error of this call)
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
                 engine: Engine | None = None, precision: str = DEFAULT_PRECISION, device: int = 0):
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
        self._engine = engine or shared_engine(input_height, input_width, device, precision, encoder)
        if self._engine.encoder != encoder or (self._engine.H, self._engine.W) != (input_height, input_width):
            raise ValueError(f"engine is {self._engine.encoder} {self._engine.H}x{self._engine.W}, DepthFrame wants {encoder} "
                             f"{input_height}x{input_width}")
        self._weights = _load_weight_arg(checkpoint_path, "monodepth", encoder)
        _ensure_loaded(self._engine, L.SD_NET_MONODEPTH, self._weights)

    @property
    def engine(self) -> Engine:
        """the shared Engine with THIS object's checkpoint in its monodepth arena: two DepthFrames of one geometry (city / kitti
        checkpoints) share the handle, so every access re-checks which weights are loaded -- as SegmentFrame.engine does"""
        _ensure_loaded(self._engine, L.SD_NET_MONODEPTH, self._weights)
        return self._engine

    def camera(self, disp_mult: float) -> Camera:
        return Camera(self.cx, self.cy, self.f, self.b, float(disp_mult))

    def compute_disparity(self, frame: np.ndarray) -> np.ndarray:
        """semantic_depth.py:667-678 -> float32 (H,W), fraction of image width."""
        e = self.engine
        fr = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))[None].to(e.device)
        disp = e.monodepth_forward(fr)[0].cpu().numpy()
        e.check_range()
        return disp

    def disp_to_image(self, disp_pp: np.ndarray, output_name: str, original_height: int, original_width: int) -> str:
        """semantic_depth.py:681-683: ``scipy.misc.imresize(disp_pp.squeeze(), [h, w])`` + ``plt.imsave(name_disp.png, cmap='gray')``
        (a PNG side effect only; not on the compute path).  imresize = min-max scaling to uint8 (bytescale: ``(v - min) * 255 / (max -
        min)`` truncated after + 0.5) followed by PIL's bilinear resize; imsave maps the uint8 image linearly onto the gray colormap,
        i.e. min -> 0, max -> 255 again.  Written as an 8-bit gray PNG ``<output_name>_disp.png``; returns its path.  PIL's resampling
        arithmetic (bilinear with antialiasing support when shrinking) is restated as plain bilinear interpolation at pixel centres --
        the reference only ever enlarges the map here (network size -> original frame size)."""
        from .outputs import write_png
        d = np.asarray(disp_pp, dtype=np.float64).squeeze()
        lo, hi = float(d.min()), float(d.max())
        scale = 255.0 / (hi - lo) if hi > lo else 1.0
        u8 = np.clip((d - lo) * scale + 0.5, 0, 255).astype(np.uint8).astype(np.float64)
        h, w = u8.shape
        ys = np.clip((np.arange(original_height) + 0.5) * h / original_height - 0.5, 0, h - 1)
        xs = np.clip((np.arange(original_width) + 0.5) * w / original_width - 0.5, 0, w - 1)
        y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
        y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        img = (u8[y0][:, x0] * (1 - fy) * (1 - fx) + u8[y0][:, x1] * (1 - fy) * fx + u8[y1][:, x0] * fy * (1 - fx) + u8[y1][:, x1] * fy * fx)
        img = np.clip(img + 0.5, 0, 255).astype(np.uint8)
        lo8, hi8 = int(img.min()), int(img.max())                  # plt.imsave normalises the array onto the colormap
        if hi8 > lo8:
            img = np.clip((img.astype(np.float64) - lo8) * (255.0 / (hi8 - lo8)) + 0.5, 0, 255).astype(np.uint8)
        return write_png("{}_disp.png".format(output_name), img)

    def compute_3D_points(self, disp: np.ndarray) -> np.ndarray:
        """semantic_depth.py:686-697: cv2.reprojectImageTo3D(disp, Q) -> float32 (H,W,3).  ``disp`` in pixels."""
        e = self.engine
        d = torch.from_numpy(np.ascontiguousarray(disp, dtype=np.float32))[None].to(e.device)
        cam = Camera(self.cx, self.cy, self.f, self.b, 1.0)       # the caller has already applied the multiplier
        out = e.fuse_backproject(d, None, None, None, [cam], dense=True, want_fence=False)
        return out["dense"][0].cpu().numpy()


class FrameProcessor:
    """Compute steps of FrameProcessor.process_frame (semantic_depth.py:98-334) on a BGR frame (any size: it is
    cubic-resized to the network shape on the GPU like :111).
    ``approach``: 'rw' or 'both' (:743-745; 'both' adds the fence chain and the fence-to-fence distance, :273-334).
    ``disp_multiplier``: None = the original width of each frame (semantic_depth.py:109), or a constant (3800, seq:105)."""

    def __init__(self, frame_segmenter: SegmentFrame, frame_depther: DepthFrame, depth: float = 10.0,
                 disp_multiplier: float | None = None, params: RoadWidthParams | None = None, approach: str = "rw",
                 fence_params: FenceParams | None = None):
        self.frame_segmenter, self.frame_depther = frame_segmenter, frame_depther
        self.depth = depth
        self.approach = approach
        self.disp_multiplier = disp_multiplier
        self.params = params or RoadWidthParams(depth=depth)
        self.fence_params = fence_params or FenceParams(depth=depth)
        if frame_segmenter._engine is None:          # built independently of the DepthFrame: join its Engine
            d = frame_depther.engine
            if (d.H, d.W) == tuple(frame_segmenter.input_shape):
                frame_segmenter._engine = d

    def process_frame(self, frame: np.ndarray, original_width: int | None = None, want_clouds: bool = False):
        es, e = self.frame_segmenter.engine, self.frame_depther.engine
        d = self.frame_depther
        # semantic_depth.py:105-112: the frame as read from disk is cubic-resized to the network shape and its ORIGINAL width
        # scales the disparities; a frame that is not H x W takes the same route here (on the GPU)
        mult = self.disp_multiplier if self.disp_multiplier is not None else (original_width or frame.shape[1])
        fr = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))[None].to(e.device)
        if tuple(fr.shape[1:3]) != (e.H, e.W):
            fr = e.resize_cubic(fr)
        cams = [d.camera(mult)]
        if es is e:
            out = e.process_batch(fr, cams, self.params, approach=self.approach, fence_params=self.fence_params, want_final=want_clouds)
        else:
            # two Engines (different geometry registries): FCN-8s on the segmenter's, everything else on the depther's
            seg = es.fcn8s_forward(fr)
            disp_pp = e.monodepth_forward(fr)
            fz = e.fuse_backproject(disp_pp, seg["road"], seg["fence"], fr, cams)
            rw = e.road_width(fz["road_xyz"], fz["n_road"], self.params, want_final=want_clouds, road_rgb=fz["road_rgb"])
            out = dict(seg=seg, disp_pp=disp_pp, fuse=fz, records=rw[0] if want_clouds else rw, f2f=None)
            if want_clouds:
                out["road_final"] = dict(xyz=rw[1], rgb=rw[2], n=rw[3])
            if self.approach == "both":
                f2 = e.fence_to_fence(fz["fence_xyz"], fz["n_fence"], out["records"], self.fence_params, fence_rgb=fz["fence_rgb"],
                                      want_clouds=want_clouds)
                out["f2f"], out["fence_final"] = f2 if want_clouds else (f2, None)
        rec = Engine.records(out["records"])[0]
        e.check_range()
        if es is not e:
            es.check_range()
        n = int(out["fuse"]["n_road"][0].item())
        res = dict(record=rec, dist_rw=float(rec["width"]) if rec["found"] else None, dist_f2f=None, f2f_record=None,
                   road_mask=out["seg"]["road"][0].cpu().numpy().astype(bool),
                   fence_mask=out["seg"]["fence"][0].cpu().numpy().astype(bool),
                   disparity=out["disp_pp"][0].cpu().numpy() * np.float32(mult),
                   road3D=out["fuse"]["road_xyz"][0, :n].cpu().numpy(), road_colors=out["fuse"]["road_rgb"][0, :n].cpu().numpy())
        if out.get("f2f") is not None:
            f2 = Engine.f2f_records(out["f2f"])[0]
            res["f2f_record"] = f2
            res["dist_f2f"] = float(f2["dist"]) if f2["ok"] else None      # semantic_depth.py:327
        if want_clouds:
            n_f = int(out["fuse"]["n_fence"][0].item()) if out["fuse"].get("n_fence") is not None else 0
            if n_f and out["fuse"].get("fence_xyz") is not None:   # the gathered fence cloud (semantic_depth.py:186-187): the save path
                res["fence3D"] = out["fuse"]["fence_xyz"][0, :n_f].cpu().numpy()      # rebuilds the visualisation planes from it
                res["fence_colors"] = out["fuse"]["fence_rgb"][0, :n_f].cpu().numpy()
            nf = int(out["road_final"]["n"][0].item())
            res["road3D_final"] = out["road_final"]["xyz"][0, :nf].cpu().numpy()
            res["road_colors_final"] = out["road_final"]["rgb"][0, :nf].cpu().numpy()
            if out.get("fence_final"):
                cl, cnt = out["fence_final"], res["f2f_record"]["counts"]
                res["fence3D_left"] = cl["left_xyz"][0, :int(cnt[5])].cpu().numpy()
                res["fence_left_colors"] = cl["left_rgb"][0, :int(cnt[5])].cpu().numpy()
                res["fence3D_right"] = cl["right_xyz"][0, :int(cnt[6])].cpu().numpy()
                res["fence_right_colors"] = cl["right_rgb"][0, :int(cnt[6])].cpu().numpy()
        return res
