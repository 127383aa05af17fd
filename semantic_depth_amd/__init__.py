"""semantic_depth_amd — MI355X-native implementation of semantic-depth's per-frame hot path.

FCN-8s forward + monodepth forward + per-pixel fusion/back-projection -> class-masked point cloud -> road width
(BASELINE.json north_star).  All compute is hand-written HIP for gfx950 in libsemdepth.so
(include/semdepth.h); this package is the Python host side mirroring the reference's operator interface:

    SegmentFrame, DepthFrame, FrameProcessor     (semantic_depth.py:82-96, :464-571, :575-697)
    pcl, point_cloud_2_ply                        (semantic_depth_lib/pcl.py, point_cloud_2_ply.py)
    outputs, frame_io, distributed, tf_import     (file outputs + focal sweep, PNG reader / feeder, the multi-GPU sequence driver,
                                                   TensorFlow-free checkpoint import)

There is no CPU fallback: without the built library or without a GPU the operators raise.
"""
from .weights import (NUM_CLASSES, fcn8s_weight_shapes, make_fcn8s_weights, make_monodepth_weights,  # noqa: F401
                      monodepth_weight_shapes)


def __getattr__(name):  # lazy: importing the package must not need torch/GPU (weights tables are pure numpy)
    if name in ("Engine", "RoadWidthParams", "FenceParams", "Camera", "RW_DTYPE", "F2F_DTYPE"):
        from . import engine
        return getattr(engine, name)
    if name in ("SegmentFrame", "DepthFrame", "FrameProcessor"):
        from . import api
        return getattr(api, name)
    if name in ("pcl", "outputs", "frame_io", "distributed", "tf_import", "point_cloud_2_ply", "api", "engine"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
