"""Weight tables and seeded synthetic weights for the two networks on the hot path.

The reference ships no weights (models/get_sem_seg_models.md, models/get_monodepth_model.sh:14,
.MISSING_LARGE_BLOBS:1-5) so every test and the bench run on seeded synthetic weights.
Arrays are kept in the layouts TensorFlow stores them in, because that is what a real
checkpoint importer would hand over:

  * conv            : HWIO  [kh, kw, cin, cout]            (tf.nn.conv2d / slim.conv2d)
  * transposed conv : HWOI  [kh, kw, cout, cin]            (tf.layers.conv2d_transpose, fcn8s/fcn.py:186-213)
  * bias            : [cout]

Names are logical (``vgg/conv1_1/filter``); INTEGRATION.md maps them to the TF variable names.

The tables here are the Python-side statement of the architectures; the C++ planner
(csrc/netplan.cpp) holds its own statement and tests/test_abi.py checks the two agree.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

NUM_CLASSES = 3  # road=0, fence=1, everything else=2 (fcn8s/helper.py:173-175, fcn8s/fcn.py:626-627)

# ---------------------------------------------------------------------------------------------
# FCN-8s (fcn8s/fcn.py:82-95 tensor names, :159-215 decoder; VGG body is the Udacity SavedModel)
# ---------------------------------------------------------------------------------------------
_VGG_CONVS = [
    ("conv1_1", 3, 64), ("conv1_2", 64, 64),
    ("conv2_1", 64, 128), ("conv2_2", 128, 128),
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256),
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512),
    ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512),
]


def fcn8s_weight_shapes(num_classes: int = NUM_CLASSES) -> "OrderedDict[str, tuple]":
    t: "OrderedDict[str, tuple]" = OrderedDict()
    for name, cin, cout in _VGG_CONVS:
        t[f"vgg/{name}/filter"] = (3, 3, cin, cout)
        t[f"vgg/{name}/biases"] = (cout,)
    t["vgg/fc6/filter"] = (7, 7, 512, 4096)
    t["vgg/fc6/biases"] = (4096,)
    t["vgg/fc7/filter"] = (1, 1, 4096, 4096)
    t["vgg/fc7/biases"] = (4096,)
    for name, cin in (("score7", 4096), ("score4", 512), ("score3", 256)):
        t[f"dec/{name}/kernel"] = (1, 1, cin, num_classes)
        t[f"dec/{name}/bias"] = (num_classes,)
    for name, k in (("deconv1", 4), ("deconv2", 4), ("deconv3", 16)):
        t[f"dec/{name}/kernel"] = (k, k, num_classes, num_classes)  # HWOI
        t[f"dec/{name}/bias"] = (num_classes,)
    return t


# ---------------------------------------------------------------------------------------------
# monodepth (un-vendored upstream; call sites semantic_depth.py:609-622,634)
# ---------------------------------------------------------------------------------------------
def _mono_decoder(t, enc_out, skips, top):
    """skips: dict level -> channels of the skip tensor; top: highest decoder level (7 vgg / 6 r50)."""
    dec_ch = {7: 512, 6: 512, 5: 256, 4: 128, 3: 64, 2: 32, 1: 16}
    cin = enc_out
    for lvl in range(top, 0, -1):
        c = dec_ch[lvl]
        t[f"dec/upconv{lvl}/weights"] = (3, 3, cin, c)
        t[f"dec/upconv{lvl}/biases"] = (c,)
        cat = c + skips.get(lvl, 0) + (2 if lvl <= 3 else 0)  # udisp (2 ch) joins from level 3 down
        t[f"dec/iconv{lvl}/weights"] = (3, 3, cat, c)
        t[f"dec/iconv{lvl}/biases"] = (c,)
        if lvl <= 4:
            t[f"dec/disp{lvl}/weights"] = (3, 3, c, 2)
            t[f"dec/disp{lvl}/biases"] = (2,)
        cin = c


def monodepth_weight_shapes(encoder: str) -> "OrderedDict[str, tuple]":
    t: "OrderedDict[str, tuple]" = OrderedDict()
    if encoder == "vgg":
        spec = [(32, 7), (64, 5), (128, 3), (256, 3), (512, 3), (512, 3), (512, 3)]
        cin = 3
        for i, (c, k) in enumerate(spec, start=1):
            t[f"enc/conv{i}a/weights"] = (k, k, cin, c)
            t[f"enc/conv{i}a/biases"] = (c,)
            t[f"enc/conv{i}b/weights"] = (k, k, c, c)
            t[f"enc/conv{i}b/biases"] = (c,)
            cin = c
        # skip_l joins at decoder level l+1: skip6->7, skip5->6, ... skip1->2
        skips = {7: 512, 6: 512, 5: 256, 4: 128, 3: 64, 2: 32}
        _mono_decoder(t, 512, skips, top=7)
    elif encoder == "resnet50":
        t["enc/conv1/weights"] = (7, 7, 3, 64)
        t["enc/conv1/biases"] = (64,)
        cin = 64
        for stage, (n, blocks) in enumerate([(64, 3), (128, 4), (256, 6), (512, 3)], start=2):
            for b in range(1, blocks + 1):
                p = f"enc/res{stage}_{b}"
                t[f"{p}/conv1/weights"] = (1, 1, cin, n)
                t[f"{p}/conv1/biases"] = (n,)
                t[f"{p}/conv2/weights"] = (3, 3, n, n)
                t[f"{p}/conv2/biases"] = (n,)
                t[f"{p}/conv3/weights"] = (1, 1, n, 4 * n)
                t[f"{p}/conv3/biases"] = (4 * n,)
                # upstream's do_proj test is always true -> every block projects its shortcut
                t[f"{p}/proj/weights"] = (1, 1, cin, 4 * n)
                t[f"{p}/proj/biases"] = (4 * n,)
                cin = 4 * n
        skips = {6: 1024, 5: 512, 4: 256, 3: 64, 2: 64}  # conv4, conv3, conv2, pool1, conv1
        _mono_decoder(t, 2048, skips, top=6)
    else:
        raise ValueError(f"unknown monodepth encoder {encoder!r}")
    return t


# ---------------------------------------------------------------------------------------------
# seeded synthetic weights
# ---------------------------------------------------------------------------------------------
def _trunc_normal(rng, shape, std):
    """tf.truncated_normal_initializer semantics: resample outside 2 sigma."""
    x = rng.standard_normal(shape, dtype=np.float32)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()), dtype=np.float32)
        bad = np.abs(x) > 2.0
    return x * np.float32(std)


def make_fcn8s_weights(seed: int = 1, num_classes: int = NUM_CLASSES, decoder_std: float = 0.01,
                       score_gain=(1.0, 1.0, 1.0), bias_std: float = 0.0):
    """He-normal for the ReLU convs, truncated_normal(decoder_std) for the six decoder layers
    (fcn8s/fcn.py:161 uses stddev 0.01; tests/bench pass a larger value so that masks are non-trivial)."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, shape in fcn8s_weight_shapes(num_classes).items():
        if len(shape) == 1:
            w[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(bias_std)
                       if bias_std else np.zeros(shape, np.float32))
        elif name.startswith("vgg/"):
            fan_in = shape[0] * shape[1] * shape[2]
            w[name] = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        else:
            g = 1.0
            for i, s in enumerate(("score7", "score4", "score3")):
                if s in name:
                    g = score_gain[i]
            w[name] = _trunc_normal(rng, shape, decoder_std * g)
    return w


def make_monodepth_weights(encoder: str = "resnet50", seed: int = 2, gain: float = 1.0,
                           bias_std: float = 0.0):
    """Xavier-uniform (slim.conv2d default initializer) times ``gain``; zero biases by default."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, shape in monodepth_weight_shapes(encoder).items():
        if len(shape) == 1:
            w[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(bias_std)
                       if bias_std else np.zeros(shape, np.float32))
        else:
            kh, kw, cin, cout = shape
            lim = gain * np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
            w[name] = (rng.random(shape, dtype=np.float32) * np.float32(2 * lim) - np.float32(lim))
    return w


def count_params(shapes) -> int:
    return int(sum(int(np.prod(s)) for s in shapes.values()))
