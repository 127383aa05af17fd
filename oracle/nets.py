"""Oracle: FCN-8s and monodepth forward passes with TensorFlow semantics, on torch-CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Tensors are NCHW inside, inputs/outputs are
NHWC like the TF graphs.  ``dtype`` may be torch.float64 to get a higher-precision check value.

Followed sources:
  * FCN-8s decoder            fcn8s/fcn.py:159-215  (score 1x1 x3, deconv 4x4 s2 x2 + adds, deconv 16x16 s8)
  * logits / argmax           fcn8s/fcn.py:241, :218-224
  * softmax + 0.5 thresholds  semantic_depth.py:550-564
  * VGG16-FCN encoder         [UPSTREAM] Udacity vgg SavedModel, tensors named at fcn8s/fcn.py:89-93
  * monodepth                 [UPSTREAM] mrharicot/monodepth monodepth_model.py; call sites
                              semantic_depth.py:609-622 (use_deconv=False, do_stereo=False), :634, :675
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

VGG_MEAN_BGR = (103.939, 116.779, 123.68)


def _t(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype)


def _conv_w(w_hwio, dtype):
    return _t(w_hwio, dtype).permute(3, 2, 0, 1).contiguous()  # -> OIHW


# ------------------------------------------------------------------------------------------
# FCN-8s
# ------------------------------------------------------------------------------------------
def vgg_preprocess(frames_u8: np.ndarray, dtype=torch.float32) -> torch.Tensor:
    """[UPSTREAM] 'Processing' block of the Udacity VGG graph: split the 3 input channels as
    (red, green, blue), subtract the VGG means and re-concatenate as (blue, green, red).
    The graph does this to whatever channel order it is fed (the reference feeds BGR frames,
    semantic_depth.py:105,552)."""
    x = _t(frames_u8, dtype)  # B,H,W,3
    c0, c1, c2 = x[..., 0], x[..., 1], x[..., 2]
    out = torch.stack([c2 - VGG_MEAN_BGR[0], c1 - VGG_MEAN_BGR[1], c0 - VGG_MEAN_BGR[2]], dim=1)
    return out  # B,3,H,W


def fcn8s_forward(frames_u8: np.ndarray, w: dict, dtype=torch.float32, return_taps: bool = False):
    """frames_u8: (B,H,W,3) uint8.  Returns logits (B,H,W,C) as numpy (the 'logits:0' tensor of
    fcn8s/fcn.py:241 before its reshape to (-1, C))."""
    x = vgg_preprocess(frames_u8, dtype)
    taps = {}

    def conv(x, name, pad):
        return F.conv2d(x, _conv_w(w[f"vgg/{name}/filter"], dtype), _t(w[f"vgg/{name}/biases"], dtype), padding=pad)

    blocks = [["conv1_1", "conv1_2"], ["conv2_1", "conv2_2"], ["conv3_1", "conv3_2", "conv3_3"],
              ["conv4_1", "conv4_2", "conv4_3"], ["conv5_1", "conv5_2", "conv5_3"]]
    pools = []
    for blk in blocks:
        for name in blk:
            x = F.relu(conv(x, name, 1))            # 3x3 s1 SAME + bias + ReLU
        x = F.max_pool2d(x, 2, 2)                   # 2x2 s2 SAME (even sizes: no padding)
        pools.append(x)
    l3, l4 = pools[2], pools[3]                     # layer3_out / layer4_out (fcn8s/fcn.py:91-92)
    x = F.relu(conv(pools[4], "fc6", 3))            # 7x7 SAME; dropout(keep_prob=1) is identity
    l7 = F.relu(conv(x, "fc7", 0))                  # layer7_out
    taps.update(layer3=l3, layer4=l4, layer7=l7)

    def score(x, name):                             # fcn8s/fcn.py:165-182
        return F.conv2d(x, _conv_w(w[f"dec/{name}/kernel"], dtype), _t(w[f"dec/{name}/bias"], dtype))

    def deconv(x, name, k, s):                      # fcn8s/fcn.py:186-213; SAME => torch padding (k-s)/2
        wt = _t(w[f"dec/{name}/kernel"], dtype).permute(3, 2, 0, 1).contiguous()  # HWOI -> I,O,H,W
        return F.conv_transpose2d(x, wt, _t(w[f"dec/{name}/bias"], dtype), stride=s, padding=(k - s) // 2)

    s7, s4, s3 = score(l7, "score7"), score(l4, "score4"), score(l3, "score3")
    first_skip = deconv(s7, "deconv1", 4, 2) + s4
    second_skip = deconv(first_skip, "deconv2", 4, 2) + s3
    last = deconv(second_skip, "deconv3", 16, 8)
    logits = last.permute(0, 2, 3, 1).contiguous().numpy()
    if return_taps:
        taps.update(score7=s7, first_skip=first_skip, second_skip=second_skip)
        return logits, {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in taps.items()}
    return logits


def softmax_masks(logits: np.ndarray):
    """semantic_depth.py:550-564 + fcn8s/fcn.py:218-224.  logits (...,3) f32.
    Returns (softmax f32, road bool, fence bool, argmax int64)."""
    lg = torch.as_tensor(logits, dtype=torch.float32)
    sm = torch.softmax(lg, dim=-1)
    road = (sm[..., 0] > 0.5).numpy()
    fence = (sm[..., 1] > 0.5).numpy()
    am = torch.argmax(sm, dim=-1).numpy()
    return sm.numpy(), road, fence, am


# ------------------------------------------------------------------------------------------
# monodepth  [UPSTREAM]
# ------------------------------------------------------------------------------------------
class _Mono:
    def __init__(self, w, dtype):
        self.w, self.dtype = w, dtype

    def conv(self, x, name, k, stride, act="elu"):
        """upstream conv(): zero-pad p=(k-1)//2 on H,W then VALID slim.conv2d with bias (+ELU)."""
        p = (k - 1) // 2
        x = F.pad(x, (p, p, p, p))
        y = F.conv2d(x, _conv_w(self.w[name + "/weights"], self.dtype), _t(self.w[name + "/biases"], self.dtype),
                     stride=stride)
        if act == "elu":
            y = F.elu(y)
        elif act == "sigmoid":
            y = torch.sigmoid(y)
        return y

    def conv_block(self, x, name, k):
        return self.conv(self.conv(x, name + "a", k, 1), name + "b", k, 2)

    def maxpool3(self, x):
        """upstream maxpool(): ZERO-pad 1 then 3x3 stride-2 VALID max (zeros take part in the max)."""
        return F.max_pool2d(F.pad(x, (1, 1, 1, 1)), 3, 2)

    def resconv(self, x, p, n, stride):
        c1 = self.conv(x, p + "/conv1", 1, 1)
        c2 = self.conv(c1, p + "/conv2", 3, stride)
        c3 = self.conv(c2, p + "/conv3", 1, 1, act=None)
        sc = self.conv(x, p + "/proj", 1, stride, act=None)   # do_proj is always true upstream
        return F.elu(c3 + sc)

    def resblock(self, x, stage, n, blocks):
        for b in range(1, blocks):
            x = self.resconv(x, f"enc/res{stage}_{b}", n, 1)
        return self.resconv(x, f"enc/res{stage}_{blocks}", n, 2)  # the LAST block strides

    @staticmethod
    def up(x):
        return F.interpolate(x, scale_factor=2, mode="nearest")   # resize_nearest_neighbor, out[y,x]=in[y//2,x//2]

    def upconv(self, x, name):
        return self.conv(self.up(x), name, 3, 1)

    def get_disp(self, x, name):
        return 0.3 * self.conv(x, name, 3, 1, act="sigmoid")

    def decoder(self, x, skips, top):
        disp_prev = None
        disps = {}
        for lvl in range(top, 0, -1):
            u = self.upconv(x, f"dec/upconv{lvl}")
            cat = [u]
            if lvl in skips:
                cat.append(skips[lvl])
            if lvl <= 3:
                cat.append(self.up(disp_prev))
            x = self.conv(torch.cat(cat, 1), f"dec/iconv{lvl}", 3, 1)
            if lvl <= 4:
                disp_prev = self.get_disp(x, f"dec/disp{lvl}")
                disps[lvl] = disp_prev
        return disps


def monodepth_forward(images_f32: np.ndarray, w: dict, encoder: str = "resnet50", dtype=torch.float32,
                      all_scales: bool = False):
    """images_f32: (N,H,W,3) float in [0,1].  Returns disp1 (N,H,W,2) (left,right) as numpy; the
    reference fetches disp_left_est[0] = disp1[...,0:1] (semantic_depth.py:675)."""
    m = _Mono(w, dtype)
    x = _t(images_f32, dtype).permute(0, 3, 1, 2).contiguous()
    if encoder == "vgg":
        feats = []
        for i, k in enumerate([7, 5, 3, 3, 3, 3, 3], start=1):
            x = m.conv_block(x, f"enc/conv{i}", k)
            feats.append(x)
        skips = {lvl: feats[lvl - 2] for lvl in range(2, 8)}   # skip_l -> decoder level l+1
        disps = m.decoder(feats[6], skips, top=7)
    elif encoder == "resnet50":
        conv1 = m.conv(x, "enc/conv1", 7, 2)
        pool1 = m.maxpool3(conv1)
        conv2 = m.resblock(pool1, 2, 64, 3)
        conv3 = m.resblock(conv2, 3, 128, 4)
        conv4 = m.resblock(conv3, 4, 256, 6)
        conv5 = m.resblock(conv4, 5, 512, 3)
        skips = {6: conv4, 5: conv3, 4: conv2, 3: pool1, 2: conv1}
        disps = m.decoder(conv5, skips, top=6)
    else:
        raise ValueError(encoder)
    out = {k: v.permute(0, 2, 3, 1).contiguous().numpy() for k, v in disps.items()}
    return out if all_scales else out[1]


def compute_disparity(frame_u8: np.ndarray, w: dict, encoder: str = "resnet50", dtype=torch.float32):
    """DepthFrame.compute_disparity, semantic_depth.py:667-678: /255, stack with fliplr, run the
    net on the pair, keep channel 0, post-process, cast to f32."""
    from .fusion import post_processing
    f = frame_u8.astype(np.float32) / 255
    pair = np.stack((f, np.fliplr(f)), 0)
    disp = monodepth_forward(pair, w, encoder, dtype)[..., 0].astype(np.float32)
    return post_processing(disp).astype(np.float32)
