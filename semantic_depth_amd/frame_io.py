"""Input stage, host side: the frame reader that replaces ``cv2.imread`` (semantic_depth.py:105; seq:123) and a
feeder that keeps the GPU supplied.

    imread(path)            8-bit PNG -> u8 [h,w,3] BGR, exactly cv2.imread's IMREAD_COLOR result (PNG is lossless); zlib inflates
                            (GIL released), libsemdepth's sd_png_unfilter_bgr reconstructs the scanlines and shuffles to BGR
    FrameFeeder             thread pool decoding the sorted file list (seq:689) into pinned staging buffers, batch by batch,
                            one batch ahead of the GPU; the cubic resize to the network shape happens ON the GPU
                            (Engine.resize_cubic), so the host never touches a pixel after the decode

Decode stays on the host on purpose: DEFLATE and the PNG predictors are serial byte recurrences; a frame costs a few
milliseconds of one core and the GPU box has hundreds (scripts/feed_rate.py measures decode, pinned H2D and resize rates
against the benchmarked frames/s; DESIGN.md quotes them).  JPEG is not decoded here (the reference's inputs are PNG).
"""
from __future__ import annotations

import ctypes as C
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib as L

_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def decode_png(buf: bytes) -> np.ndarray:
    """PNG bytes -> u8 [h,w,3] BGR (cv2.IMREAD_COLOR semantics: 3 channels, alpha dropped, gray replicated, palette expanded)"""
    if buf[:8] != _SIG:
        raise ValueError("not a PNG file")
    p, idat, hdr, plte = 8, [], None, None
    while p < len(buf):
        n, tag = struct.unpack_from(">I4s", buf, p)
        body = buf[p + 8:p + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        p += 12 + n
    if hdr is None:
        raise ValueError("PNG without IHDR")
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in _CHANNELS:
        raise ValueError(f"unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace}): 8-bit non-interlaced only")
    ch = _CHANNELS[ctype]
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != h * (1 + w * ch):
        raise ValueError("PNG: inflated size does not match the header")
    out = np.empty((h, w, 3), np.uint8)
    lib = L.load()
    st = lib.sd_png_unfilter_bgr(C.c_char_p(raw), h, w, ch, out.ctypes.data_as(C.c_void_p))
    if st != L.SD_OK:
        raise ValueError("PNG: bad filter type")
    if ctype == 3:                      # palette indices (replicated into 3 channels by the helper) -> BGR palette entries
        if plte is None:
            raise ValueError("PNG: palette image without PLTE")
        out = plte[out[..., 0]][..., ::-1].copy()
    return out


def imread(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        return decode_png(f.read())


class FrameFeeder:
    """iterate over (frames u8 [n,h,w,3] on ``device``, first global index) for the sorted ``paths``: decode on ``workers`` host
    threads into pinned buffers, upload asynchronously, one batch ahead of the consumer (double buffering)."""

    def __init__(self, paths, batch: int, device="cuda", workers: int = 16):
        import torch
        self.paths, self.batch, self.device = list(paths), batch, torch.device(device)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self._torch = torch
        self._pinned = [None, None]

    def close(self):
        self.pool.shutdown(wait=False)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _decode_into(self, slot: int, lo: int, hi: int):
        torch = self._torch
        first = imread(self.paths[lo])
        h, w = first.shape[:2]
        buf = self._pinned[slot]
        if buf is None or tuple(buf.shape[1:3]) != (h, w) or buf.shape[0] < hi - lo:
            buf = self._pinned[slot] = torch.empty((self.batch, h, w, 3), dtype=torch.uint8, pin_memory=self.device.type == "cuda")
        view = buf.numpy()
        view[0] = first

        def one(i):
            view[i - lo] = imread(self.paths[i])

        list(self.pool.map(one, range(lo + 1, hi)))
        return buf[:hi - lo]

    def __iter__(self):
        torch = self._torch
        n = len(self.paths)
        ranges = [(a, min(a + self.batch, n)) for a in range(0, n, self.batch)]
        if not ranges:
            return
        with ThreadPoolExecutor(max_workers=1) as ahead:
            fut = ahead.submit(self._decode_into, 0, *ranges[0])
            for k, (lo, hi) in enumerate(ranges):
                host = fut.result()
                if k + 1 < len(ranges):
                    fut = ahead.submit(self._decode_into, (k + 1) & 1, *ranges[k + 1])
                dev = host.to(self.device, non_blocking=True)
                if self.device.type == "cuda":
                    torch.cuda.current_stream().synchronize()      # the pinned buffer is reused two batches later
                yield dev, lo
