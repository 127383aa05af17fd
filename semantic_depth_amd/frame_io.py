"""Input stage, host side: the frame reader that replaces ``cv2.imread`` (semantic_depth.py:105; seq:123) and a
feeder that keeps the GPU supplied.

    imread(path)            8-bit PNG -> u8 [h,w,3] BGR, exactly cv2.imread's IMREAD_COLOR result (PNG is lossless): ONE native call
                            (sd_png_decode_bgr: chunk walk, zlib inflate, scanline reconstruction, BGR shuffle, palette)
    FrameFeeder             the sorted file list (seq:689) batch by batch: ONE native call per batch (sd_decode_files_bgr, C++ threads)
                            reads and decodes straight into a pinned staging buffer, one batch ahead of the GPU; the cubic resize to
                            the network shape happens ON the GPU (Engine.resize_cubic), so the host never touches a pixel after the decode

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


def png_size(buf: bytes) -> tuple[int, int]:
    """(height, width) of a PNG this reader takes (8-bit, non-interlaced); ValueError otherwise"""
    lib = L.load()
    h, w = C.c_int(), C.c_int()
    if lib.sd_png_decode_bgr(buf, len(buf), None, 0, C.byref(h), C.byref(w)) != L.SD_OK:
        if bytes(buf[:8]) != _SIG:
            raise ValueError("not a PNG file")
        raise ValueError("unsupported or corrupt PNG (8-bit, non-interlaced gray / gray+alpha / RGB / RGBA / palette only)")
    return h.value, w.value


def decode_png(buf: bytes) -> np.ndarray:
    """PNG bytes -> u8 [h,w,3] BGR (cv2.IMREAD_COLOR semantics: 3 channels, alpha dropped, gray replicated, palette expanded).
    One native call (sd_png_decode_bgr: chunk walk, zlib inflate, scanline reconstruction, shuffle), the interpreter lock released."""
    h, w = png_size(buf)
    out = np.empty((h, w, 3), np.uint8)
    lib = L.load()
    if lib.sd_png_decode_bgr(buf, len(buf), out.ctypes.data_as(C.c_void_p), out.nbytes, None, None) != L.SD_OK:
        raise ValueError("PNG: corrupt stream (inflate / filter type / palette index)")
    return out


def image_size(buf: bytes) -> tuple[int, int]:
    """(height, width) of a PNG or Huffman JPEG (SOF0 / SOF1 / SOF2) as cv2.imread would return it (JPEG: after the EXIF orientation)"""
    lib = L.load()
    h, w = C.c_int(), C.c_int()
    if lib.sd_image_decode_bgr(buf, len(buf), None, 0, C.byref(h), C.byref(w)) != L.SD_OK:
        raise ValueError("not a PNG / JPEG this reader takes (PNG: 8-bit non-interlaced; JPEG: baseline, extended-sequential or progressive Huffman, 8-bit, "
                         "gray or YCbCr 4:4:4 / 4:2:2 / 4:2:0)")
    return h.value, w.value


def decode_jpeg(buf: bytes) -> np.ndarray:
    """JPEG bytes (baseline / extended-sequential / progressive Huffman) -> u8 [h,w,3] BGR = cv2.imread: libjpeg's default decode path (ISLOW inverse DCT, fancy chroma upsampling,
    fixed-point YCbCr -> RGB) and the EXIF orientation, restated natively (sd_jpeg_decode_bgr)"""
    if bytes(buf[:2]) != b"\xff\xd8":
        raise ValueError("not a JPEG file")
    return decode_image(buf)


def decode_image(buf: bytes) -> np.ndarray:
    h, w = image_size(buf)
    out = np.empty((h, w, 3), np.uint8)
    if L.load().sd_image_decode_bgr(buf, len(buf), out.ctypes.data_as(C.c_void_p), out.nbytes, None, None) != L.SD_OK:
        raise ValueError("corrupt image stream")
    return out


def imread(path: str) -> np.ndarray:
    """cv2.imread(path) for 8-bit PNG and Huffman JPEG (SOF0 / SOF1 / SOF2) files (semantic_depth.py:105; seq:123)"""
    with open(path, "rb") as f:
        buf = f.read()
    return decode_png(buf) if buf[:8] == _SIG else decode_image(buf)


def _cgroup_cpu_quota():
    """CPUs the cgroup quota allows (v2: cpu.max "<quota> <period>" | "max <period>"; v1: cpu.cfs_quota_us / cpu.cfs_period_us), or None"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return None if q <= 0 or per <= 0 else max(1, q // per)
    except (OSError, ValueError):
        return None


def _local_ranks() -> int:
    """ranks that share this node: LOCAL_WORLD_SIZE (torch.distributed.run, bench.py's launcher), Open MPI's / MPICH's / Slurm's per-node counts
    (SLURM_TASKS_PER_NODE is always set by srun, SLURM_NTASKS_PER_NODE only with --ntasks-per-node), else WORLD_SIZE -- the conservative single-node
    assumption for a job started by hand with RANK / WORLD_SIZE: on several nodes it UNDER-subscribes the decode threads (affinity / WORLD_SIZE each),
    never over-subscribes them --, else 1"""
    import os
    for k in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "SLURM_NTASKS_PER_NODE", "SLURM_TASKS_PER_NODE", "WORLD_SIZE"):
        v = os.environ.get(k, "")
        try:
            n = int(v.split("(")[0])           # (Slurm writes "8(x2)" for heterogeneous jobs)
        except ValueError:
            continue
        if n >= 1:
            return n
    return 1


def default_decode_workers() -> int:
    """decode threads of ONE rank: the CPUs this process may run on (its affinity mask, capped by the cgroup CPU quota -- os.cpu_count()
    reports the host's 256 whatever the container may use) divided by the ranks of the node (_local_ranks), at least 1, at most 128.
    Eight ranks on a 256-CPU node get 32 threads each instead of 8 x 128."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 8
    q = _cgroup_cpu_quota()
    if q is not None:
        n = min(n, q)
    return max(1, min(n // _local_ranks(), 128))


class FrameFeeder:
    """iterate over (frames u8 [n,h,w,3] on ``device``, first global index) for the sorted ``paths``: every batch is read and decoded
    by ONE native call (sd_decode_files_bgr: ``workers`` C++ threads, no interpreter lock) straight into a pinned staging buffer,
    uploaded asynchronously, one batch ahead of the consumer (two staging buffers)."""

    def __init__(self, paths, batch: int, device="cuda", workers: int = 0):
        import os
        import torch
        self.paths, self.batch, self.device = list(paths), batch, torch.device(device)
        self.workers = workers if workers > 0 else default_decode_workers()
        self._torch = torch
        self._pinned = [None, None]
        self._lib = L.load()

    def close(self):
        self._pinned = [None, None]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _decode_into(self, slot: int, lo: int, hi: int):
        torch = self._torch
        with open(self.paths[lo], "rb") as f:
            h, w = image_size(f.read())
        buf = self._pinned[slot]
        if buf is None or tuple(buf.shape[1:3]) != (h, w) or buf.shape[0] < hi - lo:
            buf = self._pinned[slot] = torch.empty((self.batch, h, w, 3), dtype=torch.uint8, pin_memory=self.device.type == "cuda")
        n = hi - lo
        arr = (C.c_char_p * n)(*[os_fsencode(p) for p in self.paths[lo:hi]])
        status = (C.c_int * n)()
        st = self._lib.sd_decode_files_bgr(arr, n, h, w, C.c_void_p(buf.data_ptr()), h * w * 3, self.workers, status)
        if st != L.SD_OK:
            bad = [(self.paths[lo + i], status[i]) for i in range(n) if status[i] != L.SD_OK]
            raise ValueError(f"FrameFeeder: {len(bad)} frame(s) of the batch could not be read as {h}x{w} PNG / JPEG frames: {bad[:3]}")
        return buf[:n]

    def __iter__(self):
        torch = self._torch
        n = len(self.paths)
        ranges = [(a, min(a + self.batch, n)) for a in range(0, n, self.batch)]
        if not ranges:
            return
        with ThreadPoolExecutor(max_workers=1) as ahead:
            fut = ahead.submit(self._decode_into, 0, *ranges[0])
            events = [None, None]
            for k, (lo, hi) in enumerate(ranges):
                host = fut.result()
                if k + 1 < len(ranges):
                    slot = (k + 1) & 1
                    if events[slot] is not None:
                        events[slot].synchronize()          # the upload out of that staging buffer (two batches ago) has finished
                    fut = ahead.submit(self._decode_into, slot, *ranges[k + 1])
                dev = host.to(self.device, non_blocking=True)
                if self.device.type == "cuda":
                    events[k & 1] = torch.cuda.Event()
                    events[k & 1].record()
                yield dev, lo


def os_fsencode(p) -> bytes:
    import os
    return os.fsencode(p)
