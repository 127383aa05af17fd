"""Drop-in for the reference's ``semantic_depth_lib/point_cloud_2_ply.py``: ASCII PLY export of a coloured
point cloud.  Host-side output formatting, no GPU work (the rows come from the library's host helper sd_ply_format_rows); byte-identical
files (tests/test_ply.py pins it against the reference's own class and against numpy.savetxt).

Format quirks kept on purpose (point_cloud_2_ply.py:38-49, :70, :88): every header line after the first is indented by four
spaces exactly as the reference's triple-quoted literal is, rows are ``%f %f %f %d %d %d``, and
``prepare_and_save_point_cloud`` first drops every point whose z equals the minimum z (its "infinity" filter).
"""
from __future__ import annotations

import numpy as np

_HEADER_LINES = ["ply", "format ascii 1.0", "element vertex {vertex_count}", "property float x", "property float y",
                 "property float z", "property uchar red", "property uchar green", "property uchar blue", "end_header"]


def format_rows(points3D, colors, threads: int = 0) -> bytes:
    """the text numpy.savetxt(fh, np.hstack([points3D, colors]), "%f %f %f %d %d %d") writes (point_cloud_2_ply.py:70), produced by
    the library's host helper sd_ply_format_rows: 0.4 s -> 10-40 ms for a 150 k-point cloud"""
    import ctypes as C

    from . import _lib as L
    pts = np.asarray(points3D).reshape(-1, 3)
    col = np.asarray(colors).reshape(-1, 3)
    if len(pts) != len(col):
        raise ValueError("points3D and colors must have the same number of rows")       # (np.hstack raises the same way)
    n = len(pts)
    if n == 0:
        return b""
    # hstack promotes both to one float type; a float32 / integer value converts to double exactly, which is what '%f' formats
    xyz = np.ascontiguousarray(pts, dtype=np.float64)
    if not np.issubdtype(col.dtype, np.integer):
        if not np.isfinite(col).all():
            raise ValueError("cannot convert a non-finite colour to an integer")          # ('%d' % nan raises in savetxt)
        col = np.trunc(col)                                                                # '%d' of a float truncates toward zero
    rgb = np.ascontiguousarray(col, dtype=np.int64)
    finite = np.abs(xyz[np.isfinite(xyz)])
    digits = int(np.floor(np.log10(max(float(finite.max()) if finite.size else 1.0, 1.0)))) + 1
    row_cap = 3 * (digits + 9) + 3 * 21 + 8
    out = C.create_string_buffer(n * row_cap)
    w = L.load().sd_ply_format_rows(xyz.ctypes.data, rgb.ctypes.data, n, out, n * row_cap, threads)
    if w < 0:
        raise RuntimeError("sd_ply_format_rows failed")
    return out.raw[:w]


class PointCloud2Ply:
    #: same text as the reference's class attribute (first line flush left, the rest indented by 4 spaces, trailing indent)
    ply_header = _HEADER_LINES[0] + "\n" + "".join("    " + ln + "\n" for ln in _HEADER_LINES[1:]) + "    "

    def __init__(self, points3D, colors, output_name):
        self.points3D = np.asarray(points3D).reshape(-1, 3)
        self.colors = np.asarray(colors).reshape(-1, 3)
        self.output_name = output_name

    def write_ply(self, output_file):
        body = format_rows(self.points3D, self.colors)
        with open(output_file, "wb") as fh:
            fh.write(self.ply_header.format(vertex_count=len(self.points3D)).encode())
            fh.write(body)
        print("Point Cloud file generated!")

    def add_extra_point_cloud(self, points3D_extra, colors_extra):
        self.points3D = np.append(self.points3D, points3D_extra, axis=0)
        self.colors = np.append(self.colors, colors_extra, axis=0)

    def prepare_and_save_point_cloud(self):
        keep = self.points3D[:, 2] > self.points3D[:, 2].min()      # the reference's "infinity" filter
        self.points3D = self.points3D[keep]
        self.colors = self.colors[keep]
        self.write_ply("{}.ply".format(self.output_name))
