"""Drop-in for the reference's ``semantic_depth_lib/point_cloud_2_ply.py``: ASCII PLY export of a coloured
point cloud.  Host-side output formatting, no GPU work; byte-identical files (tests/test_ply.py pins it against the
reference's own class).

Format quirks kept on purpose (point_cloud_2_ply.py:38-49, :70, :88): every header line after the first is indented by four
spaces exactly as the reference's triple-quoted literal is, rows are ``%f %f %f %d %d %d``, and
``prepare_and_save_point_cloud`` first drops every point whose z equals the minimum z (its "infinity" filter).
"""
from __future__ import annotations

import numpy as np

_HEADER_LINES = ["ply", "format ascii 1.0", "element vertex {vertex_count}", "property float x", "property float y",
                 "property float z", "property uchar red", "property uchar green", "property uchar blue", "end_header"]


class PointCloud2Ply:
    #: same text as the reference's class attribute (first line flush left, the rest indented by 4 spaces, trailing indent)
    ply_header = _HEADER_LINES[0] + "\n" + "".join("    " + ln + "\n" for ln in _HEADER_LINES[1:]) + "    "

    def __init__(self, points3D, colors, output_name):
        self.points3D = np.asarray(points3D).reshape(-1, 3)
        self.colors = np.asarray(colors).reshape(-1, 3)
        self.output_name = output_name

    def write_ply(self, output_file):
        rows = np.hstack([self.points3D, self.colors])
        with open(output_file, "w") as fh:
            fh.write(self.ply_header.format(vertex_count=len(rows)))
            np.savetxt(fh, rows, "%f %f %f %d %d %d")
        print("Point Cloud file generated!")

    def add_extra_point_cloud(self, points3D_extra, colors_extra):
        self.points3D = np.append(self.points3D, points3D_extra, axis=0)
        self.colors = np.append(self.colors, colors_extra, axis=0)

    def prepare_and_save_point_cloud(self):
        keep = self.points3D[:, 2] > self.points3D[:, 2].min()      # the reference's "infinity" filter
        self.points3D = self.points3D[keep]
        self.colors = self.colors[keep]
        self.write_ply("{}.ply".format(self.output_name))
