"""CPU: the PLY writer is byte-identical to the reference's PointCloud2Ply (fixture made by
tests/golden/make_golden.py with the reference's own class)."""
import os

import numpy as np

from semantic_depth_amd.point_cloud_2_ply import PointCloud2Ply


def test_ply_bytes_match_reference(golden_dir, tmp_path, capsys):
    z = np.load(os.path.join(golden_dir, "ply_small_inputs.npz"))
    pc = PointCloud2Ply(z["pts"], z["col"], str(tmp_path / "cloud"))
    pc.add_extra_point_cloud(z["line"], z["line_col"])
    pc.prepare_and_save_point_cloud()
    got = open(tmp_path / "cloud.ply").read()
    want = open(os.path.join(golden_dir, "ply_small.ply.txt")).read()
    assert got == want
    assert "Point Cloud file generated!" in capsys.readouterr().out
    # the "infinity" filter dropped exactly the minimum-z point(s)
    n_in = len(z["pts"]) + len(z["line"])
    zs = np.concatenate([z["pts"][:, 2], z["line"][:, 2]])
    assert f"element vertex {n_in - int((zs == zs.min()).sum())}" in got
