"""CPU: the PLY writer is byte-identical to the reference's PointCloud2Ply (fixture made by
tests/golden/make_golden.py with the reference's own class)."""
import os

import numpy as np
import pytest

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


def test_native_row_formatter_equals_numpy_savetxt():
    """sd_ply_format_rows (the host helper behind write_ply) against numpy.savetxt(fh, hstack, "%f %f %f %d %d %d")
    (point_cloud_2_ply.py:70): float32 / float64 / integer coordinates, uint8 / int64 / float colours, signed zeros,
    rounding ties at the sixth decimal, float32 extremes, inf / nan, every thread count"""
    import io

    from semantic_depth_amd.point_cloud_2_ply import format_rows

    def ref(p, c):
        b = io.StringIO()
        np.savetxt(b, np.hstack([p, c]), "%f %f %f %d %d %d")
        return b.getvalue().encode()

    rng = np.random.default_rng(11)
    n = 20000
    p = (rng.normal(size=(n, 3)) * np.array([5, 1, 50])).astype(np.float32)
    c = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    edge = p[:8].copy()
    edge[0] = [np.inf, -np.inf, np.nan]
    edge[1] = [-0.0, 0.0, 1e-7]
    edge[2] = [3.4e38, -3.4e38, 0.5e-6]
    edge[3] = [2.5e-6, -2.5e-6, 0.9999995]
    edge[4] = [1.0000005, 123456.789, -1e-10]
    cases = [(p, c), (p.astype(np.float64) * 1e3 + 1e-7, c.astype(np.int64)), (edge, c[:8]), (p[:1], c[:1]),
             (np.array([[1, 2, 3]]), np.array([[255, 0, 7]])), (p[:5], c[:5].astype(np.float32) + 0.9),
             (np.array([[1e300, -1e-300, 5e-324]]), np.array([[-3, 70000, 0]]))]
    for a, b in cases:
        want = ref(a, b)
        for threads in (1, 0, 3):
            assert format_rows(a, b, threads) == want
    assert format_rows(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)) == b""
    with pytest.raises(ValueError):
        format_rows(p[:3], c[:2])
