"""CPU: the tool's text / PLY / PNG outputs and the focal-length sweep (§8f-4) against fixtures produced by
EXECUTING the reference's own statements (tests/golden/make_golden.py lifts them from semantic_depth.py's AST), plus
the pure-numpy pieces of the reference that the oracle restates (post_processing)."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import fusion
from semantic_depth_amd import outputs


@pytest.fixture(scope="module")
def txt(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_text_outputs.json")))


@pytest.fixture(scope="module")
def pieces(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_pieces.npz"))


def test_oracle_post_processing_equals_the_reference_function(pieces):
    """oracle/fusion.post_processing vs DepthFrame.post_processing (semantic_depth.py:656-664) executed from the reference:
    bit-exact in float64, for even / odd widths."""
    for tag in "abc":
        got = fusion.post_processing(pieces[f"pp_in_{tag}"])
        ref = pieces[f"pp_out_{tag}"]
        assert got.dtype == ref.dtype == np.float64
        assert np.array_equal(got, ref), tag


def test_times_and_distances_files_are_byte_identical(txt, tmp_path):
    t = txt["times_in"]
    times = dict(read=t["time_read_resize"], semantic=t["time_semantic"], disparity=t["time_disparity"], to3D=t["time_to3D"],
                 road=t["time_road"], rw=t["time_rw"], fences=t["time_fences"], f2f=t["time_f2f"])
    times["global"] = t["time_global"]
    name = str(tmp_path / "frame_output")
    assert open(outputs.write_times(name, times)).read() == txt["times_plain"]
    casts = {"float": float, "float64": np.float64, "float32": np.float32, "NoneType": lambda v: None}
    for tag in ("plain", "np", "f32"):
        vals = [casts[ty](v) for ty, v in zip(txt[f"distances_{tag}_types"], txt[f"distances_{tag}_in"])]
        assert open(outputs.write_distances(name, *vals)).read() == txt[f"distances_{tag}"], tag


def test_focal_sweep_files_are_byte_identical(txt, tmp_path):
    """main()'s args.f-is-None branch, semantic_depth.py:854-944, on the same synthetic distances the reference's
    statements were fed."""
    rows = {int(k): v for k, v in txt["sweep_rows"].items()}
    gt = txt["sweep_gt"]
    names = sorted(gt)

    class Depther:
        f = None

    d = Depther()

    def process(name):
        _, rw, ff = rows[d.f][names.index(name)]
        return rw, ff

    res = outputs.focal_sweep(process, gt, d, focal_lengths=[380, 580], results_directory=str(tmp_path))
    for f in (380, 580):
        assert open(tmp_path / str(f) / "data.txt").read() == txt[f"data_{f}"]
    assert open(tmp_path / "best_focal_lengths.txt").read() == txt["best_focal_lengths"]
    assert res["best_f_overall"] == 580 and d.f == 580
    assert res["per_f"][380]["mae_rw"] == pytest.approx(0.852)


def test_overlay_items_and_banner():
    """the strings / origins / banner of semantic_depth.py:346-401 and the sequence variant seq:301-327"""
    l_rw, r_rw = np.array([[-3.456, -1.5, -9.98]]), np.array([[3.5419, -1.5, -10.0]])
    l_f, r_f = np.array([[-4.1, -1.4, -10.0]]), np.array([[4.25, -1.4, -10.0]])
    banner, items = outputs.overlay_items(4032, 3024, 10.0, False, l_rw, r_rw, 6.9979, "both", l_f, r_f, 8.35)
    assert banner == ((0, 0), (4032, 604), (156, 157, 159))
    assert [i["text"] for i in items] == ["At 10.00m depth:", "4.10m to l fence", "4.25m to r fence", "Fence2Fence: 8.35m",
                                          "3.46m to road's l", "3.54m to road's r", "Road's width: 7.00m"]
    assert items[0]["org"] == (int(0.33 * 4032), int(0.05 * 3024)) and items[0]["fontScale"] == 4 and items[0]["thickness"] == 5
    assert items[4]["org"] == (int(0.01 * 4032), int(0.18 * 3024)) and items[5]["org"][0] == int(0.67 * 4032)
    _, city = outputs.overlay_items(2048, 1024, 10.0, True, l_rw, r_rw, 6.9979)
    assert len(city) == 4 and city[0]["fontScale"] == 2 and city[2]["org"][0] == int(0.68 * 2048)
    b2, it2 = outputs.overlay_items_sequence(2048, 1024, 10.0, True, l_rw, r_rw, 6.9979)
    assert b2[1] == (2048, 256) and it2[0]["text"] == "At 10.00 m depth:" and it2[0]["fontScale"] == 2.2
    assert it2[3]["text"] == "Road's width: 7.00 m" and it2[3]["org"] == (int(0.35 * 2048), int(0.22 * 1024))
    b3, it3 = outputs.overlay_items_sequence(2048, 1024, 10.0, False)
    assert b3 is None and it3[0]["color"] == (0, 255, 0) and it3[0]["text"].startswith("Cannot compute width of road at 10.00 m")
    img = np.zeros((100, 200, 3), np.uint8)
    out, _ = outputs.draw_overlay(img, ((0, 0), (200, 20), (156, 157, 159)), [])
    assert (out[:21] == (156, 157, 159)).all() and (out[21:] == 0).all() and (img == 0).all()


def _read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    p, idat, hdr = 8, b"", None
    while p < len(b):
        n, tag = struct.unpack(">I4s", b[p:p + 8])
        data = b[p + 8:p + 8 + n]
        assert struct.unpack(">I", b[p + 8 + n:p + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        if tag == b"IDAT":
            idat += data
        p += 12 + n
    w, h, depth, ctype = hdr[:4]
    ch = {0: 1, 2: 3}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, ch)


def test_png_writer_round_trips_and_swaps_bgr(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    back = _read_png(outputs.write_png(str(tmp_path / "a.png"), img))
    assert np.array_equal(back, img[..., ::-1])
    try:
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(tmp_path / "a.png")), img[..., ::-1])
    except ImportError:
        pass
    gray = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    assert np.array_equal(_read_png(outputs.write_png(str(tmp_path / "g.png"), gray))[..., 0], gray)


def test_plane_visualisation_grid_matches_the_reference(pieces, txt, golden_dir):
    """pcl.plane_grid (host numpy) vs the plane3D / colors_plane arrays the reference's remove_noise_by_fitting_plane returned
    for the same cloud and coefficients (pcl.py:104-124, :141-160, :176-195)."""
    from semantic_depth_amd import pcl
    mini = np.load(os.path.join(golden_dir, "pcl_mini.npz"))
    for axis in (0, 1, 2):
        c = mini[f"plane_a{axis}_coeff"]
        coeff = dict(Cx=c[0], Cy=c[1], Cz=c[2], C=c[3])
        p3d, cp = pcl.plane_grid(mini["road3d"], coeff, axis, [200, 190, 180])
        assert list(p3d.shape) == list(pieces[f"grid_a{axis}_shape"])
        assert [str(p3d.dtype), str(cp.dtype)] == txt[f"grid_a{axis}_dtype"]
        assert np.array_equal(p3d[:3], pieces[f"grid_a{axis}_first"]) and np.array_equal(p3d[-3:], pieces[f"grid_a{axis}_last"])
        assert np.allclose(p3d.sum(axis=0), pieces[f"grid_a{axis}_sum"], rtol=1e-12)
        assert np.array_equal(cp[:2], pieces[f"grid_a{axis}_colors_first"])
