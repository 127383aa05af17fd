#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own point-cloud library.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports /root/reference/semantic_depth_lib/pcl.py (the one reference module that imports without
TensorFlow/OpenCV/Open3D), feeds it inputs produced by the oracle's synthetic ground-plane scene
 and stores inputs + the reference's outputs:

  pcl_mini.npz        64x128 miniature, full arrays for every pcl function on the road-width path,
                      plus the fence-side helpers and the edge cases (empty window, MAD == 0).
  pcl_full.json       512x1024 scene: generator parameters + counts / coefficients / end points /
                      order-sensitive checksums of the reference's outputs (arrays are regenerated from
                      the seed by the oracle's scene generator at test time).

  ref_pieces.npz /    outputs of pieces of semantic_depth.py that are pure numpy / pure formatting, EXECUTED from the
  ref_text_outputs.json reference's own AST (the module itself cannot be imported: tensorflow / cv2 / open3d are absent):
                      DepthFrame.post_processing (:656-664) on seeded disparity pairs; the ``_times.txt`` /
                      ``_distances.txt`` writer statements of process_frame (:445-458) and the MAE / ``data.txt`` /
                      ``best_focal_lengths.txt`` statements of main() (:907-944) on synthetic numbers; the plane
                      visualisation grid of pcl.remove_noise_by_fitting_plane (digest).

Fixtures are data (inputs and expected outputs); no reference source is copied.
"""
import ast
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from semantic_depth_lib import pcl as ref  # noqa: E402  (the reference)
from oracle import fusion, pipeline  # noqa: E402


def checksum(a: np.ndarray) -> str:
    """Order-sensitive digest: sum_i (i+1)*bits(a_i) mod 2^64 over the raw little-endian words."""
    b = np.ascontiguousarray(a).view(np.uint8).ravel()
    pad = (-len(b)) % 4
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    wds = b.view(np.uint32).astype(np.uint64)
    idx = np.arange(1, len(wds) + 1, dtype=np.uint64)
    return hex(int((wds * idx).sum(dtype=np.uint64)))


def ref_chain(points, colors, depth=10.0):
    """the road chain of semantic_depth.py:206-219,254-259 with the reference's functions (Open3D absent)."""
    out = {}
    p, c = ref.remove_from_to(points, colors, 2, 0.0, 7.0)
    out["zcut"] = (p, c)
    p, c = ref.remove_noise_by_mad(p, c, 1, 15.0)
    out["mad_y"] = (p, c)
    p, c = ref.remove_noise_by_mad(p, c, 0, 2.0)
    out["mad_x"] = (p, c)
    p, c, _, _, coeff = ref.remove_noise_by_fitting_plane(p, c, axis=1, threshold=5.0, plane_color=[200, 200, 200])
    out["plane"] = (p, c)
    out["coeff"] = coeff
    l, r = ref.get_end_points_of_road(p.astype(np.float64), depth - 0.02)
    out["left"], out["right"] = l, r
    return out


def scene_inputs(h, w, seed, f, fences=False):
    dp, road, fence, frame, cam = pipeline.synthetic_scene(h, w, seed=seed, f=f, fences=fences)
    fz = fusion.fuse(dp, road, fence, frame, **cam)
    return dp, road, fence, frame, cam, fz


def _ref_ast():
    return ast.parse(open("/root/reference/semantic_depth.py").read())


def _find(tree, cls, fn):
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == fn:
                    return f
        if cls is None and isinstance(node, ast.FunctionDef) and node.name == fn:
            return node
    raise KeyError((cls, fn))


def _exec(nodes, ns):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, "<reference AST>", "exec"), ns)
    return ns


def reference_pieces():
    """run the reference's own pure-numpy / pure-formatting statements, lifted from its AST"""
    import tempfile
    tree = _ref_ast()
    z, txt = {}, {}
    # --- DepthFrame.post_processing, semantic_depth.py:656-664
    fn = _find(tree, "DepthFrame", "post_processing")
    ns = _exec([fn], {"np": np})
    rng = np.random.default_rng(11)
    for tag, (h, w) in (("a", (8, 40)), ("b", (16, 128)), ("c", (5, 21))):
        d = (0.3 * rng.random((2, h, w))).astype(np.float32)
        z[f"pp_in_{tag}"] = d
        z[f"pp_out_{tag}"] = ns["post_processing"](None, d)                 # f64, the caller casts (:676)
    # --- the with-open blocks that write <name>_times.txt / <name>_distances.txt, :445-458
    pf = _find(tree, "FrameProcessor", "process_frame")
    withs = [n for n in ast.walk(pf) if isinstance(n, ast.With)]
    times = dict(time_read_resize=0.348812, time_semantic=0.15283203125, time_disparity=0.023, time_to3D=0.0106,
                 time_road=0.0845, time_rw=0.0012, time_fences=0.0146, time_f2f=0.0015, time_global=0.6375)
    cases = {"plain": dict(dist_rw=4.412345678901234, dist_f2f=4.57), "np": dict(dist_rw=np.float64(6.9976945), dist_f2f=np.float64(7.25)),
             "f32": dict(dist_rw=np.float32(5.25), dist_f2f=None)}
    with tempfile.TemporaryDirectory() as td:
        for tag, dist in cases.items():
            class S:      # stands in for ``self``
                output_name = os.path.join(td, tag)
            _exec(withs, dict(self=S, **times, **dist))
            txt[f"times_{tag}"] = open(S.output_name + "_times.txt").read()
            txt[f"distances_{tag}"] = open(S.output_name + "_distances.txt").read()
            txt[f"distances_{tag}_in"] = [None if v is None else float(v) for v in (dist["dist_rw"], dist["dist_f2f"])]
            txt[f"distances_{tag}_types"] = [type(v).__name__ for v in (dist["dist_rw"], dist["dist_f2f"])]
        txt["times_in"] = times
        # --- MAE + data.txt + best_focal_lengths.txt, main() :854-944: the body of ``for f in focal_lengths`` after the inner
        #     frame loop, and the with-open that follows the loop
        mn = _find(tree, None, "main")
        floop = [n for n in ast.walk(mn) if isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == "f"][0]
        tail = [st for st in floop.body if st.lineno >= 907]
        after = [n for n in ast.walk(mn) if isinstance(n, ast.With) and n.lineno > floop.end_lineno and n.lineno < 950]
        input_frames = {"test_1.png": 5.3, "test_2.png": 4.4, "test_3.png": 5.4, "test_4.png": 3.1, "test_5.png": 4.6}
        rows = {380: [(5.3, 4.91, 6.02), (4.4, 3.12, 5.5), (5.4, 4.41, 4.57), (3.1, 3.3, 2.2), (4.6, 6.0, 5.1)],
                580: [(5.3, 5.6, 5.1), (4.4, 4.0, 4.1), (5.4, 5.55, 5.2), (3.1, 2.4, 3.0), (4.6, 4.9, 4.8)]}
        ns = dict(np=np, os=os, input_frames=input_frames, best_mae_rw=-1, best_f_rw=None, best_mae_f2f=-1, best_f_f2f=None,
                  best_mae_overall=-1, best_f_overall=None, print=lambda *a, **k: None)
        for f, rws in rows.items():
            ns["f"] = f
            ns["f_directory"] = os.path.join(td, str(f))
            os.makedirs(ns["f_directory"])
            ns["all_data"] = [[real, rw, ff, abs(real - rw), abs(real - ff)] for real, rw, ff in rws]
            _exec(tail, ns)
            txt[f"data_{f}"] = open(os.path.join(td, str(f), "data.txt")).read()
        ns["results_directory"] = td
        _exec(after[:1], ns)
        txt["best_focal_lengths"] = open(os.path.join(td, "best_focal_lengths.txt")).read()
        txt["sweep_rows"] = {str(k): v for k, v in rows.items()}
        txt["sweep_gt"] = input_frames
    return z, txt


def main():
    z_ref, txt_ref = reference_pieces()
    # ---------------- miniature: full arrays ----------------
    dp, road, fence, frame, cam, fz = scene_inputs(64, 128, seed=7, f=125.0, fences=True)
    pts, col = fz["road3d"], fz["road_rgb"]
    ch = ref_chain(pts, col)
    z = {
        "disp_pair": dp, "road_mask": road, "fence_mask": fence, "frame_bgr": frame,
        "cam": np.array([cam["cx"], cam["cy"], cam["f"], cam["b"], cam["disp_mult"]], np.float64),
        "road3d": pts, "road_rgb": col, "fence3d": fz["fence3d"], "fence_rgb": fz["fence_rgb"],
    }
    for k in ("zcut", "mad_y", "mad_x", "plane"):
        z[f"{k}_pts"], z[f"{k}_col"] = ch[k]
    z["plane_coeff"] = np.array([ch["coeff"][k] for k in ("Cx", "Cy", "Cz", "C")], np.float64)
    z["left_pts"], z["right_pts"] = ch["left"], ch["right"]
    # MAD with tighter thresholds so that something is actually removed
    for axis, thr in ((0, 1.0), (1, 2.0), (2, 0.8)):
        p, c = ref.remove_noise_by_mad(pts, col, axis, thr)
        z[f"mad_a{axis}_pts"], z[f"mad_a{axis}_col"] = p, c
    # plane fit on all three axes with a tight threshold (fence planes use axis 0, seq:262-275)
    for axis, thr in ((0, 0.5), (1, 0.02), (2, 3.0)):
        p, c, p3d, cp3d, coeff = ref.remove_noise_by_fitting_plane(pts, col, axis=axis, threshold=thr, plane_color=[200, 190, 180])
        z[f"plane_a{axis}_pts"], z[f"plane_a{axis}_col"] = p, c
        z[f"plane_a{axis}_coeff"] = np.array([coeff[k] for k in ("Cx", "Cy", "Cz", "C")], np.float64)
        # the visualisation grid (pcl.py:104-124 ...): shape, dtype and an order-sensitive digest
        z_ref[f"grid_a{axis}_shape"] = np.array(p3d.shape)
        z_ref[f"grid_a{axis}_first"] = p3d[:3].copy()
        z_ref[f"grid_a{axis}_last"] = p3d[-3:].copy()
        z_ref[f"grid_a{axis}_sum"] = p3d.sum(axis=0)
        z_ref[f"grid_a{axis}_colors_first"] = cp3d[:2].copy()
        txt_ref[f"grid_a{axis}_dtype"] = [str(p3d.dtype), str(cp3d.dtype)]
    # empty depth window -> (None, None)
    l, r = ref.get_end_points_of_road(ch["plane"][0].astype(np.float64), 500.0)
    z["empty_window_is_none"] = np.array([l is None, r is None])
    # MAD == 0: constant-y cloud -> penalty nan/inf -> everything with |dev|==0 gives nan -> dropped
    flat = pts.copy()
    flat[:, 1] = np.float32(-1.5)
    with np.errstate(all="ignore"):
        p, c = ref.remove_noise_by_mad(flat, col, 1, 15.0)
    z["mad0_in"], z["mad0_pts"] = flat, p
    # threshold edges: a cut that float32 cannot represent, points at the float32 neighbours of every cut, and points whose
    # plane residual sits within rounding of the threshold.  Captured under this container's numpy (NEP 50: the python-float
    # threshold is compared in the array's float32; the plane residual is evaluated in float64) -- the reference's pinned numpy
    # 1.16 promotes differently in exactly these two places.
    edge = []
    for cut in (7.0, 7.1, 6.999999, 35.0, 0.1):
        c32 = np.float32(-cut)
        for k in range(-3, 4):
            v = c32
            for _ in range(abs(k)):
                v = np.nextafter(v, np.float32(-np.inf if k < 0 else np.inf), dtype=np.float32)
            edge.append([0.5 * k, -1.5, v])
            edge.append([0.5 * k, -1.5, -v])
    edge = np.asarray(edge, np.float32)
    ecol = (np.arange(len(edge) * 3) % 251).astype(np.uint8).reshape(-1, 3)
    z["edge_in"], z["edge_col"] = edge, ecol
    for tag, cut in (("70", 7.0), ("71", 7.1), ("69", 6.999999)):
        p, c = ref.remove_from_to(edge, ecol, 2, 0.0, cut)
        z[f"edge_rft_{tag}_pts"], z[f"edge_rft_{tag}_col"] = p, c
    p, c = ref.threshold_complete(edge, ecol, 2, 35.0)
    z["edge_thr35_pts"] = p
    p, c = ref.threshold_complete(edge, ecol, 2, 0.1)
    z["edge_thr01_pts"] = p
    # plane y = 0.25 x - 0.125 z - 1.5 exactly representable; outliers placed at threshold +- a few ulps of the residual
    rngp = np.random.default_rng(3)
    base = np.stack([rngp.integers(-64, 64, 400) / 8.0, np.zeros(400), -rngp.integers(40, 200, 400) / 4.0], 1)
    base[:, 1] = 0.25 * base[:, 0] - 0.125 * base[:, 2] - 1.5
    sym = base.copy(); sym[:, 1] = base[:, 1]                      # (the fit is exact up to rounding: residuals of the inliers ~1e-16)
    offs = np.float64([0.75 - 2e-7, 0.75 - 6e-8, 0.75, 0.75 + 6e-8, 0.75 + 2e-7, -(0.75 - 6e-8), -0.75, -(0.75 + 6e-8)])
    outl = base[:len(offs)].copy(); outl[:, 1] += offs
    # mirror outliers so that the fitted plane stays the exact one (their residuals cancel in the normal equations)
    mirror = outl.copy(); mirror[:, 1] = 2 * base[:len(offs), 1] - outl[:, 1]
    pl = np.concatenate([base, outl, mirror]).astype(np.float32)
    pcol = (np.arange(len(pl) * 3) % 253).astype(np.uint8).reshape(-1, 3)
    p, c, _, _, coeff = ref.remove_noise_by_fitting_plane(pl, pcol, axis=1, threshold=0.75)
    z["edge_plane_in"], z["edge_plane_col"], z["edge_plane_pts"] = pl, pcol, p
    z["edge_plane_coeff"] = np.array([coeff[k] for k in ("Cx", "Cy", "Cz", "C")], np.float64)
    # fence-side helpers
    fp, fc = fz["fence3d"], fz["fence_rgb"]
    p, c = ref.threshold_complete(fp, fc, 2, 35.0)
    z["thr_pts"], z["thr_col"] = p, c
    a, ac, b, bc = ref.extract_pcls(p, c)
    z["split_left"], z["split_left_col"], z["split_right"], z["split_right_col"] = a, ac, b, bc
    z["dist3d"] = np.float64(ref.compute_distance_in_3D(np.array([[1.0, 2.0, 3.0]]), np.array([[-2.0, 0.5, 7.0]])))
    lp, rp = ch["left"][:1].copy(), ch["right"][:1].copy()
    z["line_in_left"], z["line_in_right"] = lp.copy(), rp.copy()
    line, lcol = ref.create_3Dline_from_3Dpoints(lp, rp, [250, 0, 0])
    z["line"], z["line_col"], z["line_left_after"], z["line_right_after"] = line, lcol, lp, rp
    # fence chain of semantic_depth.py:273-309 with the reference's functions (plane intersection cannot run on numpy 2)
    def fence_chain(fp, fc, tag, store):
        p1, c1 = ref.remove_noise_by_mad(fp, fc, 1, 5.0)
        p2, c2 = ref.threshold_complete(p1, c1, 2, 35.0)
        l, lc, r, rc = ref.extract_pcls(p2, c2)
        l1, lc1 = ref.remove_noise_by_mad(l, lc, 0, 5.0)
        l2, lc2, _, _, cl = ref.remove_noise_by_fitting_plane(l1, lc1, axis=0, threshold=1.0, plane_color=[40, 70, 40])
        r1, rc1 = ref.remove_noise_by_mad(r, rc, 0, 1.0)
        r2, rc2, _, _, cr = ref.remove_noise_by_fitting_plane(r1, rc1, axis=0, threshold=1.0, plane_color=[40, 70, 40])
        store(tag, dict(mad_y=p1, thr=p2, left=l, right=r, left_final=l2, right_final=r2,
                        plane_left=np.array([cl[k] for k in ("Cx", "Cy", "Cz", "C")], np.float64),
                        plane_right=np.array([cr[k] for k in ("Cx", "Cy", "Cz", "C")], np.float64),
                        mean_x=np.float32(np.mean(p2[:, 0]))))
    fence_chain(fp, fc, "fc", lambda tag, d: z.update({f"{tag}_{k}": v for k, v in d.items()}))
    np.savez_compressed(os.path.join(HERE, "pcl_mini.npz"), **z)
    np.savez_compressed(os.path.join(HERE, "ref_pieces.npz"), **z_ref)
    with open(os.path.join(HERE, "ref_text_outputs.json"), "w") as fh:
        json.dump(txt_ref, fh, indent=1, sort_keys=True)

    # ---------------- PLY writer: bytes written by the reference's own class ----------------
    from semantic_depth_lib.point_cloud_2_ply import PointCloud2Ply as RefPly
    import tempfile, io, contextlib
    pts_p, col_p = ch["plane"][0][:40].astype(np.float64), ch["plane"][1][:40]
    lp2, rp2 = ch["left"][:1].copy(), ch["right"][:1].copy()
    line2, lcol2 = ref.create_3Dline_from_3Dpoints(lp2, rp2, [250, 0, 0])
    with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(io.StringIO()):
        pc = RefPly(pts_p, col_p, os.path.join(td, "cloud"))
        pc.add_extra_point_cloud(line2[:25], lcol2[:25])
        pc.prepare_and_save_point_cloud()
        ply_text = open(os.path.join(td, "cloud.ply")).read()
    with open(os.path.join(HERE, "ply_small.ply.txt"), "w") as fh:
        fh.write(ply_text)
    np.savez_compressed(os.path.join(HERE, "ply_small_inputs.npz"), pts=pts_p, col=col_p, line=line2[:25], line_col=lcol2[:25])

    # ---------------- full size: digests ----------------
    full = {"scene": dict(h=512, w=1024, seed=1234, f=1000.0, fences=False)}
    dp, road, fence, frame, cam, fz = scene_inputs(512, 1024, seed=1234, f=1000.0)
    ch = ref_chain(fz["road3d"], fz["road_rgb"])
    full["n_road"] = int(fz["road3d"].shape[0])
    full["road3d_checksum"] = checksum(fz["road3d"])
    full["road_rgb_checksum"] = checksum(fz["road_rgb"])
    for k in ("zcut", "mad_y", "mad_x", "plane"):
        full[f"n_{k}"] = int(ch[k][0].shape[0])
        full[f"{k}_checksum"] = checksum(ch[k][0])
    full["plane_coeff"] = {k: float(v) for k, v in ch["coeff"].items()}
    full["x_left"] = float(ch["left"][0][0])
    full["x_right"] = float(ch["right"][0][0])
    full["left_pt"] = [float(v) for v in ch["left"][0]]
    full["right_pt"] = [float(v) for v in ch["right"][0]]
    full["dist_rw"] = float(abs(ch["left"][0][0] - ch["right"][0][0]))
    # fence scene at full size: digests of the reference's fence chain
    dpf, roadf, fencef, framef, camf, fzf = scene_inputs(512, 1024, seed=77, f=1000.0, fences=True)
    fdig = {}
    def store_digest(tag, d):
        for k, v in d.items():
            if v.ndim == 2:
                fdig[f"n_{k}"] = int(v.shape[0]); fdig[f"{k}_checksum"] = checksum(v)
            elif v.ndim == 1:
                fdig[k] = [float(x) for x in v]
            else:
                fdig[k] = float(v)
    fence_chain(fzf["fence3d"], fzf["fence_rgb"], "ff", store_digest)
    fdig["scene"] = dict(h=512, w=1024, seed=77, f=1000.0, fences=True)
    fdig["n_fence"] = int(fzf["fence3d"].shape[0])
    full["fence"] = fdig
    full["numpy"] = np.__version__
    with open(os.path.join(HERE, "pcl_full.json"), "w") as fh:
        json.dump(full, fh, indent=1, sort_keys=True)
    print("wrote pcl_mini.npz, pcl_full.json;", {k: full[k] for k in ("n_road", "n_zcut", "n_plane", "dist_rw")})


if __name__ == "__main__":
    main()
