"""On-disk outputs of the tool and the focal-length sweep driver (§8f-4) — host-side formatting only.

    write_times / write_distances   <- semantic_depth.py:445-458  (``<name>_times.txt`` / ``<name>_distances.txt``)
    overlay_items / draw_overlay    <- semantic_depth.py:339-406  (banner + the text the reference draws; seq:301-327)
    save_frame_outputs              <- semantic_depth.py:339-441  (``_only_segmentation.png``, the annotated ``.png``, ``_ROAD`` / ``_FENCE`` /
                                       combined / ``_ALL`` PLY files, incl. the three visualisation planes of the combined cloud)
    road_plane_grid / fence_plane_grids <- the ``plane3D`` arrays of :215-219, :294-309 rebuilt through the reference's own pcl call sequence
    focal_sweep                     <- semantic_depth.py:854-944  (``results/<f>/data.txt``, ``best_focal_lengths.txt``)
    write_png                       <- cv2.imwrite(...png) at :406 (zlib-deflated 8-bit RGB; pixel-identical, not byte-identical)

The text files are byte-identical to what the reference's own statements write (tests/test_outputs.py holds fixtures produced
by executing those statements, extracted from the reference by ``tests/golden/make_golden.py``).  Glyph rendering of
``cv2.putText`` (Hershey fonts, part of OpenCV) is NOT reproduced: ``draw_overlay`` paints the banner rectangle and returns
the text items (string, origin, scale, colour, thickness) the reference passes to putText; they are also written next to the
image as ``<name>_overlay.json``.
"""
from __future__ import annotations

import json
import os
import struct
import zlib

import numpy as np

from . import pcl
from .point_cloud_2_ply import PointCloud2Ply

TIME_KEYS = ("read", "semantic", "disparity", "to3D", "road", "rw", "fences", "f2f", "global")


def write_times(output_name: str, t: dict) -> str:
    """semantic_depth.py:445-454.  ``t``: seconds under the keys of TIME_KEYS."""
    path = "{}_times.txt".format(output_name)
    with open(path, "w") as f:
        f.write("Time read:       {}\n".format(t["read"]))
        f.write("Time semantic:   {}\n".format(t["semantic"]))
        f.write("Time disparity:  {}\n".format(t["disparity"]))
        f.write("Time to3D:       {}\n".format(t["to3D"]))
        f.write("Time road:       {}\n".format(t["road"]))
        f.write("Time rw:      {}\n".format(t["rw"]))
        f.write("Time fences:     {}\n".format(t["fences"]))
        f.write("Time f2f:   {}\n".format(t["f2f"]))
        f.write("Time global:     {}\n".format(t["global"]))
    return path


def write_distances(output_name: str, dist_rw, dist_f2f) -> str:
    """semantic_depth.py:456-458"""
    path = "{}_distances.txt".format(output_name)
    with open(path, "w") as f:
        f.write("rw distance:    {}\n".format(dist_rw))
        f.write("f2f distance: {}\n".format(dist_f2f))
    return path


# ------------------------------------------------------------------------------------------------ overlay
def overlay_items(w: int, h: int, depth: float, is_city: bool, left_pt_rw, right_pt_rw, dist_rw, approach: str = "rw",
                  left_pt_f2f=None, right_pt_f2f=None, dist_f2f=None):
    """the cv2.rectangle / cv2.putText calls of semantic_depth.py:346-401 as data: (banner, [items]).
    banner = ((x0, y0), (x1, y1), bgr); item = dict(text, org, fontFace, fontScale, color, thickness)."""
    if is_city:
        thickness, fontScale, left, right, middle = 2, 2, 0.01, 0.68, 0.33
    else:
        thickness, fontScale, left, right, middle = 5, 4, 0.01, 0.67, 0.33
    h_zero, h_first, h_second = 0.05 * h, 0.12 * h, 0.18 * h
    banner = ((0, 0), (w, int(0.2 * h)), (156, 157, 159))

    def item(text, x, y):
        return dict(text=text, org=(int(x * w), int(y)), fontFace=16, fontScale=fontScale, color=(255, 255, 255), thickness=thickness)

    items = [item("At {:.2f}m depth:".format(depth), middle, h_zero)]
    if approach == "both":
        items.append(item("{:.2f}m to l fence".format(-left_pt_f2f[0][0]), left, h_first))
        items.append(item("{:.2f}m to r fence".format(right_pt_f2f[0][0]), right, h_first))
        items.append(item("Fence2Fence: {:.2f}m".format(dist_f2f), middle, h_first))
    items.append(item("{:.2f}m to road's l".format(-left_pt_rw[0][0]), left, h_second))
    items.append(item("{:.2f}m to road's r".format(right_pt_rw[0][0]), right, h_second))
    items.append(item("Road's width: {:.2f}m".format(dist_rw), middle, h_second))
    return banner, items


def overlay_items_sequence(w: int, h: int, depth: float, line_found: bool, left_pt_rw=None, right_pt_rw=None, dist_rw=None):
    """the sequence tool's variant, semantic_depth_cityscapes_sequence.py:301-327 (fontScale 2 / 2.2, 25 % banner, or the
    green 'Cannot compute' line and no banner)."""
    thickness, fontScale = 2, 2

    def item(text, x, y, scale, color=(255, 255, 255)):
        return dict(text=text, org=(int(x * w), int(y * h)), fontFace=16, fontScale=scale, color=color, thickness=thickness)

    if not line_found:
        return None, [item("Cannot compute width of road at {:.2f} m depth:".format(depth), 0.28, 0.035, fontScale + 0.2, (0, 255, 0))]
    banner = ((0, 0), (w, int(0.25 * h)), (156, 157, 159))
    return banner, [item("At {:.2f} m depth:".format(depth), 0.36, 0.05, fontScale + 0.2),
                    item("{:.2f}m to road's left end".format(-left_pt_rw[0][0]), 0.05, 0.13, fontScale),
                    item("{:.2f}m to road's right end".format(right_pt_rw[0][0]), 0.5, 0.13, fontScale),
                    item("Road's width: {:.2f} m".format(dist_rw), 0.35, 0.22, fontScale)]


def draw_overlay(segmented_frame: np.ndarray, banner, items):
    """cv2.rectangle(img, pt1, pt2, color, -1) of :346 (both corners inclusive, clipped to the image); text is returned, not
    rasterised (module docstring)."""
    img = np.array(segmented_frame, copy=True)
    if banner is not None:
        (x0, y0), (x1, y1), col = banner
        img[max(y0, 0):min(y1, img.shape[0] - 1) + 1, max(x0, 0):min(x1, img.shape[1] - 1) + 1] = np.asarray(col, img.dtype)
    return img, items


def write_png(path: str, img_bgr: np.ndarray, level: int = 3) -> str:
    """8-bit PNG of a BGR (cv2 convention) or single-channel image; filter type 0 rows, one IDAT."""
    a = np.ascontiguousarray(img_bgr, dtype=np.uint8)
    if a.ndim == 3:
        a = a[..., ::-1]                       # cv2 stores BGR, PNG is RGB
        ctype, ch = 2, 3
    else:
        ctype, ch = 0, 1
    h, w = a.shape[:2]
    raw = np.zeros((h, 1 + w * ch), np.uint8)
    raw[:, 1:] = a.reshape(h, w * ch)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), level)))
        f.write(chunk(b"IEND", b""))
    return path


# ------------------------------------------------------------------------------------------------ per-frame outputs
def road_plane_grid(road3D, road_colors, z_cut: float = 7.0, mad_y: float = 15.0, mad_x: float = 2.0, plane_thr: float = 5.0):
    """``road_plane3D, road_colors_plane`` of semantic_depth.py:215-219: the 5 cm visualisation lattice over the bounding box of the
    cloud that ENTERS the plane fit (after the z-cut and the two MAD filters, :206-212), lifted onto the fitted plane.  The batched
    road chain keeps only the filtered cloud, so the save path replays the reference's own call sequence through the pcl module
    (the same kernels, one cloud at a time) on the gathered road cloud.  (None, None) when a filter empties the cloud."""
    p3, c = np.asarray(road3D), np.asarray(road_colors)
    if not len(p3):
        return None, None
    p3, c = pcl.remove_from_to(p3, c, 2, 0.0, z_cut)
    for axis, thr in ((1, mad_y), (0, mad_x)):
        if not len(p3):
            return None, None
        p3, c = pcl.remove_noise_by_mad(p3, c, axis, thr)
    if not len(p3):
        return None, None
    _, _, plane3D, colors_plane, _ = pcl.remove_noise_by_fitting_plane(p3, c, axis=1, threshold=plane_thr, plane_color=[200, 200, 200])
    return plane3D, colors_plane


def fence_plane_grids(fence3D, fence_colors, mad_y: float = 5.0, z_max: float = 35.0, mad_left: float = 5.0, mad_right: float = 1.0,
                      plane_thr: float = 1.0):
    """``fence_left_plane3D, fence_left_colors_plane, fence_right_plane3D, fence_right_colors_plane`` of semantic_depth.py:279-309,
    replayed through the pcl module like road_plane_grid.  A side that runs empty gives (None, None) for that side."""
    out = [None, None, None, None]
    p3, c = np.asarray(fence3D), np.asarray(fence_colors)
    if not len(p3):
        return tuple(out)
    p3, c = pcl.remove_noise_by_mad(p3, c, 1, mad_y)
    if not len(p3):
        return tuple(out)
    p3, c = pcl.threshold_complete(p3, c, 2, z_max)
    if not len(p3):
        return tuple(out)
    l3, lc, r3, rc = pcl.extract_pcls(p3, c)
    for i, (q3, qc, thr) in enumerate(((l3, lc, mad_left), (r3, rc, mad_right))):
        if not len(q3):
            continue
        q3, qc = pcl.remove_noise_by_mad(q3, qc, 0, thr)
        if not len(q3):
            continue
        _, _, out[2 * i], out[2 * i + 1], _ = pcl.remove_noise_by_fitting_plane(q3, qc, axis=0, threshold=plane_thr, plane_color=[40, 70, 40])
    return tuple(out)


def resize_to_original(segmented_frame: np.ndarray, original_width: int, original_height: int) -> np.ndarray:
    """``cv2.resize(segmented_frame, (original_width, original_height), interpolation=cv2.INTER_CUBIC)`` of semantic_depth.py:341-342,
    on the GPU (sd_resize_cubic_u8, the kernel of the input stage)."""
    import torch
    e = pcl._eng()
    fr = torch.from_numpy(np.ascontiguousarray(segmented_frame, dtype=np.uint8))[None].to(e.device)
    return e.resize_cubic(fr, original_height, original_width)[0].cpu().numpy()


def save_frame_outputs(output_name: str, res: dict, depth: float, approach: str = "rw", segmented_frame: np.ndarray | None = None,
                       is_city: bool = False, times: dict | None = None, road_plane3D=None, road_colors_plane=None,
                       points3D_all=None, colors_all=None, original_size: tuple | None = None, params=None, fence_params=None):
    """what FrameProcessor.process_frame writes when --save_data is set (semantic_depth.py:339-458), from the dict
    ``api.FrameProcessor.process_frame(..., want_clouds=True)`` returns.  Returns the list of files written.
    ``original_size`` = (original_height, original_width): the overlay is cubic-resized back to it before anything is drawn (:341);
    ``params`` / ``fence_params`` (engine.RoadWidthParams / FenceParams): the chain literals the visualisation planes are rebuilt with."""
    files = []
    rec = res["record"]
    if not rec["found"]:
        # semantic_depth.py indexes left_pt_rw[0] unconditionally (:259) and dies on (None, None); the sequence tool guards it
        # (seq:232-234).  Here: the same TypeError the reference raises, before anything is written.
        raise TypeError("'NoneType' object is not subscriptable (no road point in the depth window: left_pt_rw is None)")
    left_rw, right_rw = rec["left_pt"].astype(np.float64)[None, :], rec["right_pt"].astype(np.float64)[None, :]
    dist_rw = res["dist_rw"]
    line_rw, colors_line_rw = pcl.create_3Dline_from_3Dpoints(left_rw.copy(), right_rw.copy(), [250, 0, 0])
    line_rw[:, 2] += 0.2                                   # :265 "for better visualization, shift it a bit"
    f2 = res.get("f2f_record")
    both = approach == "both" and f2 is not None
    if both:
        left_f2f, right_f2f = f2["left_pt"][None, :].copy(), f2["right_pt"][None, :].copy()
        line_f2f, colors_line_f2f = pcl.create_3Dline_from_3Dpoints(left_f2f.copy(), right_f2f.copy(), [0, 255, 0])
    if segmented_frame is not None:
        if original_size is not None and tuple(segmented_frame.shape[:2]) != tuple(original_size):
            segmented_frame = resize_to_original(segmented_frame, int(original_size[1]), int(original_size[0]))     # :341-342
        files.append(write_png("{}_only_segmentation.png".format(output_name), segmented_frame))                  # :345
        h, w = segmented_frame.shape[:2]
        banner, items = overlay_items(w, h, depth, is_city, left_rw, right_rw, dist_rw, "both" if both else "rw",
                                      left_f2f if both else None, right_f2f if both else None, res.get("dist_f2f"))
        img, items = draw_overlay(segmented_frame, banner, items)
        files.append(write_png("{}.png".format(output_name), img))
        with open("{}_overlay.json".format(output_name), "w") as f:
            json.dump(dict(banner=banner, items=items), f)
        files.append("{}_overlay.json".format(output_name))
    road3D, road_colors = res["road3D_final"].astype(np.float64), res["road_colors_final"]
    pc = PointCloud2Ply(road3D, road_colors, "{}_ROAD".format(output_name))                 # :408-410
    pc.prepare_and_save_point_cloud()
    files.append("{}_ROAD.ply".format(output_name))
    if both and "fence3D_left" in res:                                                      # :412-415
        pc = PointCloud2Ply(res["fence3D_left"], res["fence_left_colors"], "{}_FENCE".format(output_name))
        pc.add_extra_point_cloud(res["fence3D_right"], res["fence_right_colors"])
        pc.prepare_and_save_point_cloud()
        files.append("{}_FENCE.ply".format(output_name))
    pc = PointCloud2Ply(road3D, road_colors, output_name)                                   # :421-434
    if road_plane3D is None and "road3D" in res:                                            # the visualisation plane of :215-219
        kw = {k: getattr(params, k) for k in ("z_cut", "mad_y", "mad_x", "plane_thr")} if params is not None else {}
        road_plane3D, road_colors_plane = road_plane_grid(res["road3D"], res["road_colors"], **kw)
    if road_plane3D is not None:
        pc.add_extra_point_cloud(road_plane3D, road_colors_plane)
    pc.add_extra_point_cloud(line_rw, colors_line_rw)
    if both and "fence3D_left" in res:
        pc.add_extra_point_cloud(res["fence3D_left"], res["fence_left_colors"])
        pc.add_extra_point_cloud(res["fence3D_right"], res["fence_right_colors"])
        if "fence3D" in res:                                                                # :428-431
            kw = ({k: getattr(fence_params, k) for k in ("mad_y", "z_max", "mad_left", "mad_right", "plane_thr")} if fence_params is not None else {})
            lp, lcp, rp, rcp = fence_plane_grids(res["fence3D"], res["fence_colors"], **kw)
            if lp is not None:
                pc.add_extra_point_cloud(lp, lcp)
            if rp is not None:
                pc.add_extra_point_cloud(rp, rcp)
        pc.add_extra_point_cloud(line_f2f, colors_line_f2f)
    pc.prepare_and_save_point_cloud()
    files.append("{}.ply".format(output_name))
    if points3D_all is not None:                                                            # :437-441
        pc = PointCloud2Ply(np.asarray(points3D_all).reshape(-1, 3), np.asarray(colors_all).reshape(-1, 3), "{}_ALL".format(output_name))
        pc.add_extra_point_cloud(line_rw, colors_line_rw)
        if both:
            pc.add_extra_point_cloud(line_f2f, colors_line_f2f)
        pc.prepare_and_save_point_cloud()
        files.append("{}_ALL.ply".format(output_name))
    if times is not None:
        files.append(write_times(output_name, times))
    files.append(write_distances(output_name, dist_rw, res.get("dist_f2f")))
    return files


# ------------------------------------------------------------------------------------------------ per-frame metrics log
def append_metrics_jsonl(path: str, name: str, res: dict, times: dict | None = None, extra: dict | None = None) -> str:
    """one JSON line per processed frame (SURVEY §5 "Metrics / logging": the reference only prints): frame name, both distances, the
    kept-point count after every stage of the road chain, the plane, the end points, the stage times when given.  ``res`` is the dict
    api.FrameProcessor.process_frame returns."""
    rec = res["record"]
    line = {"frame": name, "dist_rw": res.get("dist_rw"), "dist_f2f": res.get("dist_f2f"), "found": bool(rec["found"]),
            "points": {k: int(rec[k]) for k in ("n_road", "n_zcut", "n_mad_y", "n_mad_x", "n_plane", "n_sor", "n_ror")},
            "road_plane": [float(v) for v in rec["plane"]],
            "left_pt": [float(v) for v in rec["left_pt"]] if rec["found"] else None,
            "right_pt": [float(v) for v in rec["right_pt"]] if rec["found"] else None}
    if res.get("f2f_record") is not None:
        f2 = res["f2f_record"]
        line["fence"] = {"ok": bool(f2["ok"]), "counts": [int(c) for c in f2["counts"]], "plane_left": [float(v) for v in f2["plane_left"]],
                         "plane_right": [float(v) for v in f2["plane_right"]]}
    if times is not None:
        line["times_s"] = {k: float(times[k]) for k in TIME_KEYS if k in times}
    if extra:
        line.update(extra)
    with open(path, "a") as f:
        f.write(json.dumps(line) + "\n")
    return path


# ------------------------------------------------------------------------------------------------ focal-length sweep
def write_sweep_data(f_directory: str, all_data, n_frames: int):
    """semantic_depth.py:907-936: rows (real, rw, f2f, |real-rw|, |real-f2f|) + a last row holding the two MAEs in columns
    3 and 4, ``fmt='%1.4f'``.  Returns (mae_rw, mae_f2f)."""
    all_data_array = np.asarray(all_data)
    mae_rw = np.sum(all_data_array[:, 3]) / n_frames
    mae_f2f = np.sum(all_data_array[:, 4]) / n_frames
    mae_for_file = np.zeros((1, 5))
    mae_for_file[:, 3] = mae_rw
    mae_for_file[:, 4] = mae_f2f
    np.savetxt("{}/data.txt".format(f_directory), np.concatenate((all_data_array, mae_for_file)), fmt="%1.4f")
    return mae_rw, mae_f2f


def focal_sweep(process, input_frames: dict, frame_depther, focal_lengths=(380, 580), results_directory: str = "results"):
    """the ``args.f is None`` branch of main(), semantic_depth.py:854-944.
    ``process(name) -> (dist_rw, dist_f2f)`` runs the pipeline on one frame (FrameProcessor.process_frame);
    ``input_frames``: name -> ground-truth width at the measuring depth (:837); ``frame_depther.f`` is reassigned per trial
    (:859).  Writes ``<results>/<f>/data.txt`` and ``<results>/best_focal_lengths.txt``; returns the summary dict."""
    best = dict(rw=(-1, None), f2f=(-1, None), overall=(-1, None))
    per_f = {}
    for f in focal_lengths:
        frame_depther.f = f
        f_directory = os.path.join(results_directory, str(f))
        os.makedirs(f_directory, exist_ok=True)
        all_data = []
        for name, real_distance in sorted(input_frames.items()):
            dist_rw, dist_f2f = process(name)
            all_data.append([real_distance, dist_rw, dist_f2f, abs(real_distance - dist_rw), abs(real_distance - dist_f2f)])
        mae_rw, mae_f2f = write_sweep_data(f_directory, all_data, len(input_frames))
        mae_overall = mae_rw + mae_f2f
        per_f[f] = dict(mae_rw=float(mae_rw), mae_f2f=float(mae_f2f), rows=all_data)
        for key, mae in (("rw", mae_rw), ("f2f", mae_f2f), ("overall", mae_overall)):
            if best[key][0] == -1 or mae < best[key][0]:
                best[key] = (mae, f)
    with open("{}/best_focal_lengths.txt".format(results_directory), "w") as fh:
        fh.write("Best f road's width: {}\n".format(best["rw"][1]))
        fh.write("Best f fence2fence:  {}\n".format(best["f2f"][1]))
        fh.write("Best f overall:      {}\n".format(best["overall"][1]))
    return dict(best_f_rw=best["rw"][1], best_f_f2f=best["f2f"][1], best_f_overall=best["overall"][1], per_f=per_f)
