#!/usr/bin/env python3
"""Golden vectors for the JPEG reader (semantic_depth.py:105 cv2.imread), made in the build container with Pillow
(libjpeg-turbo at libjpeg's default decode settings: ISLOW inverse DCT, fancy chroma upsampling -- what cv2.imread uses too):

  * the reference's own example frame /root/reference/assets/images/test_munich/test_3.jpg (progressive, 4032 x 3024): sha256 of the
    decoded BGR array + a few probe pixels.  The 1.5 MB file itself is not copied: the test that uses this entry runs where
    /root/reference exists.
  * small synthetic JPEGs written by Pillow (baseline / optimised Huffman / progressive, 4:4:4 / 4:2:2 / 4:2:0, gray, odd sizes,
    restart-free) as base64 with the decoded BGR arrays, so the test does not need Pillow.
    python tests/golden/make_jpeg_golden.py    ->  tests/golden/jpeg_golden.json
"""
import base64, hashlib, io, json, os
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/assets/images/test_munich/test_3.jpg"


def decode(buf):
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(buf)).convert("RGB"))[..., ::-1])


def main():
    rng = np.random.default_rng(7)
    out = {"made_with": "Pillow (libjpeg-turbo), default decode settings", "files": [], "reference_frame": None}
    for i, (h, w, ss, q, kw, gray) in enumerate([(17, 23, 0, 75, {}, False), (33, 31, 1, 90, {"optimize": True}, False), (40, 56, 2, 60, {}, False),
                                                 (24, 19, 2, 85, {"progressive": True}, False), (16, 40, 0, 95, {"progressive": True, "optimize": True}, False),
                                                 (21, 13, 0, 80, {}, True)]):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(yy * 5 + xx * 2) % 256, (xx * 7 + yy) % 256, (yy * yy // 5 + xx * 3) % 256], -1).astype(np.uint8) ^ rng.integers(0, 24, (h, w, 3), dtype=np.uint8)
        img = Image.fromarray(a[..., 0] if gray else a)
        b = io.BytesIO()
        img.save(b, "JPEG", quality=q, **({} if gray else {"subsampling": ss}), **kw)
        buf = b.getvalue()
        bgr = decode(buf)
        out["files"].append({"name": f"synthetic_{i}", "height": h, "width": w, "subsampling": None if gray else ["4:4:4", "4:2:2", "4:2:0"][ss], "quality": q,
                             "options": kw, "gray": gray, "jpeg_base64": base64.b64encode(buf).decode(), "bgr_sha256": hashlib.sha256(bgr.tobytes()).hexdigest(),
                             "bgr": bgr.reshape(-1).tolist()})
    if os.path.exists(REF):
        buf = open(REF, "rb").read()
        bgr = decode(buf)
        probes = [(0, 0), (1511, 2017), (3023, 4031), (100, 4000), (2999, 7)]
        out["reference_frame"] = {"path": "assets/images/test_munich/test_3.jpg", "file_sha256": hashlib.sha256(buf).hexdigest(), "shape": list(bgr.shape),
                                  "progressive": True, "bgr_sha256": hashlib.sha256(bgr.tobytes()).hexdigest(),
                                  "probe_pixels_bgr": {f"{y},{x}": bgr[y, x].tolist() for y, x in probes}}
    with open(os.path.join(HERE, "jpeg_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "jpeg_golden.json"), os.path.getsize(os.path.join(HERE, "jpeg_golden.json")), "bytes")


if __name__ == "__main__":
    main()
