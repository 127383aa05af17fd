"""CPU: the frame reader that replaces cv2.imread (semantic_depth.py:105): PNG decode through zlib +
libsemdepth's host-side scanline reconstruction, against PIL's decoder and against files written with every PNG filter type."""
import os
import struct
import zlib

import numpy as np
import pytest

import __graft_entry__ as graft
from semantic_depth_amd import frame_io, outputs


@pytest.fixture(scope="module", autouse=True)
def built():
    graft.build()


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _filter_rows(img, ftype):
    """PNG encoder side of filter types 0..4 (spec §9.2), written independently of the decoder under test"""
    h, w, ch = img.shape
    rows = img.reshape(h, w * ch).astype(np.int32)
    out = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        cur = rows[y]
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        upleft = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        ft = ftype if ftype < 5 else y % 5
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        out += bytes([ft]) + (f & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def _png(img, ctype, ftype, split_idat=False):
    h, w, ch = img.shape
    z = zlib.compress(_filter_rows(img, ftype), 6)
    idat = _chunk(b"IDAT", z) if not split_idat else _chunk(b"IDAT", z[:len(z) // 2]) + _chunk(b"IDAT", z[len(z) // 2:])
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + _chunk(b"tEXt", b"k\0v") + idat + _chunk(b"IEND", b"")


@pytest.mark.parametrize("ftype", [0, 1, 2, 3, 4, 5])          # 5 = a different filter on every row
def test_every_filter_type_rgb_and_rgba(ftype):
    rng = np.random.default_rng(ftype)
    rgb = rng.integers(0, 256, (19, 23, 3), dtype=np.uint8)
    assert np.array_equal(frame_io.decode_png(_png(rgb, 2, ftype, split_idat=ftype == 5)), rgb[..., ::-1])
    rgba = rng.integers(0, 256, (7, 11, 4), dtype=np.uint8)
    assert np.array_equal(frame_io.decode_png(_png(rgba, 6, ftype)), rgba[..., 2::-1])
    gray = rng.integers(0, 256, (9, 5, 1), dtype=np.uint8)
    assert np.array_equal(frame_io.decode_png(_png(gray, 0, ftype)), np.repeat(gray, 3, axis=2))


def test_against_pil_and_round_trip_with_the_writer(tmp_path):
    rng = np.random.default_rng(3)
    # a smooth image (what real encoders pick Paeth / Sub / Up for) + noise
    yy, xx = np.mgrid[0:120, 0:200]
    img = np.stack([(yy * 2 + xx) % 256, (xx * 3) % 256, (yy + 2 * xx) % 256], -1).astype(np.uint8) ^ rng.integers(0, 8, (120, 200, 3), dtype=np.uint8)
    p = str(tmp_path / "frame.png")
    outputs.write_png(p, img)                                  # BGR in, like cv2.imwrite
    assert np.array_equal(frame_io.imread(p), img)             # BGR out, like cv2.imread
    PIL = pytest.importorskip("PIL.Image")
    q = str(tmp_path / "pil.png")
    PIL.fromarray(img[..., ::-1]).save(q, optimize=True)       # PIL chooses its own (adaptive) filters
    assert np.array_equal(frame_io.imread(q), img)
    assert np.array_equal(np.asarray(PIL.open(p).convert("RGB")), img[..., ::-1])
    pal = PIL.fromarray(img[..., ::-1]).quantize(200)          # > 16 colours: an 8-bit palette image
    pal.save(str(tmp_path / "pal.png"))
    assert np.array_equal(frame_io.imread(str(tmp_path / "pal.png")), np.asarray(pal.convert("RGB"))[..., ::-1])


def test_rejects_what_it_does_not_decode(tmp_path):
    with pytest.raises(ValueError):
        frame_io.decode_png(b"\xff\xd8\xff\xe0 not a png")
    bad = bytearray(_png(np.zeros((4, 4, 3), np.uint8), 2, 0))
    bad[8 + 8 + 8] = 16                                        # bit depth 16
    with pytest.raises(ValueError, match="unsupported"):
        frame_io.decode_png(bytes(bad))


def test_frame_feeder_batches_in_order(tmp_path):
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (7, 16, 24, 3), dtype=np.uint8)
    paths = []
    for i, f in enumerate(frames):
        paths.append(outputs.write_png(str(tmp_path / f"stuttgart_{i:06d}.png"), f))
    got, los = [], []
    for dev, lo in frame_io.FrameFeeder(sorted(paths), batch=3, device="cpu", workers=4):
        got.append(dev.numpy().copy()); los.append(lo)
    assert los == [0, 3, 6] and [g.shape[0] for g in got] == [3, 3, 1]
    assert np.array_equal(np.concatenate(got), frames)


@pytest.mark.gpu
def test_frame_feeder_uploads_through_pinned_buffers(tmp_path):
    """the same feeder onto the GPU (pinned staging, asynchronous upload, double buffering) followed by the GPU cubic resize"""
    import torch
    from semantic_depth_amd.engine import Engine
    from oracle import resize as oresize
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, (5, 128, 192, 3), dtype=np.uint8)
    paths = [outputs.write_png(str(tmp_path / f"f{i:03d}.png"), f) for i, f in enumerate(frames)]
    eng = Engine(64, 128, 2, "resnet50")
    got = []
    for dev_frames, lo in frame_io.FrameFeeder(paths, batch=2, device="cuda", workers=3):
        assert dev_frames.is_cuda and dev_frames.dtype == torch.uint8
        got.append(eng.resize_cubic(dev_frames).cpu().numpy())
    got = np.concatenate(got)
    assert got.shape == (5, 64, 128, 3)
    for i in range(5):
        assert np.array_equal(got[i], oresize.resize_cubic_u8(frames[i], 64, 128))


def test_batch_reader_reports_the_files_it_could_not_decode(tmp_path):
    """sd_decode_files_bgr (the native batch reader under FrameFeeder): a missing file and a frame of another shape are reported per
    file; the frames around them are still decoded."""
    import ctypes as C
    from semantic_depth_amd import _lib as L
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (3, 10, 12, 3), dtype=np.uint8)
    paths = [outputs.write_png(str(tmp_path / f"a{i}.png"), f) for i, f in enumerate(frames)]
    other = outputs.write_png(str(tmp_path / "other_shape.png"), rng.integers(0, 256, (8, 12, 3), dtype=np.uint8))
    lst = [paths[0], str(tmp_path / "missing.png"), paths[1], other, paths[2]]
    out = np.zeros((5, 10, 12, 3), np.uint8)
    status = (C.c_int * 5)()
    arr = (C.c_char_p * 5)(*[p.encode() for p in lst])
    st = L.load().sd_decode_files_bgr(arr, 5, 10, 12, out.ctypes.data_as(C.c_void_p), 10 * 12 * 3, 3, status)
    assert st == -1 and list(status) == [0, -4, 0, -1, 0]                  # SD_ERR_NOTFOUND = -4, SD_ERR_INVALID = -1
    assert np.array_equal(out[[0, 2, 4]], frames)
    with pytest.raises(ValueError, match="could not be read"):
        list(frame_io.FrameFeeder(lst, batch=5, device="cpu", workers=2))
    # size query of the single-file entry point, and a truncated stream
    buf = open(paths[0], "rb").read()
    assert frame_io.png_size(buf) == (10, 12)
    with pytest.raises(ValueError):
        frame_io.decode_png(buf[:len(buf) // 2])


# ------------------------------------------------------------------------------------------------ JPEG (the reference's own frames)
def _jpeg_golden():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_golden.json")))


def test_jpeg_reader_against_the_committed_golden_vectors():
    """frames decoded by Pillow / libjpeg-turbo at libjpeg's default settings (= cv2.imread, semantic_depth.py:105) in the build
    container (tests/golden/make_jpeg_golden.py): baseline, optimised-Huffman and progressive files, 4:4:4 / 4:2:2 / 4:2:0, gray,
    odd sizes -- bit for bit"""
    import base64
    import hashlib
    g = _jpeg_golden()
    assert len(g["files"]) >= 6
    for f in g["files"]:
        buf = base64.b64decode(f["jpeg_base64"])
        got = frame_io.decode_image(buf)
        want = np.asarray(f["bgr"], np.uint8).reshape(f["height"], f["width"], 3)
        assert got.shape == want.shape and np.array_equal(got, want), f["name"]
        assert hashlib.sha256(got.tobytes()).hexdigest() == f["bgr_sha256"]
        assert frame_io.image_size(buf) == (f["height"], f["width"])


def test_jpeg_reader_on_the_references_example_frame():
    """assets/images/test_munich/test_3.jpg (progressive, 4032 x 3024): the frame the reference's README runs on.  Runs where the
    reference tree exists (the build container); the expected digest was made there with Pillow."""
    import hashlib
    ref = _jpeg_golden()["reference_frame"]
    path = os.path.join("/root/reference", ref["path"])
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    buf = open(path, "rb").read()
    assert hashlib.sha256(buf).hexdigest() == ref["file_sha256"]
    got = frame_io.imread(path)
    assert list(got.shape) == ref["shape"]
    for k, v in ref["probe_pixels_bgr"].items():
        y, x = map(int, k.split(","))
        assert got[y, x].tolist() == v, k
    assert hashlib.sha256(got.tobytes()).hexdigest() == ref["bgr_sha256"]


def test_jpeg_matrix_against_pillow_and_exif_orientation(tmp_path):
    """every combination of size x subsampling x quality x entropy mode Pillow writes, decoded by both; the EXIF orientations 1..8 as
    cv2.imread applies them (= ImageOps.exif_transpose); restart intervals; what the reader refuses"""
    import io
    PILImage = pytest.importorskip("PIL.Image")
    from PIL import ImageOps
    rng = np.random.default_rng(5)

    def img(h, w):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(yy * 3 + xx * 2) % 256, (xx * 5 + yy) % 256, (yy * yy // 7 + xx) % 256], -1).astype(np.uint8)
        return a ^ rng.integers(0, 32, a.shape, dtype=np.uint8)

    n = 0
    for (h, w) in [(16, 16), (17, 23), (64, 48), (8, 9), (57, 130)]:
        for ss in (0, 1, 2):
            for q in (35, 90, 100):
                for kw in ({}, {"optimize": True}, {"progressive": True}):
                    b = io.BytesIO()
                    try:
                        PILImage.fromarray(img(h, w)).save(b, "JPEG", quality=q, subsampling=ss, **kw)
                    except OSError:
                        continue            # (Pillow's encoder refuses some tiny optimised files on an in-memory stream)
                    want = np.asarray(PILImage.open(io.BytesIO(b.getvalue())).convert("RGB"))[..., ::-1]
                    assert np.array_equal(frame_io.decode_image(b.getvalue()), want), (h, w, ss, q, kw)
                    n += 1
    assert n > 100
    # restart intervals (DRI / RSTn)
    for ss in (0, 2):
        b = io.BytesIO()
        PILImage.fromarray(img(50, 70)).save(b, "JPEG", quality=80, subsampling=ss, restart_marker_blocks=3)
        assert b"\xff\xdd" in b.getvalue()
        want = np.asarray(PILImage.open(io.BytesIO(b.getvalue())).convert("RGB"))[..., ::-1]
        assert np.array_equal(frame_io.decode_image(b.getvalue()), want)
    # EXIF orientation: cv2.imread rotates / flips, Pillow does so on request
    base = PILImage.fromarray(img(24, 40))
    for o in range(1, 9):
        ex = PILImage.Exif()
        ex[0x0112] = o
        p = str(tmp_path / f"o{o}.jpg")
        base.save(p, "JPEG", quality=90, subsampling=0, exif=ex)
        want = np.asarray(ImageOps.exif_transpose(PILImage.open(p)).convert("RGB"))[..., ::-1]
        got = frame_io.imread(p)
        assert got.shape == want.shape and np.array_equal(got, want), o
        assert frame_io.image_size(open(p, "rb").read()) == want.shape[:2]
    # a feeder batch of JPEG frames
    paths = []
    for i in range(3):
        p = str(tmp_path / f"frame{i}.jpg")
        PILImage.fromarray(img(32, 48)).save(p, "JPEG", quality=85)
        paths.append(p)
    got = [d.numpy().copy() for d, _ in frame_io.FrameFeeder(paths, batch=2, device="cpu", workers=2)]
    for i, p in enumerate(paths):
        assert np.array_equal(np.concatenate(got)[i], np.asarray(PILImage.open(p).convert("RGB"))[..., ::-1])
    # refused: CMYK, truncated
    b = io.BytesIO()
    PILImage.fromarray(img(16, 16)).convert("CMYK").save(b, "JPEG")
    with pytest.raises(ValueError):
        frame_io.decode_image(b.getvalue())
    b = io.BytesIO()
    PILImage.fromarray(img(32, 32)).save(b, "JPEG")
    with pytest.raises(ValueError):
        frame_io.decode_image(b.getvalue()[:200])


def test_readers_survive_mutated_files():
    """truncated, bit-flipped, 0xFF-flooded and header-mutated JPEG / PNG files: the readers either decode to the announced shape or raise
    ValueError -- never crash, never write past the buffer (guard bytes), never size an allocation from a header the caller's buffer does
    not cover (scripts/fuzz_decoders.cpp is the same loop under ASan + UBSan: 42 000 mutations, no finding)"""
    import ctypes as C
    import io
    PILImage = pytest.importorskip("PIL.Image")
    from semantic_depth_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:45, 0:61]
    a = (np.stack([(yy * 3 + xx) % 256, (xx * 5) % 256, (yy * xx) % 256], -1).astype(np.uint8)) ^ rng.integers(0, 32, (45, 61, 3), dtype=np.uint8)
    seeds = []
    for kw in ({"subsampling": 2}, {"subsampling": 0, "progressive": True}, {"subsampling": 1, "restart_marker_blocks": 2}):
        b = io.BytesIO()
        PILImage.fromarray(a).save(b, "JPEG", quality=80, **kw)
        seeds.append(b.getvalue())
    for mode in ("RGB", "L", "RGBA", "P"):
        b = io.BytesIO()
        PILImage.fromarray(a).convert(mode).save(b, "PNG")
        seeds.append(b.getvalue())
    GUARD = 64
    decoded = refused = 0
    for seed in seeds:
        for it in range(150):
            f = bytearray(seed)
            kind = it % 5
            if kind == 0:
                del f[int(rng.integers(0, len(f))):]
            elif kind == 1:
                for _ in range(int(rng.integers(1, 8))):
                    f[int(rng.integers(0, len(f)))] = int(rng.integers(0, 256))
            elif kind == 2:
                for _ in range(int(rng.integers(1, 48))):
                    f[int(rng.integers(0, len(f)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 3:
                p = int(rng.integers(0, len(f)))
                f[p:p + int(rng.integers(1, 48))] = b"\xff" * 8
            else:
                for _ in range(int(rng.integers(1, 6))):
                    f[int(rng.integers(0, min(len(f), 640)))] = int(rng.integers(0, 256))
            f = bytes(f)
            h, w = C.c_int(0), C.c_int(0)
            if lib.sd_image_decode_bgr(f, len(f), None, 0, C.byref(h), C.byref(w)) != L.SD_OK:
                refused += 1
                continue
            assert h.value > 0 and w.value > 0            # (a PNG header may announce up to 2^31 - 1 per side; JPEG 65535)
            need = h.value * w.value * 3
            cap = min(need, 1 << 22)                       # a mutated header may announce gigapixels: the reader must refuse on capacity
            buf = np.full(cap + GUARD, 0xA5, np.uint8)
            st = lib.sd_image_decode_bgr(f, len(f), buf.ctypes.data_as(C.c_void_p), cap, None, None)
            assert (buf[cap:] == 0xA5).all()
            if st == L.SD_OK:
                assert need <= cap
                decoded += 1
            else:
                refused += 1
    assert decoded > 50 and refused > 50


def _seg(marker, payload):
    return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)


def _crafted_jpegs():
    """hand-built files for the structural holes random byte flips do not reach"""
    out = {}
    # (1) DHT whose code-length counts are not a prefix code: 200 codes of length 1
    bits = [200] + [0] * 15
    out["dht_oversubscribed"] = b"\xff\xd8" + _seg(0xC4, [0x00] + bits + list(range(200)))
    bits = [0, 0, 0, 0, 0, 0, 0, 0, 255] + [0] * 7            # 255 codes of length 9 fit; 1 + 255 of lengths 1 and 9 do not
    out["dht_len9_ok_header_only"] = b"\xff\xd8" + _seg(0xC4, [0x00] + bits + list(range(255)))
    bits = [1, 0, 0, 0, 0, 0, 0, 0, 255] + [0] * 7
    out["dht_len1_plus_len9"] = b"\xff\xd8" + _seg(0xC4, [0x00] + bits + list(range(256)))
    dqt = _seg(0xDB, [0x00] + [1] * 64)
    dht_dc = _seg(0xC4, [0x00] + [1] + [0] * 15 + [0])       # one code "0" -> category 0
    dht_ac = _seg(0xC4, [0x10] + [1] + [0] * 15 + [0])       # one code "0" -> EOB

    def sof(m, h, w):
        return _seg(m, [8] + list(h.to_bytes(2, "big")) + list(w.to_bytes(2, "big")) + [1, 1, 0x11, 0])
    sos = _seg(0xDA, [1, 1, 0x00, 0, 63, 0]) + b"\x00" * 4
    # (2) a second frame header of another shape after the planes exist (8x8 gray, then 64x1: same H*W*3)
    out["two_sof_reshaped"] = b"\xff\xd8" + dqt + dht_dc + dht_ac + sof(0xC0, 8, 8) + sos + sof(0xC0, 1, 64) + sos + b"\xff\xd9"
    # (3) baseline scan first, then a progressive frame header: prog_dc would run on an empty coefficient vector
    sos_p = _seg(0xDA, [1, 1, 0x00, 0, 0, 0]) + b"\x00" * 4
    out["sof0_then_sof2"] = b"\xff\xd8" + dqt + dht_dc + dht_ac + sof(0xC0, 8, 8) + sos + sof(0xC2, 8, 8) + sos_p + b"\xff\xd9"
    # (4) progressive file without a DQT: prog_finish would dequantise with an undefined table
    out["progressive_without_dqt"] = b"\xff\xd8" + dht_dc + dht_ac + sof(0xC2, 8, 8) + sos_p + b"\xff\xd9"
    # control: the same skeleton, well-formed (gray 8x8, all-zero coefficients -> level 128)
    out["control_ok"] = b"\xff\xd8" + dqt + dht_dc + dht_ac + sof(0xC0, 8, 8) + sos + b"\xff\xd9"
    return out


def test_jpeg_reader_refuses_structurally_hostile_files():
    """over-subscribed Huffman tables (libjpeg: JERR_BAD_HUFF_TABLE), a second frame header, a progressive file without quantisation
    tables: SD_ERR_INVALID from the size query and from the decode, nothing written past the buffer.  The same files run under
    ASan + UBSan in scripts/fuzz_decoders.cpp (structure-aware mutations)"""
    import ctypes as C
    from semantic_depth_amd import _lib as L
    lib = L.load()
    files = _crafted_jpegs()
    for name, f in files.items():
        h, w = C.c_int(0), C.c_int(0)
        st_q = lib.sd_image_decode_bgr(f, len(f), None, 0, C.byref(h), C.byref(w))
        cap = 64 * 64 * 3
        buf = np.full(cap + 64, 0xA5, np.uint8)
        st_d = lib.sd_image_decode_bgr(f, len(f), buf.ctypes.data_as(C.c_void_p), cap, C.byref(h), C.byref(w))
        assert (buf[cap:] == 0xA5).all(), name
        if name == "control_ok":
            assert st_q == L.SD_OK and st_d == L.SD_OK and (h.value, w.value) == (8, 8)
            assert (buf[:8 * 8 * 3] == 128).all()
        elif name == "dht_len9_ok_header_only":
            assert st_q != L.SD_OK and st_d != L.SD_OK     # a valid table, but no frame header follows
        elif name == "progressive_without_dqt":
            assert st_d != L.SD_OK, name                   # (the size query only reads the header)
        else:
            assert st_d != L.SD_OK, name
            if name.startswith("dht"):
                assert st_q != L.SD_OK, name


def test_default_decode_workers_divides_by_the_ranks_of_the_node(monkeypatch):
    """launchers that do not set LOCAL_WORLD_SIZE (mpirun, srun, RANK / WORLD_SIZE by hand) must not give every rank the whole
    affinity mask; cgroup v1 quotas count like v2 ones"""
    from semantic_depth_amd import frame_io
    for k in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(frame_io, "_cgroup_cpu_quota", lambda: None)
    monkeypatch.setattr("os.sched_getaffinity", lambda pid: set(range(64)), raising=False)
    assert frame_io.default_decode_workers() == 64
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert frame_io.default_decode_workers() == 8
    monkeypatch.setenv("SLURM_NTASKS_PER_NODE", "4(x2)")
    assert frame_io.default_decode_workers() == 16
    monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_SIZE", "2")
    assert frame_io.default_decode_workers() == 32
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "64")
    assert frame_io.default_decode_workers() == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    monkeypatch.setattr(frame_io, "_cgroup_cpu_quota", lambda: 16)
    assert frame_io.default_decode_workers() == 16
