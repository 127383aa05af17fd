"""TensorFlow-free readers for checkpoint V2 bundles and frozen GraphDefs.  No TensorFlow and no real
checkpoint exist in the build container, so the fixtures come from the writers of the same module (UNPINNED against TF)."""
import numpy as np
import pytest

from semantic_depth_amd import tf_import as T
from semantic_depth_amd import weights as W


def test_tensor_bundle_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {f"model/encoder/Conv_{i}/weights": rng.standard_normal((3, 3, 4, 5)).astype(np.float32) for i in range(23)}
    tensors["model/encoder/Conv/biases"] = rng.standard_normal(7).astype(np.float32)
    tensors["global_step"] = np.array(123456, np.int64)
    tensors["scalar"] = np.float32(2.5)
    prefix = str(tmp_path / "model")
    T.write_tensor_bundle(prefix, tensors)
    got = T.read_tensor_bundle(prefix)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == np.asarray(v).dtype and np.array_equal(got[k], v), k
    with pytest.raises(ValueError):
        open(prefix + ".index", "ab").write(b"x")        # a damaged footer is reported, not mis-parsed
        T.read_tensor_bundle(prefix)


def test_snappy_block_and_varints():
    raw = b"abcabcabcabc" + bytes(range(70)) + b"abcabc"
    # literal(3) "abc", copy(len 9, off 3), literal 70 (length >= 60 form), copy2(len 6, off 79)
    comp = T._put_varint(len(raw)) + bytes([2 << 2]) + b"abc" + bytes([(5 << 2) | 1 | (0 << 5), 3]) + bytes([60 << 2, 69]) + bytes(range(70)) + \
        bytes([((6 - 1) << 2) | 2]) + (79).to_bytes(2, "little")
    assert T._snappy(comp) == raw
    for v in (0, 1, 127, 128, 300, 2 ** 35 + 17):
        assert T._varint(T._put_varint(v), 0) == (v, len(T._put_varint(v)))


def test_frozen_graph_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    consts = {"conv1_1/filter": rng.standard_normal((3, 3, 3, 8)).astype(np.float32), "conv1_1/biases": rng.standard_normal(8).astype(np.float32)}
    p = str(tmp_path / "frozen.pb")
    T.write_frozen_graph(p, consts)
    got = T.read_frozen_graph(p)
    assert all(np.array_equal(got[k], v) for k, v in consts.items())


@pytest.mark.parametrize("encoder", ["vgg", "resnet50"])
def test_monodepth_checkpoint_to_slots(tmp_path, encoder):
    shapes = W.monodepth_weight_shapes(encoder)
    nm = T.monodepth_name_map(encoder)
    assert list(nm) == list(shapes)                                   # every slot mapped, same order
    assert nm["enc/conv1a/weights" if encoder == "vgg" else "enc/conv1/weights"] == "model/encoder/Conv/weights"
    assert len(set(nm.values())) == len(nm)
    if encoder == "resnet50":       # conv1 + 16 blocks x (3 convs + projection) = 65 encoder convs; the last one is Conv_64
        assert nm["enc/res5_3/proj/biases"] == "model/encoder/Conv_64/biases"
        assert nm["enc/res2_1/conv2/weights"] == "model/encoder/Conv_2/weights"
    assert nm["dec/upconv7/weights" if encoder == "vgg" else "dec/upconv6/weights"] == "model/decoder/Conv/weights"
    rng = np.random.default_rng(2)
    w = {slot: rng.standard_normal(shape).astype(np.float32) for slot, shape in shapes.items()}
    prefix = str(tmp_path / "model_cityscapes")
    T.write_tensor_bundle(prefix, {nm[s]: a for s, a in w.items()} | {"global_step": np.array(1, np.int64)})
    out = str(tmp_path / "mono.npz")
    T.main(["--monodepth", prefix, "--encoder", encoder, "--out", out])
    z = np.load(out)
    assert set(z.files) == set(shapes) and all(np.array_equal(z[s], w[s]) for s in shapes)


def test_fcn8s_frozen_to_slots(tmp_path):
    shapes = W.fcn8s_weight_shapes()
    nm = T.fcn8s_name_map()
    assert nm["vgg/fc6/filter"] == "fc6/weights" and nm["vgg/conv3_2/biases"] == "conv3_2/biases"
    assert nm["dec/score4/kernel"] == "conv2d_1/kernel" and nm["dec/deconv3/bias"] == "conv2d_transpose_2/bias"
    rng = np.random.default_rng(3)
    shapes = {s: shp for s, shp in shapes.items() if np.prod(shp) < 3e6}        # (fc6 alone is 411 MB: keep the fixture small)
    nm = {s: nm[s] for s in shapes}
    small = {s: rng.standard_normal(shape).astype(np.float32) for s, shape in shapes.items()}
    p = str(tmp_path / "frozen.pb")
    T.write_frozen_graph(p, {nm[s] + "": a for s, a in small.items()})
    got = T.convert(T.read_frozen_graph(p), nm, shapes)
    assert all(np.array_equal(got[s], small[s]) for s in shapes)
    with pytest.raises(ValueError):
        bad = dict(small); bad["vgg/conv1_1/filter"] = np.zeros((3, 3, 3, 63), np.float32)
        T.write_frozen_graph(p, {nm[s]: a for s, a in bad.items()})
        T.convert(T.read_frozen_graph(p), nm, shapes)


# ------------------------------------------------------------------------------------------------ public vectors / an independent writer
def test_crc32c_public_vectors():
    """RFC 3720 appendix B.4 (iSCSI CRC-32C examples) and the classic check value"""
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    # leveldb::crc32c::Mask: rotate right by 15, add 0xa282ead8
    c = T.crc32c(b"foo")
    assert T.crc_mask(c) == ((((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF) and T.crc_mask(c) != c
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == 0xE3069283          # incremental form


def _crc32c_bitwise(data):
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


def test_reader_against_a_table_assembled_from_the_format_description(tmp_path):
    """An index file put together here, byte by byte, from LevelDB's table_format.md / TensorFlow's tensor_bundle.proto — NOT
    with the module's writer: prefix-compressed keys with restart interval 2, one block snappy-compressed (type 1, literal
    elements only), real masked CRC-32C trailers from an independent bitwise CRC, a two-level index."""
    import struct
    vi = T._put_varint

    def msg(*fields):           # (number, wire type, value)
        out = b""
        for n, wt, v in fields:
            out += vi((n << 3) | wt)
            out += vi(v) if wt == 0 else vi(len(v)) + v
        return out

    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4) * 0.5
    b = np.array([7, -9], np.int32)
    data = a.tobytes() + b.tobytes()
    shape = lambda s: b"".join(msg((2, 2, msg((1, 0, d)))) for d in s)
    entries = [(b"", msg((1, 0, 1), (2, 0, 0))),                                                   # BundleHeaderProto
               (b"model/a/weights", msg((1, 0, 1), (2, 2, shape(a.shape)), (3, 0, 0), (4, 0, 0), (5, 0, a.nbytes))),
               (b"model/b/weights", msg((1, 0, 3), (2, 2, shape(b.shape)), (3, 0, 0), (4, 0, a.nbytes), (5, 0, b.nbytes)))]

    def block(items, interval):
        out, restarts, prev = b"", [], b""
        for i, (k, v) in enumerate(items):
            sh = 0
            if i % interval == 0:
                restarts.append(len(out))
            else:
                while sh < min(len(k), len(prev)) and k[sh] == prev[sh]:
                    sh += 1
            out += vi(sh) + vi(len(k) - sh) + vi(len(v)) + k[sh:] + v
            prev = k
        return out + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))

    def snappy_literals(raw):   # varint length, then literal elements of <= 60 bytes (tag = (len-1) << 2)
        out = vi(len(raw))
        for i in range(0, len(raw), 60):
            piece = raw[i:i + 60]
            out += bytes([(len(piece) - 1) << 2]) + piece
        return out

    f = b""
    handles = []
    for items, compress in ((entries[:2], False), (entries[2:], True)):
        raw = block(items, 2)
        body, ctype = (snappy_literals(raw), 1) if compress else (raw, 0)
        handles.append((items[-1][0], vi(len(f)) + vi(len(body))))
        f += body + bytes([ctype]) + struct.pack("<I", T.crc_mask(_crc32c_bitwise(body + bytes([ctype]))))
    meta = block([], 1)
    meta_h = vi(len(f)) + vi(len(meta))
    f += meta + b"\x00" + struct.pack("<I", T.crc_mask(_crc32c_bitwise(meta + b"\x00")))
    idx = block(handles, 1)
    idx_h = vi(len(f)) + vi(len(idx))
    f += idx + b"\x00" + struct.pack("<I", T.crc_mask(_crc32c_bitwise(idx + b"\x00")))
    footer = meta_h + idx_h
    f += footer + bytes(40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    prefix = str(tmp_path / "hand")
    open(prefix + ".index", "wb").write(f)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    got = T.read_tensor_bundle(prefix)
    assert list(got) == ["model/a/weights", "model/b/weights"]
    assert got["model/a/weights"].dtype == np.float32 and np.array_equal(got["model/a/weights"], a)
    assert got["model/b/weights"].dtype == np.int32 and np.array_equal(got["model/b/weights"], b)
    # a flipped payload bit is caught by the block checksum
    bad = bytearray(f)
    bad[5] ^= 0x10
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="crc"):
        T.read_tensor_bundle(prefix)


def test_two_shard_bundle_with_many_blocks_assembled_by_hand(tmp_path):
    """what a real multi-device checkpoint looks like -- 40 variables spread over TWO data shards
    (``.data-00000-of-00002`` / ``.data-00001-of-00002``, BundleHeaderProto.num_shards = 2, BundleEntryProto.shard_id / offset per
    entry), the index cut into NINE data blocks of at most 5 entries (restart interval 3, so every block has shared-prefix entries
    behind two restart points), alternate blocks snappy-compressed, a multi-entry index block -- assembled here from the format
    descriptions with an independent CRC, NOT with the module's writer."""
    import struct
    vi = T._put_varint

    def msg(*fields):           # (number, wire type, value): 0 varint, 2 length-delimited, 5 fixed32
        out = b""
        for n, wt, v in fields:
            out += vi((n << 3) | wt)
            out += vi(v) if wt == 0 else (v if wt == 5 else vi(len(v)) + v)
        return out

    rng = np.random.default_rng(11)
    names = sorted([f"model/encoder/Conv_{i}/weights" for i in range(20)] + [f"model/encoder/Conv_{i}/biases" for i in range(19)] + ["global_step"])
    tensors, shard_of, off_of = {}, {}, {}
    shards = [b"", b""]
    for j, n in enumerate(names):
        a = (np.array(987654321, np.int64) if n == "global_step" else
             rng.standard_normal((3, 3, 2, 1 + j % 5) if n.endswith("weights") else (1 + j % 7,)).astype(np.float32))
        sh = (j * 7 // 3) % 2                                   # an irregular assignment to the two shards
        tensors[n], shard_of[n], off_of[n] = a, sh, len(shards[sh])
        shards[sh] += a.tobytes() + b"\xee" * (j % 3)           # (padding between tensors: offsets, not order, locate a tensor)
    shape = lambda s: b"".join(msg((2, 2, msg((1, 0, d)))) for d in s)
    entries = [(b"", msg((1, 0, 2), (2, 0, 0), (3, 2, msg((1, 0, 1)))))]      # header: num_shards 2, little endian, version {producer 1}
    for n in names:
        a = tensors[n]
        entries.append((n.encode(), msg((1, 0, 9 if a.dtype == np.int64 else 1), (2, 2, shape(a.shape)), (3, 0, shard_of[n]), (4, 0, off_of[n]),
                                        (5, 0, a.nbytes), (6, 5, struct.pack("<I", 0)))))     # (+ fixed32 crc32c field, ignored by the reader)

    def block(items, interval):
        out, restarts, prev = b"", [], b""
        for i, (k, v) in enumerate(items):
            sh = 0
            if i % interval == 0:
                restarts.append(len(out))
            else:
                while sh < min(len(k), len(prev)) and k[sh] == prev[sh]:
                    sh += 1
            out += vi(sh) + vi(len(k) - sh) + vi(len(v)) + k[sh:] + v
            prev = k
        return out + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts)), len(restarts)

    def snappy_literals(raw):
        out = vi(len(raw))
        for i in range(0, len(raw), 60):
            piece = raw[i:i + 60]
            out += bytes([(len(piece) - 1) << 2]) + piece
        return out

    f, handles, nblocks, nrestarts = b"", [], 0, 0
    for i in range(0, len(entries), 5):
        items = entries[i:i + 5]
        raw, nr = block(items, 3)
        nrestarts += nr
        compress = (nblocks % 2) == 1
        body, ctype = (snappy_literals(raw), 1) if compress else (raw, 0)
        handles.append((items[-1][0], vi(len(f)) + vi(len(body))))
        f += body + bytes([ctype]) + struct.pack("<I", T.crc_mask(_crc32c_bitwise(body + bytes([ctype]))))
        nblocks += 1
    assert nblocks == 9 and nrestarts == 17
    meta, _ = block([], 1)
    meta_h = vi(len(f)) + vi(len(meta))
    f += meta + b"\x00" + struct.pack("<I", T.crc_mask(_crc32c_bitwise(meta + b"\x00")))
    idx, _ = block(handles, 1)
    idx_h = vi(len(f)) + vi(len(idx))
    f += idx + b"\x00" + struct.pack("<I", T.crc_mask(_crc32c_bitwise(idx + b"\x00")))
    footer = meta_h + idx_h
    f += footer + bytes(40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    prefix = str(tmp_path / "model_two_shards")
    open(prefix + ".index", "wb").write(f)
    for s in (0, 1):
        open(f"{prefix}.data-{s:05d}-of-00002", "wb").write(shards[s])
    got = T.read_tensor_bundle(prefix)
    assert list(got) == names                                  # table order = sorted keys
    for n in names:
        assert got[n].dtype == tensors[n].dtype and got[n].shape == tensors[n].shape and np.array_equal(got[n], tensors[n]), n
    assert len({shard_of[n] for n in names}) == 2
    # a missing shard file is an error, not a silent skip
    import os
    os.remove(f"{prefix}.data-00001-of-00002")
    with pytest.raises(FileNotFoundError):
        T.read_tensor_bundle(prefix)


# The CREATION order of the slim.conv2d layers in mrharicot/monodepth's monodepth_model.py, written out from the upstream source
# (build_resnet50 / build_vgg, resconv, conv_block, upconv = upsample_nn + conv, get_disp) -- slim names them Conv, Conv_1, ... per
# variable scope in exactly this order, which is all that identifies a variable in model_cityscapes / model_kitti (many layers share
# shapes, so a wrong map would pass every shape check).
#   resconv(x, n, stride): conv1 = conv(x, n, 1, 1); conv2 = conv(conv1, n, 3, stride); conv3 = conv(conv2, 4n, 1, 1, None);
#                          shortcut = conv(x, 4n, 1, stride, None)        -> four layers per block: conv1, conv2, conv3, proj
#   resblock(x, n, blocks): blocks - 1 x resconv(., n, 1), then resconv(., n, 2)
#   encoder scope: conv1 = conv(input, 64, 7, 2); resblock(pool1, 64, 3); resblock(., 128, 4); resblock(., 256, 6); resblock(., 512, 3)
#   decoder scope: upconv6, iconv6, upconv5, iconv5, upconv4, iconv4, disp4, upconv3, iconv3, disp3, upconv2, iconv2, disp2,
#                  upconv1, iconv1, disp1
_RESNET50_ENCODER_ORDER = (["enc/conv1"] +
                           [f"enc/res2_{b}/{l}" for b in (1, 2, 3) for l in ("conv1", "conv2", "conv3", "proj")] +
                           [f"enc/res3_{b}/{l}" for b in (1, 2, 3, 4) for l in ("conv1", "conv2", "conv3", "proj")] +
                           [f"enc/res4_{b}/{l}" for b in (1, 2, 3, 4, 5, 6) for l in ("conv1", "conv2", "conv3", "proj")] +
                           [f"enc/res5_{b}/{l}" for b in (1, 2, 3) for l in ("conv1", "conv2", "conv3", "proj")])
_RESNET50_DECODER_ORDER = ["dec/upconv6", "dec/iconv6", "dec/upconv5", "dec/iconv5", "dec/upconv4", "dec/iconv4", "dec/disp4", "dec/upconv3", "dec/iconv3",
                           "dec/disp3", "dec/upconv2", "dec/iconv2", "dec/disp2", "dec/upconv1", "dec/iconv1", "dec/disp1"]
#   build_vgg: conv_block(x, n, k) = conv(x, n, k, 1) then conv(., n, k, 2): conv1 .. conv7 -> a, b of each; decoder from upconv7 down
_VGG_ENCODER_ORDER = [f"enc/conv{i}{ab}" for i in range(1, 8) for ab in ("a", "b")]
_VGG_DECODER_ORDER = ["dec/upconv7", "dec/iconv7", "dec/upconv6", "dec/iconv6", "dec/upconv5", "dec/iconv5", "dec/upconv4", "dec/iconv4", "dec/disp4",
                      "dec/upconv3", "dec/iconv3", "dec/disp3", "dec/upconv2", "dec/iconv2", "dec/disp2", "dec/upconv1", "dec/iconv1", "dec/disp1"]
# spot values a reader of the upstream code can check by counting: the literal names of a few layers
_RESNET50_LITERALS = {"enc/conv1": "Conv", "enc/res2_1/conv1": "Conv_1", "enc/res2_1/proj": "Conv_4", "enc/res2_3/conv2": "Conv_10", "enc/res3_1/conv1": "Conv_13",
                      "enc/res3_4/proj": "Conv_28", "enc/res4_1/conv1": "Conv_29", "enc/res4_6/proj": "Conv_52", "enc/res5_1/conv1": "Conv_53", "enc/res5_3/proj": "Conv_64",
                      "dec/upconv6": "Conv", "dec/iconv4": "Conv_5", "dec/disp4": "Conv_6", "dec/upconv3": "Conv_7", "dec/disp2": "Conv_12", "dec/iconv1": "Conv_14",
                      "dec/disp1": "Conv_15"}


@pytest.mark.parametrize("encoder", ["resnet50", "vgg"])
def test_slim_creation_order_against_the_upstream_construction_order(encoder):
    nm = T.monodepth_name_map(encoder)
    enc, dec = (_RESNET50_ENCODER_ORDER, _RESNET50_DECODER_ORDER) if encoder == "resnet50" else (_VGG_ENCODER_ORDER, _VGG_DECODER_ORDER)
    assert len(enc) == (65 if encoder == "resnet50" else 14) and len(dec) == (16 if encoder == "resnet50" else 18)
    want = {}
    for scope, order in (("model/encoder", enc), ("model/decoder", dec)):
        for i, layer in enumerate(order):
            for leaf in ("weights", "biases"):
                want[f"{layer}/{leaf}"] = f"{scope}/Conv{'' if i == 0 else '_' + str(i)}/{leaf}"
    assert nm == want                                          # every slot, both scopes, no extras
    if encoder == "resnet50":
        for layer, conv in _RESNET50_LITERALS.items():
            scope = "model/encoder" if layer.startswith("enc/") else "model/decoder"
            assert nm[layer + "/weights"] == f"{scope}/{conv}/weights", layer
    # the slot table itself lists the layers in that order (the engine's plan and the importer agree on what 'res3_2/proj' is)
    slots = [s[:-len("/weights")] for s in W.monodepth_weight_shapes(encoder) if s.endswith("/weights")]
    assert slots == enc + dec
