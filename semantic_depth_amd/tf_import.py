"""Real-weight import without TensorFlow: readers for the three containers the reference loads its
weights from, and the mapping of their variable names onto this library's weight slots.

  * TF checkpoint V2 / TensorBundle (``prefix.index`` + ``prefix.data-0000i-of-0000n``): monodepth ``model_cityscapes`` /
    ``model_kitti`` (semantic_depth.py:627-653) and the FCN-8s training checkpoints (semantic_depth.py:498-541, non-frozen);
    also the ``variables/variables`` bundle of the Udacity ``vgg`` SavedModel (fcn8s/fcn.py:85).
  * frozen GraphDef ``.pb`` (semantic_depth.py:516-541, ``use_frozen``): Const nodes.

The index file is a LevelDB-format table (blocks of prefix-compressed keys, 48-byte footer, magic 0xdb4775248b80fb57)
whose values are BundleEntryProto messages; both that and GraphDef are decoded with a minimal protobuf wire reader.
UNPINNED: no TensorFlow and no real checkpoint exist in the build container; the readers are exercised against files
produced by the writers in this module (tests/test_tf_import.py), which follow the published formats.

CLI:  python -m semantic_depth_amd.tf_import --monodepth models/monodepth/model_cityscapes --encoder resnet50 --out mono.npz
      python -m semantic_depth_amd.tf_import --fcn8s-frozen models/sem_seg/.../frozen.pb --out fcn8s.npz
"""
from __future__ import annotations

import argparse
import glob
import os
import struct
from collections import OrderedDict

import numpy as np

from . import weights as W

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 6: np.int8, 9: np.int64, 10: np.bool_, 19: np.float16}
_DT_OF = {np.dtype(v): k for k, v in _DTYPES.items()}


# --------------------------------------------------------------------------------------------- wire helpers
def _varint(b: bytes, p: int):
    r = s = 0
    while True:
        c = b[p]; p += 1
        r |= (c & 0x7F) << s
        if c < 0x80:
            return r, p
        s += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        c = v & 0x7F
        v >>= 7
        out.append(c | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _fields(b: bytes):
    """protobuf wire format -> list of (field number, wire type, value)"""
    p, out = 0, []
    while p < len(b):
        key, p = _varint(b, p)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _varint(b, p)
        elif wt == 1:
            v = b[p:p + 8]; p += 8
        elif wt == 2:
            n, p = _varint(b, p)
            v = b[p:p + n]; p += n
        elif wt == 5:
            v = b[p:p + 4]; p += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.append((f, wt, v))
    return out


def _msg(fields) -> bytes:
    """[(field, wire type, value)] -> bytes (value: int for varint, bytes otherwise)"""
    out = bytearray()
    for f, wt, v in fields:
        out += _put_varint((f << 3) | wt)
        if wt == 0:
            out += _put_varint(v)
        elif wt == 2:
            out += _put_varint(len(v)) + v
        else:
            out += v
    return bytes(out)


def _snappy(b: bytes) -> bytes:
    n, p = _varint(b, 0)
    out = bytearray()
    while p < len(b):
        tag = b[p]; p += 1
        t = tag & 3
        if t == 0:
            ln = tag >> 2
            if ln >= 60:
                k = ln - 59
                ln = int.from_bytes(b[p:p + k], "little"); p += k
            ln += 1
            out += b[p:p + ln]; p += ln
            continue
        if t == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | b[p]; p += 1
        elif t == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(b[p:p + 2], "little"); p += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(b[p:p + 4], "little"); p += 4
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == n, "snappy: length mismatch"
    return bytes(out)


# --------------------------------------------------------------------------------------------- CRC-32C (Castagnoli), LevelDB masking
_CRC_TABLE = None


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C as LevelDB / TensorFlow compute it (reflected polynomial 0x82F63B78; RFC 3720 B.4 has the test vectors)"""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc_mask(crc: int) -> int:
    """leveldb::crc32c::Mask — the form stored in block trailers and in BundleEntryProto.crc32c"""
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------------- LevelDB table
def _read_block(buf: bytes, off: int, size: int) -> bytes:
    data, ctype = buf[off:off + size], buf[off + size]
    stored = struct.unpack_from("<I", buf, off + size + 1)[0]
    if stored != 0 and stored != crc_mask(crc32c(buf[off:off + size + 1])):      # trailer: type byte + masked crc32c(block | type)
        raise ValueError(f"table block at {off}: crc mismatch (file damaged)")      # (0 = written without a checksum)
    if ctype == 1:
        data = _snappy(data)
    elif ctype != 0:
        raise ValueError(f"unknown block compression {ctype}")
    return data


def _block_entries(block: bytes):
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    p, key = 0, b""
    while p < end:
        shared, p = _varint(block, p)
        non_shared, p = _varint(block, p)
        vlen, p = _varint(block, p)
        key = key[:shared] + block[p:p + non_shared]; p += non_shared
        yield key, block[p:p + vlen]
        p += vlen


def _table_items(path: str):
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
        raise ValueError(f"{path}: not a TensorBundle index (bad table magic)")
    footer = buf[-48:]
    _, p = _varint(footer, 0); _, p = _varint(footer, p)            # metaindex handle
    ioff, p = _varint(footer, p); isz, p = _varint(footer, p)       # index handle
    for _, handle in _block_entries(_read_block(buf, ioff, isz)):
        boff, q = _varint(handle, 0); bsz, q = _varint(handle, q)
        yield from _block_entries(_read_block(buf, boff, bsz))


def read_tensor_bundle(prefix: str) -> "OrderedDict[str, np.ndarray]":
    """all tensors of a TF checkpoint V2 (``prefix.index`` + data shards) by variable name"""
    shards: dict[int, bytes] = {}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    nshards = 1
    for key, val in _table_items(prefix + ".index"):
        f = _fields(val)
        if key == b"":                                              # BundleHeaderProto: num_shards = 1, endianness = 2
            nshards = next((v for n, _, v in f if n == 1), 1)
            if next((v for n, _, v in f if n == 2), 0) != 0:
                raise ValueError("big-endian bundle")
            continue
        dtype = shard = off = size = 0
        shape: list[int] = []
        for n, _, v in f:
            if n == 1: dtype = v
            elif n == 2: shape = [next((x for m, _, x in _fields(d) if m == 1), 0) for m2, _, d in _fields(v) if m2 == 2]
            elif n == 3: shard = v
            elif n == 4: off = v
            elif n == 5: size = v
        if dtype not in _DTYPES:
            continue                                                # strings etc. (e.g. saver bookkeeping)
        if shard not in shards:
            shards[shard] = open(f"{prefix}.data-{shard:05d}-of-{nshards:05d}", "rb").read()
        a = np.frombuffer(shards[shard], dtype=_DTYPES[dtype], count=int(np.prod(shape, dtype=np.int64)) if shape else 1, offset=off)
        assert a.nbytes == size, (key, a.nbytes, size)
        out[key.decode()] = a.reshape(shape).copy()
    return out


def write_tensor_bundle(prefix: str, tensors: dict, block_entries: int = 7) -> None:
    """single-shard, uncompressed TensorBundle writer (test fixture generator; same format as BundleWriter)"""
    data = bytearray()
    items = [(b"", _msg([(1, 0, 1), (2, 0, 0), (3, 2, _msg([(1, 0, 1)]))]))]      # header: 1 shard, little endian, version
    for name in sorted(tensors):
        a = np.asarray(tensors[name]).copy(order="C")            # (ascontiguousarray would turn a scalar into shape [1])
        shape = _msg([(2, 2, _msg([(1, 0, int(d))])) for d in a.shape])
        entry = [(1, 0, _DT_OF[a.dtype]), (2, 2, shape), (3, 0, 0), (4, 0, len(data)), (5, 0, a.nbytes)]
        data += a.tobytes()
        items.append((name.encode(), _msg(entry)))
    items.sort(key=lambda kv: kv[0])

    def block(entries, restart_every=4):
        out, restarts, prev = bytearray(), [], b""
        for i, (k, v) in enumerate(entries):
            shared = 0
            if i % restart_every == 0:
                restarts.append(len(out))
            else:
                while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
                    shared += 1
            out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
            prev = k
        for r in restarts:
            out += struct.pack("<I", r)
        out += struct.pack("<I", len(restarts))
        return bytes(out)

    f = bytearray()
    index = []
    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        b = block(chunk)
        index.append((chunk[-1][0], _put_varint(len(f)) + _put_varint(len(b))))
        f += b + b"\x00" + struct.pack("<I", crc_mask(crc32c(b + b"\x00")))      # trailer: type 0 + masked crc32c
    meta = block([])
    meta_h = _put_varint(len(f)) + _put_varint(len(meta))
    f += meta + b"\x00" + struct.pack("<I", crc_mask(crc32c(meta + b"\x00")))
    ib = block(index, restart_every=1)
    idx_h = _put_varint(len(f)) + _put_varint(len(ib))
    f += ib + b"\x00" + struct.pack("<I", crc_mask(crc32c(ib + b"\x00")))
    footer = meta_h + idx_h
    f += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    open(prefix + ".index", "wb").write(bytes(f))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))


# --------------------------------------------------------------------------------------------- frozen GraphDef
def _tensor_proto(b: bytes) -> np.ndarray | None:
    dtype, shape, content, floats = 0, [], None, []
    for n, wt, v in _fields(b):
        if n == 1: dtype = v
        elif n == 2: shape = [next((x for m, _, x in _fields(d) if m == 1), 0) for m2, _, d in _fields(v) if m2 == 2]
        elif n == 4: content = v
        elif n == 5: floats += list(np.frombuffer(v, "<f4")) if wt == 2 else [struct.unpack("<f", v)[0]]
    if dtype not in _DTYPES:
        return None
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    if content is not None:
        return np.frombuffer(content, _DTYPES[dtype], count).reshape(shape).copy()
    if floats:
        a = np.asarray(floats, np.float32)
        return (np.full(count, a[0], np.float32) if a.size == 1 else a).reshape(shape)
    return None


def read_frozen_graph(path: str) -> "OrderedDict[str, np.ndarray]":
    """Const nodes of a frozen GraphDef by node name"""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for n, _, node in _fields(open(path, "rb").read()):
        if n != 1:
            continue
        name = op = None
        value = None
        for m, _, v in _fields(node):
            if m == 1: name = v.decode()
            elif m == 2: op = v.decode()
            elif m == 5:
                kv = dict((k, x) for k, _, x in _fields(v))
                if kv.get(1) == b"value":
                    value = next((x for k, _, x in _fields(kv[2]) if k == 8), None)
        if op == "Const" and value is not None:
            t = _tensor_proto(value)
            if t is not None:
                out[name] = t
    return out


def write_frozen_graph(path: str, consts: dict) -> None:
    """minimal frozen GraphDef with one Const node per entry (test fixture generator)"""
    g = bytearray()
    for name, a in consts.items():
        a = np.asarray(a).copy(order="C")
        shape = _msg([(2, 2, _msg([(1, 0, int(d))])) for d in a.shape])
        tensor = _msg([(1, 0, _DT_OF[a.dtype]), (2, 2, shape), (4, 2, a.tobytes())])
        attr = _msg([(1, 2, b"value"), (2, 2, _msg([(8, 2, tensor)]))])
        g += _msg([(1, 2, _msg([(1, 2, name.encode()), (2, 2, b"Const"), (5, 2, attr)]))])
    open(path, "wb").write(bytes(g))


# --------------------------------------------------------------------------------------------- name maps
def _slim(scope: str, i: int, leaf: str) -> str:
    return f"{scope}/Conv{'' if i == 0 else '_' + str(i)}/{leaf}"


def monodepth_name_map(encoder: str) -> "OrderedDict[str, str]":
    """slot -> TF variable.  slim.conv2d names its layers Conv, Conv_1, ... in CREATION order inside each variable scope
    (model/encoder, model/decoder); the order below is the order of the conv() calls in monodepth_model.py
    (build_vgg / build_resnet50: encoder first, then upconv, iconv, get_disp per level from the top)."""
    m: "OrderedDict[str, str]" = OrderedDict()
    # weights.monodepth_weight_shapes lists the slots in that same creation order (encoder: conv1 / per resconv conv1, conv2,
    # conv3, projection shortcut; decoder: per level from the top upconv, iconv, disp)
    counters = {"enc": 0, "dec": 0}
    for s in W.monodepth_weight_shapes(encoder):
        if not s.endswith("/weights"):
            continue
        part = s.split("/")[0]
        scope = "model/encoder" if part == "enc" else "model/decoder"
        base = s[:-len("/weights")]
        m[base + "/weights"] = _slim(scope, counters[part], "weights")
        m[base + "/biases"] = _slim(scope, counters[part], "biases")
        counters[part] += 1
    return m


def fcn8s_name_map() -> "OrderedDict[str, str]":
    m: "OrderedDict[str, str]" = OrderedDict()
    for s in W.fcn8s_weight_shapes():
        part, layer, leaf = s.split("/")
        if part == "vgg":
            m[s] = f"{layer}/{'weights' if layer in ('fc6', 'fc7') and leaf == 'filter' else leaf}"
        else:       # tf.layers default names in creation order (fcn8s/fcn.py:165-213)
            if layer.startswith("score"):
                i = {"score7": 0, "score4": 1, "score3": 2}[layer]
                m[s] = f"conv2d{'' if i == 0 else '_' + str(i)}/{leaf}"
            else:
                i = int(layer[-1]) - 1
                m[s] = f"conv2d_transpose{'' if i == 0 else '_' + str(i)}/{leaf}"
    return m


def convert(tensors: dict, name_map: dict, shapes: dict) -> "OrderedDict[str, np.ndarray]":
    """pick the mapped variables (an optional ':0' suffix or leading scope is tolerated) and check their shapes"""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for slot, tfname in name_map.items():
        cand = [k for k in tensors if k == tfname or k == tfname + ":0" or k.endswith("/" + tfname)]
        if not cand:
            raise KeyError(f"{tfname} (for {slot}) not found; have e.g. {list(tensors)[:5]}")
        a = np.asarray(tensors[cand[0]], np.float32)
        if tuple(a.shape) != tuple(shapes[slot]):
            raise ValueError(f"{slot}: {tfname} has shape {a.shape}, expected {shapes[slot]}")
        out[slot] = a
    return out


def load_any(path: str, net: str, encoder: str | None = None) -> "OrderedDict[str, np.ndarray]":
    """what SegmentFrame.restore_model / DepthFrame.restore_model accept (semantic_depth.py:498-541, :627-653): a frozen
    GraphDef (``*.pb``), or a checkpoint prefix (a directory is searched for its ``*.index``).  ``net``: 'fcn8s' | 'monodepth'."""
    if net == "fcn8s":
        names, shapes = fcn8s_name_map(), W.fcn8s_weight_shapes()
    else:
        names, shapes = monodepth_name_map(encoder or "vgg"), W.monodepth_weight_shapes(encoder or "vgg")
    if path.endswith(".pb"):
        return convert(read_frozen_graph(path), names, shapes)
    if os.path.isdir(path):
        pbs = sorted(glob.glob(os.path.join(path, "*.pb")))
        idx = sorted(glob.glob(os.path.join(path, "*.index")) + glob.glob(os.path.join(path, "variables", "*.index")))
        if idx:
            return convert(read_tensor_bundle(idx[-1][:-6]), names, shapes)
        if pbs:
            return convert(read_frozen_graph(pbs[-1]), names, shapes)
        raise FileNotFoundError(f"no *.index / *.pb under {path}")
    prefix = path if os.path.exists(path + ".index") else glob.glob(path + "*.index")[0][:-6]
    return convert(read_tensor_bundle(prefix), names, shapes)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--monodepth", help="checkpoint prefix (model_cityscapes / model_kitti)")
    ap.add_argument("--encoder", default="resnet50", choices=["vgg", "resnet50"])
    ap.add_argument("--fcn8s", help="checkpoint prefix of a trained FCN-8s (variables of the VGG body + decoder)")
    ap.add_argument("--fcn8s-frozen", help="frozen GraphDef .pb")
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    if a.monodepth:
        prefix = a.monodepth if os.path.exists(a.monodepth + ".index") else glob.glob(a.monodepth + "*.index")[0][:-6]
        w = convert(read_tensor_bundle(prefix), monodepth_name_map(a.encoder), W.monodepth_weight_shapes(a.encoder))
    elif a.fcn8s:
        w = convert(read_tensor_bundle(a.fcn8s), fcn8s_name_map(), W.fcn8s_weight_shapes())
    elif a.fcn8s_frozen:
        w = convert(read_frozen_graph(a.fcn8s_frozen), fcn8s_name_map(), W.fcn8s_weight_shapes())
    else:
        ap.error("one of --monodepth / --fcn8s / --fcn8s-frozen")
    np.savez(a.out, **w)
    print(f"{len(w)} tensors -> {a.out}")


if __name__ == "__main__":
    main()
