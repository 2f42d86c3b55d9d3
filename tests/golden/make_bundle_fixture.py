"""Write a TensorFlow "tensor bundle" checkpoint BY HAND from the published formats -- independently of
byolo/tf_checkpoint.py (nothing is imported from the package; own varint, CRC-32C, masking, block builder):

    python tests/golden/make_bundle_fixture.py        # -> tests/golden/tf_bundle/model-4242.{index,data-00000-of-00001}
                                                      #    (the expected tensors are tensors() below, seeded)

Layout choices are those of TensorFlow's BundleWriter / table builder as published (tensorflow/core/util/tensor_bundle/
tensor_bundle.cc, tensorflow/core/lib/io/table_builder.cc, format.cc, table_format.txt):
  * data file: tensors in key order, raw little-endian bytes, back to back;
  * index: LevelDB-format table -- data blocks of prefix-compressed entries with a restart point every 16 entries,
    each block followed by a 1-byte type (0 = no compression, what BundleWriter sets) and a masked CRC-32C of block +
    type; an (empty) metaindex block; an index block whose keys are SHORTEST SEPARATORS between the last key of a block
    and the first key of the next (FindShortestSeparator) and a SHORT SUCCESSOR after the last block; 48-byte footer;
  * key "" -> BundleHeaderProto {num_shards = 1, endianness = LITTLE, version {producer = 1}};
    key <variable name> -> BundleEntryProto {dtype, shape, shard_id = 0 (omitted), offset, size, crc32c (masked)}.
The block size is lowered to 700 bytes so that this small fixture spans several data blocks (TensorFlow's default is
256 KB; a real YOLOv3 index of ~370 variables has one or two).  Variable names are real ones of the reference's graph
(long shared prefixes), plus the optimizer slots and global_step a training checkpoint carries.
NOT written by TensorFlow: there is none in this environment.  What this pins is the reader against a second,
independent implementation of the published format."""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tf_bundle")


def varint(x):
    o = bytearray()
    while x >= 0x80:
        o.append((x & 0x7F) | 0x80)
        x >>= 7
    o.append(x)
    return bytes(o)


_T = []
for i in range(256):
    c = i
    for _ in range(8):
        c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    _T.append(c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = (c >> 8) ^ _T[(c ^ b) & 0xFF]
    return c ^ 0xFFFFFFFF


def masked(crc):                                    # crc32c::Mask
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def pb_varint_field(num, val):
    return varint(num << 3) + varint(val)


def pb_bytes_field(num, payload):
    return varint((num << 3) | 2) + varint(len(payload)) + payload


DT = {np.dtype("float32"): 1, np.dtype("int64"): 9, np.dtype("int32"): 3}


def entry_proto(arr, offset):
    raw = arr.tobytes()
    shape = b"".join(pb_bytes_field(2, pb_varint_field(1, d)) for d in arr.shape)        # TensorShapeProto.dim{size}
    e = pb_varint_field(1, DT[arr.dtype]) + pb_bytes_field(2, shape)
    if offset:
        e += pb_varint_field(4, offset)                                                   # proto3: zero fields are omitted
    e += pb_varint_field(5, len(raw))
    e += varint((6 << 3) | 5) + struct.pack("<I", masked(crc32c(raw)))                   # fixed32 crc32c
    return e, raw


def header_proto():
    return pb_varint_field(1, 1) + pb_bytes_field(3, pb_varint_field(1, 1))               # num_shards=1, (endianness=0), version{producer=1}


class BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.n, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, val):
        shared = 0
        if self.n % self.ri == 0 and self.n:
            self.restarts.append(len(self.buf))
        elif self.n:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += varint(shared) + varint(len(key) - shared) + varint(len(val)) + key[shared:] + val
        self.last, self.n = key, self.n + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def shortest_separator(a, b):                       # BytewiseComparator::FindShortestSeparator
    n = 0
    while n < min(len(a), len(b)) and a[n] == b[n]:
        n += 1
    if n < min(len(a), len(b)) and a[n] < 0xFF and a[n] + 1 < b[n]:
        return a[:n] + bytes([a[n] + 1])
    return a


def short_successor(a):                             # FindShortSuccessor
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def tensors():
    g = np.random.default_rng(4242)
    t = {}
    for scope, cin, cout, k in (("darknet53/conv", 3, 8, 3), ("darknet53/conv_1", 8, 16, 3), ("darknet53/conv_2", 16, 8, 1),
                                ("darknet53/conv_10", 8, 8, 3), ("det_net_1/conv", 8, 4, 1), ("det_net_1/conv_1", 4, 8, 3)):
        t[scope + "/conv2d/kernel"] = g.standard_normal((k, k, cin, cout)).astype(np.float32)
        for v in ("gamma", "beta", "moving_mean", "moving_variance"):
            t[scope + "/batch_normalization/" + v] = g.standard_normal(cout).astype(np.float32)
        t[scope + "/conv2d/kernel/Adam"] = np.zeros((k, k, cin, cout), np.float32)        # optimizer slots of a training run
        t[scope + "/conv2d/kernel/Adam_1"] = np.zeros((k, k, cin, cout), np.float32)
    t["det_net_1/detection/conv2d/kernel"] = g.standard_normal((1, 1, 8, 42)).astype(np.float32)
    t["det_net_1/detection/conv2d/bias"] = g.standard_normal(42).astype(np.float32)
    t["global_step"] = np.array(4242, dtype=np.int64)
    t["beta1_power"] = np.array(0.5, dtype=np.float32)                                    # a scalar: empty shape
    return t


def main(block_size=700, restart_interval=16, out=None):
    """`block_size`, `restart_interval`: table_builder.cc's Options::block_size (TensorFlow: 256 KB) and
    Options::block_restart_interval (16).  tests/test_checkpoint.py writes the same tensors under OTHER values of both into a
    temporary directory: a reader fitted to this one fixture's geometry would not read those."""
    out = out or OUT
    os.makedirs(out, exist_ok=True)
    t = tensors()
    prefix = os.path.join(out, "model-4242")
    keys = sorted(t, key=lambda s: s.encode())
    items, offset = [(b"", header_proto())], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for k in keys:
            e, raw = entry_proto(t[k], offset)
            f.write(raw)
            offset += len(raw)
            items.append((k.encode(), e))
    with open(prefix + ".index", "wb") as f:
        def emit(block):
            pos = f.tell()
            trailer = b"\x00"                                                              # kNoCompression
            f.write(block + trailer + struct.pack("<I", masked(crc32c(block + trailer))))
            return varint(pos) + varint(len(block))
        index, bb, pending = BlockBuilder(restart_interval=1), BlockBuilder(restart_interval), None
        for key, val in items:
            if pending is not None:                                                        # first key of the next block is known now
                index.add(shortest_separator(pending[0], key), pending[1])
                pending = None
            bb.add(key, val)
            if bb.size() >= block_size:
                pending = (bb.last, emit(bb.finish()))
                bb = BlockBuilder(restart_interval)
        if bb.n:
            pending = (bb.last, emit(bb.finish()))
        index.add(short_successor(pending[0]), pending[1])
        meta = emit(BlockBuilder().finish())
        idx = emit(index.finish())
        foot = meta + idx
        f.write(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", 0xDB4775248B80FB57))
    print("wrote", prefix, "index", os.path.getsize(prefix + ".index"), "B,", index.n, "data blocks; data", offset, "B")


# ---------------------------------------------------------------------------------------------------------------
# A TFRecord file of tf.train.Example protos, likewise assembled from the published formats (tensorflow/core/lib/io/
# record_writer.cc: u64 length, masked CRC-32C of the length, payload, masked CRC-32C of the payload;
# tensorflow/core/example/{example,feature}.proto) with the feature keys the reference's dataset writer uses
# (create_tf_records_citypersons.py:132-147) -- the input-feed side of SURVEY.md section 8(f1).  The PNGs are encoded by
# Pillow; the expected pixels are records() below.
def records(n=3, H=32, W=32):
    g = np.random.default_rng(99)
    return [("frame_%02d.png" % i, g.integers(0, 256, (H, W, 3), dtype=np.uint8)) for i in range(n)]


def example_proto(name, img_u8):
    import io
    from PIL import Image
    png = io.BytesIO()
    Image.fromarray(img_u8).save(png, format="PNG")

    def feature_bytes(b):
        return pb_bytes_field(1, pb_bytes_field(1, b))                    # Feature{bytes_list{value}}

    def feature_int(v):
        return pb_bytes_field(3, pb_bytes_field(1, varint(v)))            # Feature{int64_list{value (packed)}}

    feats = {"image/encoded": feature_bytes(png.getvalue()), "image/filename": feature_bytes(name.encode()),
             "image/format": feature_bytes(b"png"), "image/height": feature_int(img_u8.shape[0]),
             "image/width": feature_int(img_u8.shape[1])}
    entries = b"".join(pb_bytes_field(1, pb_bytes_field(1, k.encode()) + pb_bytes_field(2, v)) for k, v in sorted(feats.items()))
    return pb_bytes_field(1, entries)                                     # Example{features{feature map}}


def write_tfrecord():
    path = os.path.join(OUT, "ecp-day-val-00000-of-00001")
    with open(path, "wb") as f:
        for name, img in records():
            payload = example_proto(name, img)
            head = struct.pack("<Q", len(payload))
            f.write(head + struct.pack("<I", masked(crc32c(head))) + payload + struct.pack("<I", masked(crc32c(payload))))
    print("wrote", path, os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
    write_tfrecord()
