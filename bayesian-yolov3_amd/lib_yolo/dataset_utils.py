"""Input feed of the inference scripts -- mirror of `lib_yolo/dataset_utils.py` `TestingDataset`
(:188-219) and `decode_img` (:6-11), without TensorFlow:

  TFRecord framing   u64 length | u32 masked-crc32c(length) | payload | u32 masked-crc32c(payload)
  payload            a `tf.train.Example` protobuf; features read (schema written by
                     `create_tf_records_citypersons.py:132-147`): `image/encoded` (PNG bytes),
                     `image/filename`, `image/height`, `image/width`
  image              PNG -> uint8 HWC -> float32 * (1/255)   (tf.image.convert_image_dtype)

The reference lists the shard files, interleaves them two at a time (cycle_length=2, block_length=1),
batches and prefetches one batch; this iterator does the same in sorted file order (the reference's
`list_files` order is shuffled and therefore not reproducible)."""
import ctypes
import glob
import io
import os
import struct

import numpy as np

from lib_yolo import model as _model

_MASK_DELTA = 0xA282EAD8


def _masked_crc(data):
    from byolo._lib import lib
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
    crc = lib.byolo_crc32c(buf, len(data))
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def read_tfrecords(path, verify_crc=True):
    """Yield the payload bytes of every record in a TFRecord file."""
    size = os.path.getsize(path)
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError('truncated TFRecord header in {}'.format(path))
            length, = struct.unpack('<Q', head[:8])
            len_crc, = struct.unpack('<I', head[8:])
            if verify_crc and _masked_crc(head[:8]) != len_crc:
                raise IOError('corrupt TFRecord length CRC in {}'.format(path))
            if length > size - f.tell():              # a corrupt length must not turn into a giant read
                raise IOError('truncated TFRecord in {}'.format(path))
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise IOError('truncated TFRecord in {}'.format(path))
            if verify_crc and _masked_crc(data) != struct.unpack('<I', tail)[0]:
                raise IOError('corrupt TFRecord data CRC in {}'.format(path))
            yield data


def write_tfrecords(path, payloads):
    """Inverse of read_tfrecords (test fixtures / small demo sets)."""
    with open(path, 'wb') as f:
        for data in payloads:
            head = struct.pack('<Q', len(data))
            f.write(head + struct.pack('<I', _masked_crc(head)) + data + struct.pack('<I', _masked_crc(data)))


# ---- minimal protobuf wire format (varint / length-delimited only) --------------------------------
def _varint(buf, pos):
    out = shift = 0
    while True:
        if pos >= len(buf) or shift > 63:
            raise ValueError('corrupt protobuf varint')
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if ln > n - pos:
                raise ValueError('corrupt protobuf length')
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type {}'.format(wt))
        yield num, wt, val


def parse_example(data):
    """tf.train.Example -> {feature name: list of bytes / int / float values}.  Malformed input raises ValueError."""
    try:
        return _parse_example(data)
    except (IndexError, TypeError, OverflowError, UnicodeDecodeError) as e:
        raise ValueError('corrupt tf.train.Example: {}'.format(e))


def _parse_example(data):
    out = {}
    for num, _, features in _fields(data):
        if num != 1:                                  # Example.features
            continue
        for fnum, _, entry in _fields(features):
            if fnum != 1:                             # Features.feature (map entry)
                continue
            name, feature = None, b''
            for k, _, v in _fields(entry):
                if k == 1:
                    name = bytes(v).decode('utf-8')
                elif k == 2:
                    feature = v
            vals = []
            for kind, _, lst in _fields(feature):     # 1 bytes_list, 2 float_list, 3 int64_list
                for vn, wt, v in _fields(lst):
                    if vn != 1:
                        continue
                    if kind == 1:
                        vals.append(bytes(v))
                    elif kind == 3:
                        if wt == 0:
                            vals.append(v - (1 << 64) if v >> 63 else v)
                        else:                         # packed
                            p = 0
                            while p < len(v):
                                x, p = _varint(v, p)
                                vals.append(x - (1 << 64) if x >> 63 else x)
                    elif kind == 2:
                        vals.extend(np.frombuffer(bytes(v), dtype='<f4').tolist())
            out[name] = vals
    return out


def _pb_bytes(num, payload):
    def vi(x):
        o = b''
        while True:
            b = x & 0x7F
            x >>= 7
            o += bytes([b | (0x80 if x else 0)])
            if not x:
                return o
    return vi((num << 3) | 2) + vi(len(payload)) + payload


def make_example(features):
    """{name: bytes | str | int} -> serialized tf.train.Example (fixtures / demos)."""
    def vi(x):
        x &= (1 << 64) - 1
        o = b''
        while True:
            b = x & 0x7F
            x >>= 7
            o += bytes([b | (0x80 if x else 0)])
            if not x:
                return o
    entries = b''
    for name, v in features.items():
        if isinstance(v, int):
            feat = _pb_bytes(3, vi((1 << 3) | 0) + vi(v))
        else:
            v = v.encode('utf-8') if isinstance(v, str) else v
            feat = _pb_bytes(1, _pb_bytes(1, v))
        entries += _pb_bytes(1, _pb_bytes(1, name.encode('utf-8')) + _pb_bytes(2, feat))
    return _pb_bytes(1, entries)


def decode_img(encoded, shape):
    """`lib_yolo/dataset_utils.py:6-11`: decode PNG and scale to [0, 1] as float32."""
    from PIL import Image
    img = np.asarray(Image.open(io.BytesIO(encoded)))
    if img.ndim == 2:
        img = img[:, :, None]
    if img.dtype != np.uint8:
        raise ValueError('only 8-bit PNGs are on this path (decode_png dtype=tf.uint8)')
    if tuple(img.shape) != tuple(shape):
        raise ValueError('image shape {} != config full_img_size {}'.format(img.shape, tuple(shape)))
    return img.astype(np.float32) * np.float32(1.0 / 255.0)        # convert_image_dtype(uint8 -> float32)


class TestingDataset:
    """Iterable of (images [B,H,W,C] float32, [filenames]) batches; `placeholder` is what the model is
    built on.  `batch_size` is the GLOBAL batch, as in the reference (`lib_yolo/dataset_utils.py:188-219`).
    Multi-GPU (torchrun, one process per GPU): `iter_shards(rank, world)` -- every rank reads the same record stream,
    parses the (cheap) Example protos of a global batch for the file names and decodes only the PNGs of ITS
    contiguous block of the batch axis (byolo.dist.shard_range)."""
    __test__ = False

    def __init__(self, config, config_key='data'):
        self.__config = config
        info = config[config_key]
        self.files = sorted(glob.glob(info['file_pattern']))
        self.batch_size = config['batch_size']
        self.shape = tuple(config['full_img_size'])
        self.verify_crc = info.get('verify_crc', True)
        self.placeholder = _model.Placeholder((self.batch_size,) + self.shape)

    def _records(self):
        # files.interleave(TFRecordDataset, cycle_length=2, block_length=1)
        pending = list(self.files)
        active = []
        while pending or active:
            while len(active) < 2 and pending:
                active.append(read_tfrecords(pending.pop(0), self.verify_crc))
            for it in list(active):
                try:
                    yield next(it)
                except StopIteration:
                    active.remove(it)

    def _global_batches(self):
        recs = []
        for rec in self._records():
            recs.append(rec)
            if len(recs) == self.batch_size:
                yield recs
                recs = []
        if recs:                                       # last, smaller batch (tf.data batch keeps the remainder)
            yield recs

    @staticmethod
    def _filename(feats):
        names = feats.get('image/filename', [])
        return names[0].decode('utf-8') if names else ''

    def parse_example(self, example):
        feats = parse_example(example)
        return decode_img(feats['image/encoded'][0], self.shape), self._filename(feats)

    def __iter__(self):
        for imgs, names, _ in self.iter_shards(0, 1):
            yield imgs, names

    def iter_shards(self, rank, world):
        """Yields (images of this rank's block [n_local,H,W,C], ALL filenames of the global batch, lo): the block is
        images [lo, lo + n_local) of the global batch; n_local may be 0 for a last, short batch."""
        from byolo.dist import shard_range
        for recs in self._global_batches():
            lo, hi = shard_range(len(recs), rank, world)
            feats = [parse_example(r) for r in recs]
            imgs = [decode_img(f['image/encoded'][0], self.shape) for f in feats[lo:hi]]
            x = np.stack(imgs) if imgs else np.empty((0,) + self.shape, dtype=np.float32)
            yield x, [self._filename(f) for f in feats], lo
