"""Input feed of the inference scripts -- mirror of `lib_yolo/dataset_utils.py` `TestingDataset`
(:188-219) and `decode_img` (:6-11), without TensorFlow:

  TFRecord framing   u64 length | u32 masked-crc32c(length) | payload | u32 masked-crc32c(payload)
  payload            a `tf.train.Example` protobuf; features read (schema written by
                     `create_tf_records_citypersons.py:132-147`): `image/encoded` (PNG bytes),
                     `image/filename`, `image/height`, `image/width`
  image              PNG -> uint8 HWC -> float32 * (1/255)   (tf.image.convert_image_dtype)

The reference's pipeline (:190-199) is  list_files -> interleave(TFRecordDataset, cycle_length=2, block_length=1) ->
map(parse_example, num_parallel_calls=config['cpu_thread_cnt']) -> batch -> prefetch(1).  Here:

  * records are taken in sorted file order, interleaved two files at a time (the reference's `list_files` order is shuffled
    and therefore not reproducible);
  * the map stage runs on `cpu_thread_cnt` native threads (csrc/host_io.cpp through the C-ABI: byolo_feed_records, one call
    per batch that holds no GIL): each reads its record's payload, checks the CRC, finds the PNG in the Example and inflates /
    unfilters it (zlib) straight into its frame of a batch buffer the caller may have pinned;
  * `prefetch` batches (config['data']['prefetch'], default max(2, cpu_thread_cnt / images per batch)) are decoded ahead of the
    consumer, several at a time, so that the pool is full at batch_size = 1 as well (the reference's default);
  * frames leave the host as uint8 -- the * (1/255) runs on the device (byolo_normalize_u8), bit-identical -- except through
    the plain iterator (`for imgs, names in dataset`), which yields the reference's float32 batches;
  * multi-GPU: a rank only reads, checks and decodes the records of ITS block of every global batch; the others are skipped by
    their framing (12 header bytes each)."""
import collections
import ctypes
import glob
import io
import os
import queue
import struct
import threading

import numpy as np

from lib_yolo import model as _model

_MASK_DELTA = 0xA282EAD8


def _masked_crc(data):
    from byolo._lib import lib
    crc = lib.byolo_crc32c(data if isinstance(data, bytes) else bytes(data), len(data))     # (ctypes drops the GIL for the call)
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
        return _parse_example(memoryview(data))         # nested messages are slices of one buffer, not copies
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




def decode_png_u8(encoded, shape):
    """`tf.image.decode_png(encoded, dtype=tf.uint8)` + `set_shape` of decode_img (`lib_yolo/dataset_utils.py:6-9`) for one
    record through the general decoder (Pillow): what the native pool falls back to for PNG flavours it does not read."""
    from PIL import Image
    img = np.asarray(Image.open(io.BytesIO(encoded)))
    if img.ndim == 2:
        img = img[:, :, None]
    if img.dtype != np.uint8:
        raise ValueError('only 8-bit PNGs are on this path (decode_png dtype=tf.uint8)')
    if tuple(img.shape) != tuple(shape):
        raise ValueError('image shape {} != config full_img_size {}'.format(img.shape, tuple(shape)))
    return img


def decode_img(encoded, shape):
    """`lib_yolo/dataset_utils.py:6-11`: decode PNG and scale to [0, 1] as float32."""
    return decode_png_u8(encoded, shape).astype(np.float32) * np.float32(1.0 / 255.0)      # convert_image_dtype(uint8 -> float32)


class _RecordFile:
    """Records of one TFRecord file by their framing only: (offset, length) of every payload; payloads are read on demand
    (os.pread: thread-safe), so a rank never touches the bytes of another rank's records."""

    def __init__(self, path, verify_crc):
        self.path, self.verify_crc = path, verify_crc
        self.fd = os.open(path, os.O_RDONLY)
        self.size = os.fstat(self.fd).st_size

    def __del__(self):
        try:
            os.close(self.fd)
        except Exception:
            pass

    def __iter__(self):
        pos = 0
        while pos < self.size:
            head = os.pread(self.fd, 12, pos)
            if len(head) < 12:
                raise IOError('truncated TFRecord header in {}'.format(self.path))
            length, = struct.unpack('<Q', head[:8])
            if self.verify_crc and _masked_crc(head[:8]) != struct.unpack('<I', head[8:])[0]:
                raise IOError('corrupt TFRecord length CRC in {}'.format(self.path))
            if length > self.size - pos - 16:         # a corrupt length must not turn into a giant read
                raise IOError('truncated TFRecord in {}'.format(self.path))
            yield (self, pos + 12, length)
            pos += 16 + length

    def payload(self, off, length):
        data = os.pread(self.fd, length + 4, off)
        if len(data) < length + 4:
            raise IOError('truncated TFRecord in {}'.format(self.path))
        if self.verify_crc and _masked_crc(data[:length]) != struct.unpack('<I', data[length:])[0]:
            raise IOError('corrupt TFRecord data CRC in {}'.format(self.path))
        return memoryview(data)[:length]


Shard = collections.namedtuple('Shard', 'u8 names lo n_global release error', defaults=(None,))
Shard.__doc__ = """One rank's block of one global batch: u8 [n_local,H,W,C] uint8 (a view of a batch buffer: call release() when
the frames have left it -- the buffer then goes back to the feeder), the block's file names, its offset `lo` in the global
batch (-> first_image: the dropout stream position) and the global batch's size.  `error` (iter_shards_u8(errors='yield')
only): the exception that reading / checking / decoding this rank's records of the batch raised -- the shard is then empty."""


class TestingDataset:
    """Iterable of (images [B,H,W,C] float32, [filenames]) batches; `placeholder` is what the model is
    built on.  `batch_size` is the GLOBAL batch, as in the reference (`lib_yolo/dataset_utils.py:188-219`).
    The entry points read it through `iter_shards_u8(rank, world)` (module docstring)."""
    __test__ = False

    def __init__(self, config, config_key='data'):
        self.__config = config
        info = config[config_key]
        self.files = sorted(glob.glob(info['file_pattern']))
        self.batch_size = config['batch_size']
        self.shape = tuple(config['full_img_size'])
        self.verify_crc = info.get('verify_crc', True)
        self.threads = max(1, int(config.get('cpu_thread_cnt', 1)))          # num_parallel_calls of the map stage (:196)
        # one process per GPU: N ranks x cpu_thread_cnt decode threads must fit the host (8 x 24 = 192 on a 256-thread node is fine,
        # 8 x 24 on a 64-thread one is not): at most the rank's share of the hardware threads (LOCAL_WORLD_SIZE: torchrun)
        local_world = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1))
        self.threads = max(1, min(self.threads, (os.cpu_count() or 1) // local_world))
        self.prefetch = int(info.get('prefetch', 0))                         # batches decoded ahead (:199 prefetches 1); 0 = derive it (iter_shards_u8)
        self.placeholder = _model.Placeholder((self.batch_size,) + self.shape)

    def _records(self):
        # files.interleave(TFRecordDataset, cycle_length=2, block_length=1): record handles, no payload is read here
        pending = list(self.files)
        active = []
        while pending or active:
            while len(active) < 2 and pending:
                active.append(iter(_RecordFile(pending.pop(0), self.verify_crc)))
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

    def _load_block(self, recs, buf, threads=None):
        """The map stage for the records of one block: payload (+ CRC) -> Example -> PNG decoded into buf[j] ([H,W,C] uint8), on
        `cpu_thread_cnt` native threads in ONE call that holds no GIL (byolo_feed_records); returns the file names."""
        from byolo import _lib
        n = len(recs)
        if n == 0:
            return []
        i32, i64 = ctypes.c_int32, ctypes.c_int64
        fds = (i32 * n)(*[r[0].fd for r in recs])
        offs = (i64 * n)(*[r[1] for r in recs])
        lens = (i64 * n)(*[r[2] for r in recs])
        status = (i32 * n)()
        found = (i32 * (3 * n))()
        cap = 1024
        names = ctypes.create_string_buffer(n * cap)
        h, w, c = self.shape
        assert buf.dtype == np.uint8 and buf.flags['C_CONTIGUOUS'] and buf.shape[0] >= n and tuple(buf.shape[1:]) == self.shape
        rc = _lib.lib.byolo_feed_records(fds, offs, lens, n, int(bool(self.verify_crc)), h, w, c, ctypes.c_void_p(buf.ctypes.data),
                                         int(threads or self.threads), names, cap, status, found)
        if rc < 0:
            raise ValueError('byolo_feed_records: bad argument (full_img_size {})'.format(self.shape))
        out = []
        for j, (rf, off, length) in enumerate(recs):
            st = status[j]
            if st == _lib.PNG_OK:
                out.append(names[j * cap:(j + 1) * cap].split(b'\0', 1)[0].decode('utf-8'))
                continue
            if st == _lib.FEED_IO:
                raise IOError('truncated TFRecord in {}'.format(rf.path))
            if st == _lib.FEED_CRC:
                raise IOError('corrupt TFRecord data CRC in {}'.format(rf.path))
            if st == _lib.PNG_SHAPE:
                raise ValueError('image shape {} != config full_img_size {}'.format(tuple(found[3 * j:3 * j + 3]), self.shape))
            # a malformed Example, or a PNG flavour the native decoder does not read (interlaced / palette / 16-bit) or finds
            # damaged: the general path decides (and raises what it always raised)
            feats = parse_example(rf.payload(off, length))
            enc = feats.get('image/encoded')
            if not enc:
                raise ValueError('record without image/encoded in {}'.format(rf.path))
            buf[j] = decode_png_u8(enc[0], self.shape)
            out.append(self._filename(feats))
        return out

    def __iter__(self):
        for sh in self.iter_shards_u8(0, 1):
            x = sh.u8.astype(np.float32) * np.float32(1.0 / 255.0)          # convert_image_dtype(uint8 -> float32)
            sh.release()
            yield x, sh.names

    def iter_shards_u8(self, rank, world, alloc=None, extra_buffers=2, errors='raise'):
        """Yields a `Shard` per global batch: the frames of this rank's contiguous block (byolo.dist.shard_range; may be empty
        for a last, short batch).  alloc(shape) -> uint8 ndarray provides the batch buffers (the driver passes pinned memory);
        `prefetch` + 1 + extra_buffers of them circulate: `prefetch` decoded or being decoded ahead of the consumer,
        extra_buffers with the consumer.

        The map stage runs `cpu_thread_cnt` decodes at a time whatever the batch size: up to `prefetch` batches are in the
        decode pool TOGETHER (each on min(cpu_thread_cnt, images of its block) native threads) and are handed over in order --
        at the reference's default batch_size = 1 (inference_epistemic.py:222) a 1024 x 1920 frame is 50 ms of inflate on one
        core, so one batch at a time would feed 20 img/s to a device that takes 41.  `prefetch` defaults to
        max(2, cpu_thread_cnt / images per block), at most 16.

        errors='raise': a record of this rank's block that cannot be read / checked / decoded raises here, at its batch's position.
        errors='yield': the batch arrives as an empty Shard carrying the exception (`error`) and the iteration goes on -- the
        multi-GPU driver needs that: the other ranks read only THEIR records and are about to enter the batch's collective, so
        this rank must take part in it (with a 'feed failed' status word) instead of leaving (byolo/inference.py)."""
        from concurrent.futures import ThreadPoolExecutor
        from byolo.dist import shard_range
        alloc = alloc or (lambda shape: np.empty(shape, dtype=np.uint8))
        cap = max(1, shard_range(self.batch_size, 0, world)[1])              # the largest block of a full batch
        prefetch = self.prefetch if self.prefetch > 0 else max(2, min(16, -(-self.threads // cap)))
        # `prefetch` calls decode at once: each gets its share of cpu_thread_cnt (a block of 64 images used to start 2 x 24 threads)
        # (a LOCAL of this iterator: two iterators of one dataset -- another world, another prefetch -- keep their own)
        per_call = max(1, -(-self.threads // prefetch))
        free = queue.Queue()
        for _ in range(prefetch + 1 + max(1, extra_buffers)):
            free.put(alloc((cap,) + self.shape))
        ready = queue.Queue(maxsize=prefetch)                                # futures, in batch order
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def load(recs, buf, lo, n_glob):
            try:
                return buf, len(recs), self._load_block(recs, buf, threads=per_call), lo, n_glob
            except Exception as e:                                           # a failed record reaches the consumer in order
                return buf, 0, e, lo, n_glob

        def feeder():
            pool = ThreadPoolExecutor(max_workers=prefetch, thread_name_prefix='byolo-feed')
            try:
                for recs in self._global_batches():
                    lo, hi = shard_range(len(recs), rank, world)
                    buf = None
                    while buf is None:
                        if stop.is_set():
                            return
                        try:
                            buf = free.get(timeout=0.1)
                        except queue.Empty:
                            pass
                    if not put(pool.submit(load, recs[lo:hi], buf, lo, len(recs))):
                        return
                put(None)
            except BaseException as e:                                        # (the record index itself: framing errors) delivered in order
                put(e)
            finally:
                pool.shutdown(wait=True)

        th = threading.Thread(target=feeder, name='byolo-feeder', daemon=True)
        th.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                buf, n, names, lo, n_glob = item.result()
                if isinstance(names, Exception):
                    if errors != 'yield':
                        raise names
                    yield Shard(buf[:0], [], lo, n_glob, (lambda b=buf: free.put(b)), names)
                    continue
                yield Shard(buf[:n], names, lo, n_glob, (lambda b=buf: free.put(b)))
        finally:
            stop.set()
            th.join(timeout=10)
