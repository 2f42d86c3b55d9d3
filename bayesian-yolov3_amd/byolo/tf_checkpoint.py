"""Reader (and minimal writer) for TensorFlow "tensor bundle" checkpoints -- what
`tf.train.Saver().restore(sess, checkpoint)` reads in the reference (`inference_epistemic.py:58`,
`detect.py:107`) -- without TensorFlow.

A checkpoint prefix `model-500000` consists of
  model-500000.index                 an SSTable (LevelDB table format) mapping  "" -> BundleHeaderProto and
                                     <variable name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
  model-500000.data-SSSSS-of-NNNNN   the raw little-endian tensor bytes
Table format: data blocks of prefix-compressed (shared, unshared, value_len, key_delta, value) entries + restart
array, each block followed by a 1-byte compression type (0 none, 1 snappy) and a masked CRC-32C; an index
block maps last keys to block handles; 48-byte footer {metaindex handle, index handle, padding, magic}.

Written from the published formats (TensorFlow tensor_bundle.proto / table_format.txt, Snappy format
description).  NOTE: there is no TensorFlow in the build environment, so this reader is validated only
against files produced by `write()` below and hand-made snappy streams (tests/test_checkpoint.py), not
against a checkpoint written by TensorFlow itself.
"""
import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 9: np.dtype('<i8'), 10: np.dtype(np.bool_)}
_DT_OF = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('int64'): 9}


# ---- varints / protobuf -------------------------------------------------------------------------
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(x):
    o = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        o.append(b | (0x80 if x else 0))
        if not x:
            return bytes(o)


def _fields(buf):
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            val = struct.unpack('<I', buf[pos:pos + 4])[0]
            pos += 4
        elif wt == 1:
            val = struct.unpack('<Q', buf[pos:pos + 8])[0]
            pos += 8
        else:
            raise ValueError('unsupported wire type %d' % wt)
        yield num, wt, val


def _parse_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for num, wt, val in _fields(buf):
        if num == 1:
            e['dtype'] = val
        elif num == 2:                                   # TensorShapeProto
            for n2, _, v2 in _fields(val):
                if n2 == 2:                              # Dim
                    size = 0
                    for n3, _, v3 in _fields(v2):
                        if n3 == 1:
                            size = v3
                    e['shape'].append(size)
        elif num == 3:
            e['shard_id'] = val
        elif num == 4:
            e['offset'] = val
        elif num == 5:
            e['size'] = val
        elif num == 6:
            e['crc32c'] = val
        elif num == 7:
            e['sliced'] = True
    return e


# ---- snappy ---------------------------------------------------------------------------------------
def snappy_decompress(buf):
    n, pos = _varint(buf, 0)
    out = bytearray()
    L = len(buf)
    while pos < L:
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                    # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('corrupt snappy stream')
        for _ in range(ln):                              # may overlap (run-length)
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy: length mismatch')
    return bytes(out)


# ---- table ----------------------------------------------------------------------------------------
def _masked_crc(data):
    from ._lib import lib
    import ctypes
    buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
    crc = lib.byolo_crc32c(buf, len(data))
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _read_block(f, offset, size, verify):
    end = os.fstat(f.fileno()).st_size
    if offset < 0 or size < 0 or offset + size + 5 > end:     # a corrupt handle must not become a giant read
        raise IOError('table block handle outside the file')
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) < size + 5:
        raise IOError('truncated table block')
    data, ctype, crc = raw[:size], raw[size], struct.unpack('<I', raw[size + 1:size + 5])[0]
    if verify and _masked_crc(raw[:size + 1]) != crc:
        raise IOError('table block CRC mismatch')
    if ctype == 1:
        data = snappy_decompress(data)
    elif ctype != 0:
        raise IOError('unknown block compression %d' % ctype)
    return data


def _block_entries(block):
    nrestarts = struct.unpack('<I', block[-4:])[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_index(prefix, verify=True):
    """{name: entry dict} plus the header under key ''.  A malformed file raises IOError."""
    try:
        return _read_index(prefix, verify)
    except (IndexError, struct.error, UnicodeDecodeError, TypeError, ValueError, OverflowError, MemoryError) as e:
        raise IOError('corrupt checkpoint index %s.index: %s' % (prefix, e))


def _read_index(prefix, verify):
    path = prefix + '.index'
    out = {}
    with open(path, 'rb') as f:
        f.seek(0, os.SEEK_END)
        size = f.tell()
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack('<Q', footer[40:])[0] != _MAGIC:
            raise IOError('%s is not a TF checkpoint index (bad magic)' % path)
        p = 0
        _, p = _varint(footer, p); _, p = _varint(footer, p)          # metaindex handle
        ioff, p = _varint(footer, p); isz, p = _varint(footer, p)     # index handle
        for _, handle in _block_entries(_read_block(f, ioff, isz, verify)):
            boff, q = _varint(handle, 0)
            bsz, q = _varint(handle, q)
            for key, val in _block_entries(_read_block(f, boff, bsz, verify)):
                out[key.decode('utf-8')] = val
    header = {'num_shards': 1}
    for num, _, val in _fields(out.pop('', b'')):
        if num == 1:
            header['num_shards'] = val
        elif num == 2 and val != 0:
            raise IOError('big-endian checkpoints are not supported')
    return header, {k: _parse_entry(v) for k, v in out.items()}


def read(prefix, names=None, verify=True):
    """Read the tensors of a TF checkpoint prefix into {variable name: ndarray}.  `names` limits the
    set; partitioned (sliced) variables are not supported."""
    header, entries = read_index(prefix, verify)
    out = {}
    files = {}
    try:
        for name, e in entries.items():
            if names is not None and name not in names:
                continue
            if e['sliced']:
                raise NotImplementedError('partitioned variable %s' % name)
            dt = _DTYPES.get(e['dtype'])
            if dt is None:
                continue                                  # strings etc.: nothing on this path needs them
            sid = e['shard_id']
            if sid not in files:
                files[sid] = open('%s.data-%05d-of-%05d' % (prefix, sid, header['num_shards']), 'rb')
            f = files[sid]
            if e['offset'] < 0 or e['size'] < 0 or e['offset'] + e['size'] > os.fstat(f.fileno()).st_size:
                raise IOError('truncated data for %s' % name)
            f.seek(e['offset'])
            raw = f.read(e['size'])
            if len(raw) != e['size']:
                raise IOError('truncated data for %s' % name)
            if verify and e['crc32c'] is not None and _masked_crc(raw) != e['crc32c']:
                raise IOError('data CRC mismatch for %s' % name)
            try:
                out[name] = np.frombuffer(raw, dtype=dt).reshape(e['shape']).copy()
            except (ValueError, TypeError) as err:
                raise IOError('corrupt entry for %s: %s' % (name, err))
    finally:
        for f in files.values():
            f.close()
    return out


# ---- writer (fixtures / converting .npz weights into the TF layout) -------------------------------
def _pb(num, wt, payload):
    key = _put_varint((num << 3) | wt)
    if wt == 0:
        return key + _put_varint(payload)
    if wt == 2:
        return key + _put_varint(len(payload)) + payload
    if wt == 5:
        return key + struct.pack('<I', payload)
    raise ValueError(wt)


def _block(entries):
    body = bytearray()
    for k, v in entries:                                  # no prefix sharing, one restart point
        body += _put_varint(0) + _put_varint(len(k)) + _put_varint(len(v)) + k + v
    body += struct.pack('<II', 0, 1)
    return bytes(body)


def write(prefix, tensors):
    """Write {name: ndarray} as a single-shard, uncompressed tensor bundle + a `checkpoint` state file."""
    names = sorted(tensors, key=lambda s: s.encode('utf-8'))
    data_path = '%s.data-00000-of-00001' % prefix
    entries = [(b'', _pb(1, 0, 1) + _pb(3, 2, _pb(1, 0, 1)))]          # header: num_shards=1, version.producer=1
    off = 0
    with open(data_path, 'wb') as f:
        for n in names:
            a = np.asarray(tensors[n])
            if a.ndim and not a.flags.c_contiguous:      # (ascontiguousarray would turn a scalar into shape (1,))
                a = np.ascontiguousarray(a)
            raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
            f.write(raw)
            shape = b''.join(_pb(2, 2, _pb(1, 0, int(d))) for d in a.shape)
            e = _pb(1, 0, _DT_OF[a.dtype]) + _pb(2, 2, shape) + (_pb(4, 0, off) if off else b'') + _pb(5, 0, len(raw)) + \
                _pb(6, 5, _masked_crc(raw))
            entries.append((n.encode('utf-8'), e))
            off += len(raw)
    with open(prefix + '.index', 'wb') as f:
        def put_block(b):
            pos = f.tell()
            f.write(b + b'\x00' + struct.pack('<I', _masked_crc(b + b'\x00')))
            return _put_varint(pos) + _put_varint(len(b))
        h_data = put_block(_block(entries))
        h_meta = put_block(_block([]))
        h_index = put_block(_block([(entries[-1][0] + b'\x00', h_data)]))
        footer = h_meta + h_index
        f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC))
    with open(os.path.join(os.path.dirname(prefix), 'checkpoint'), 'w') as f:
        base = os.path.basename(prefix)
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
