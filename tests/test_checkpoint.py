"""TF tensor-bundle checkpoint reader (byolo/tf_checkpoint.py) -- format-level tests: round trip through
the bundled writer, prefix-compressed / multi-block / snappy-compressed index tables built by hand, CRC
checks, and the `find_checkpoint` + `restore` flow of the inference scripts (no GPU: restore stops at the
parameter store)."""
import os
import struct

import numpy as np
import pytest

from byolo import tf_checkpoint as tc


def _tensors():
    g = np.random.default_rng(0)
    return {"darknet53/conv/conv2d/kernel": g.standard_normal((3, 3, 3, 32)).astype(np.float32),
            "darknet53/conv/batch_normalization/gamma": g.standard_normal(32).astype(np.float32),
            "det_net_1/detection/conv2d/bias": g.standard_normal(42).astype(np.float32),
            "global_step": np.array(500000, dtype=np.int64),
            "darknet53/conv/conv2d/kernel/Adam": np.zeros((3, 3, 3, 32), np.float32)}


def test_round_trip(tmp_path):
    t = _tensors()
    prefix = str(tmp_path / "model-500000")
    tc.write(prefix, t)
    got = tc.read(prefix)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and np.array_equal(got[k], t[k]), k
    only = tc.read(prefix, names={"global_step"})
    assert list(only) == ["global_step"] and int(only["global_step"]) == 500000
    # corrupt one data byte -> CRC failure
    p = prefix + ".data-00000-of-00001"
    raw = bytearray(open(p, "rb").read()); raw[10] ^= 1
    open(p, "wb").write(bytes(raw))
    with pytest.raises(IOError, match="CRC"):
        tc.read(prefix)
    assert tc.read(prefix, verify=False)                       # still readable without verification


def test_snappy_known_streams():
    # literal only
    assert tc.snappy_decompress(bytes([5, 4 << 2]) + b"hello") == b"hello"
    # literal "ab" + copy(offset 2, len 6) -> run-length overlap
    s = bytes([8, 1 << 2]) + b"ab" + bytes([(2 << 2) | 1 | (0 << 5), 2])      # kind 1: len = 2 + 4 = 6
    assert tc.snappy_decompress(s) == b"abababab"
    # 2-byte-offset copy
    s = bytes([7, 2 << 2]) + b"xyz" + bytes([((4 - 1) << 2) | 2, 3, 0])
    assert tc.snappy_decompress(s) == b"xyzxyzx"
    with pytest.raises(ValueError):
        tc.snappy_decompress(bytes([3, 0 << 2]) + b"a")         # declared 3 bytes, got 1


def _snappy_literal(data):
    """Valid snappy stream consisting of literals only."""
    out = bytearray(tc._put_varint(len(data)))
    for i in range(0, len(data), 60):
        chunk = data[i:i + 60]
        out.append((len(chunk) - 1) << 2)
        out += chunk
    return bytes(out)


def test_prefix_compressed_multiblock_snappy_index(tmp_path):
    """An index as TensorFlow's table builder would lay it out: shared key prefixes, several data blocks,
    snappy-compressed blocks."""
    t = _tensors()
    prefix = str(tmp_path / "model-7")
    tc.write(prefix, t)
    header, entries = tc.read_index(prefix)
    # rebuild the index by hand
    names = sorted(entries, key=lambda s: s.encode())
    raw_entries = {}
    with open(prefix + ".index", "rb") as f:
        blob = f.read()
    # recover the serialized BundleEntryProtos through the public reader's block parser
    size = len(blob)
    footer = blob[size - 48:]
    p = 0
    _, p = tc._varint(footer, p); _, p = tc._varint(footer, p)
    ioff, p = tc._varint(footer, p); isz, p = tc._varint(footer, p)
    with open(prefix + ".index", "rb") as f:
        (k, handle), = list(tc._block_entries(tc._read_block(f, ioff, isz, True)))
        boff, q = tc._varint(handle, 0); bsz, q = tc._varint(handle, q)
        kv = list(tc._block_entries(tc._read_block(f, boff, bsz, True)))

    def block(items):                                            # prefix-compressed, restart every 2 entries
        body, restarts, prev = bytearray(), [], b""
        for i, (key, val) in enumerate(items):
            if i % 2 == 0:
                restarts.append(len(body)); shared = 0
            else:
                shared = len(os.path.commonprefix([prev, key]))
            body += tc._put_varint(shared) + tc._put_varint(len(key) - shared) + tc._put_varint(len(val)) + key[shared:] + val
            prev = key
        for r in restarts:
            body += struct.pack("<I", r)
        return bytes(body + struct.pack("<I", len(restarts)))

    with open(prefix + ".index", "wb") as f:
        def put(b, snappy):
            payload = _snappy_literal(b) if snappy else b
            pos = f.tell()
            t_ = bytes([1 if snappy else 0])
            f.write(payload + t_ + struct.pack("<I", tc._masked_crc(payload + t_)))
            return tc._put_varint(pos) + tc._put_varint(len(payload))
        h1 = put(block(kv[:3]), True)
        h2 = put(block(kv[3:]), False)
        hm = put(block([]), False)
        hi = put(block([(kv[2][0], h1), (kv[-1][0] + b"\x00", h2)]), True)
        foot = hm + hi
        f.write(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", tc._MAGIC))
    got = tc.read(prefix)
    for k in t:
        assert np.array_equal(got[k], t[k]), k


def test_hand_assembled_bundle_fixture():
    """tests/golden/tf_bundle/: a checkpoint assembled byte by byte from the published formats by
    tests/golden/make_bundle_fixture.py -- its own varint / CRC-32C / masking / block builder, nothing imported from
    byolo -- laid out the way TensorFlow's BundleWriter and table builder do it: restart interval 16, prefix-compressed
    keys over several data blocks, shortest-separator index keys, proto3 zero-field omission (offset 0, scalar shape),
    optimizer slots and global_step beside the model variables.  (Not written by TensorFlow -- none here; this pins
    the reader against a second implementation of the format.)"""
    import importlib.util
    from conftest import GOLDEN
    spec = importlib.util.spec_from_file_location("make_bundle_fixture", os.path.join(GOLDEN, "make_bundle_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    want = mk.tensors()
    prefix = os.path.join(GOLDEN, "tf_bundle", "model-4242")
    got = tc.read(prefix)
    assert set(got) == set(want)
    for k, v in want.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    assert got["beta1_power"].shape == () and int(got["global_step"]) == 4242
    header, entries = tc.read_index(prefix)
    assert len(entries) == len(want)
    sub = tc.read(prefix, names={k for k in want if "Adam" not in k and k not in ("global_step", "beta1_power")})
    assert len(sub) == 6 * 5 + 2                                     # what a restore into the model asks for


@pytest.mark.parametrize("block_size,restart_interval", [(262144, 16), (97, 16), (700, 1), (700, 2), (1500, 64), (4096, 3)])
def test_bundles_of_other_block_sizes_and_restart_intervals(block_size, restart_interval, tmp_path):
    """The committed fixture has ONE geometry (block size 700, restart interval 16).  The independent writer
    (tests/golden/make_bundle_fixture.py) lays the same tensors out under other values of table_builder.cc's two knobs --
    TensorFlow's own default block size (256 KB: the whole index is one data block, as for a real YOLOv3 checkpoint), blocks of
    less than one entry's size (every entry its own block: only restart points, shared = 0), restart at every entry / every
    second entry (no or little prefix compression), an interval longer than a block, an odd one -- and the reader must return
    the same tensors.  Which published rule each byte follows:
      entry        table_format.txt "shared_bytes: varint32 | unshared_bytes: varint32 | value_length: varint32 | key_delta | value";
      block tail   table_format.txt "restarts: uint32[num_restarts] | num_restarts: uint32"; shared_bytes = 0 at a restart point;
      trailer      format.cc kBlockTrailerSize = 5: "type: uint8 | crc: uint32" with crc32c::Mask(crc of block + type);
      index block  block handles "offset: varint64 | size: varint64", keys by FindShortestSeparator / FindShortSuccessor;
      footer       format.h Footer::kEncodedLength = 2 * BlockHandle::kMaxEncodedLength + 8 = 48, magic 0xdb4775248b80fb57;
      values       tensor_bundle.proto BundleHeaderProto / BundleEntryProto, proto3 wire format (zero fields omitted)."""
    import importlib.util
    from conftest import GOLDEN
    spec = importlib.util.spec_from_file_location("make_bundle_fixture", os.path.join(GOLDEN, "make_bundle_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    mk.main(block_size=block_size, restart_interval=restart_interval, out=str(tmp_path))
    want = mk.tensors()
    got = tc.read(str(tmp_path / "model-4242"))
    assert set(got) == set(want)
    for k, v in want.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    # and the committed fixture is byte-identical to what the writer makes at ITS geometry (the generator is the fixture's definition)
    if (block_size, restart_interval) == (700, 1):
        mk.main(block_size=700, restart_interval=16, out=str(tmp_path / "same"))
        for ext in (".index", ".data-00000-of-00001"):
            assert open(os.path.join(GOLDEN, "tf_bundle", "model-4242" + ext), "rb").read() == open(str(tmp_path / "same" / ("model-4242" + ext)), "rb").read()


def test_find_and_restore(tmp_path):
    from byolo import inference as inf
    from conftest import build_model, golden_params
    params = golden_params("yolov3")
    ck = tmp_path / "ckpts" / "run1"
    ck.mkdir(parents=True)
    tc.write(str(ck / "model-100"), {"x": np.zeros(1, np.float32)})
    extra = dict(params); extra["global_step"] = np.array(200, np.int64)
    tc.write(str(ck / "model-200"), extra)                      # newest; also updates the `checkpoint` state file
    cfg = {"checkpoint_path": str(tmp_path / "ckpts"), "run_id": "run1", "step": "last"}
    path = inf.find_checkpoint(cfg)
    assert path.endswith("model-200.index") and inf.step_of(path) == "200"
    cfg["step"] = 100
    assert inf.find_checkpoint(cfg).endswith("model-100.index")
    cfg["step"] = 123
    with pytest.raises(AssertionError, match="could not find checkpoint"):
        inf.find_checkpoint(cfg)
    # restore by variable name into the engine's parameter store (finalize needs a GPU: stop before it)
    _, m = build_model("yolov3", 64, 64)
    m.engine.set_params(tc.read(os.path.splitext(path)[0]), strict=True)
    for k, v in params.items():
        assert np.array_equal(m.engine.get_param(k, v.shape), v)


def test_corrupt_files_raise_io_errors(tmp_path):
    """Random damage to either file of a checkpoint (byte flips, deletions, insertions; found by an ad-hoc fuzz: struct /
    index / decode errors and a MemoryError from a corrupt length used to escape): the reader either returns the
    tensors or raises IOError / NotImplementedError -- never another exception type, never a giant allocation."""
    import glob
    import random
    prefix = str(tmp_path / "model.ckpt-1")
    tc.write(prefix, _tensors())
    files = {f: open(f, "rb").read() for f in glob.glob(prefix + "*")}
    rnd = random.Random(3)
    outcomes = set()
    for i in range(400):
        for f, good in files.items():
            b = bytearray(good)
            for _ in range(rnd.randint(0, 3)):
                op = rnd.randint(0, 2)
                if op == 0:
                    b[rnd.randrange(len(b))] = rnd.randrange(256)
                elif op == 1:
                    del b[rnd.randrange(len(b)):][:rnd.randint(1, 40)]
                else:
                    at = rnd.randrange(len(b))
                    b[at:at] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 16)))
            open(f, "wb").write(bytes(b))
        try:
            tc.read(prefix, verify=bool(i % 2))
            outcomes.add("ok")
        except (IOError, NotImplementedError):
            outcomes.add("refused")
    assert outcomes == {"ok", "refused"}
