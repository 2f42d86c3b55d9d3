"""N > 1 path on CPU: two processes, gloo backend (rendezvous on 127.0.0.1)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_allgather_world2(tmp_path):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), "2", str(port), str(tmp_path)],
                              env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(2)]
    for r in range(2):
        # every rank sees every rank's images, rank-major, bit-exact (incl. the int32 bit-casts)
        for k, gk in (("rows", "g_rows"), ("kept", "g_kept"), ("count", "g_count")):
            want = torch.cat([res[0][k], res[1][k]], 0)
            assert torch.equal(res[r][gk], want), (r, k)
    assert res[0]["shard"] == (0, 4) and res[1]["shard"] == (4, 7)
    # byolo.dist.agree_on_any_error: the failing rank keeps its own exception, its peer learns of it (the fp32 twin)
    assert res[0]["agree"][0] and res[1]["agree"][0]
    assert res[1]["agree"][1] == "MemoryError" and res[0]["agree"][1] == "RuntimeError" and "rank 1 failed" in res[0]["agree"][2]
    for r in range(2):                               # the short batch: global order, trimmed, identical on both ranks
        assert [t.shape[0] for t in res[r]["u_rows"]] == [1, 2, 3]
        for g in range(3):
            assert torch.equal(res[r]["u_rows"][g], (100.0 * g + torch.arange(23.0)).expand(g + 1, 23))
            assert res[r]["u_kept"][g].tolist() == [1000 * g + i for i in range(g + 1)]


def _records(tmp_path, n, H=32, W=32):
    import io
    import numpy as np
    from PIL import Image
    from lib_yolo import dataset_utils as du
    rng = np.random.default_rng(5)
    exs, imgs = [], []
    for i in range(n):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        b = io.BytesIO(); Image.fromarray(img).save(b, format="PNG")
        exs.append(du.make_example({"image/encoded": b.getvalue(), "image/filename": "f%02d.png" % i, "image/height": H, "image/width": W}))
        imgs.append(img.astype(np.float32) * np.float32(1 / 255.))
    du.write_tfrecords(str(tmp_path / "val-00000-of-00001"), exs)
    return imgs


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_dataset_shards_cover_every_global_batch(tmp_path, world):
    """TestingDataset.iter_shards_u8: batch_size is the GLOBAL batch (as in the reference); rank r gets the contiguous
    block shard_range(len(batch), r, world) of every global batch as uint8 frames, its offset `lo` (-> first_image: the dropout
    stream position), the size of the global batch and the file names of ITS block; the blocks of the ranks tile the batch, also
    the last short one (blocks may be empty).  The plain iterator yields the reference's float32 batches."""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "bayesian-yolov3_amd"))
    from lib_yolo import dataset_utils as du
    from byolo import dist as bdist
    imgs = _records(tmp_path, 11)
    cfg = {"batch_size": 4, "full_img_size": [32, 32, 3], "cpu_thread_cnt": 3, "data": {"file_pattern": str(tmp_path / "val-*")}}
    per_rank = []
    for r in range(world):
        got = []
        for sh in du.TestingDataset(cfg).iter_shards_u8(r, world):
            got.append((sh.u8.copy(), sh.names, sh.lo, sh.n_global))
            sh.release()
        per_rank.append(got)
    plain = list(du.TestingDataset(cfg))
    assert [len(n) for _, n in plain] == [4, 4, 3]
    for step in range(3):
        names = plain[step][1]
        assert names == ["f%02d.png" % (4 * step + i) for i in range(len(names))]
        pos = 0
        for r in range(world):
            x, n_r, lo, n_glob = per_rank[r][step]
            assert n_glob == len(names) and lo == pos == bdist.shard_range(len(names), r, world)[0]
            assert n_r == names[lo:lo + x.shape[0]]
            assert x.dtype == np.uint8 and x.shape[1:] == (32, 32, 3)
            for j in range(x.shape[0]):
                assert np.array_equal(x[j].astype(np.float32) * np.float32(1 / 255.), imgs[4 * step + lo + j])
                assert np.array_equal(plain[step][0][lo + j], imgs[4 * step + lo + j])
            pos += x.shape[0]
        assert pos == len(names)


def test_feed_prefetches_and_stops_cleanly(tmp_path):
    """The feeder thread decodes `prefetch` batches ahead and no further (its buffers are finite: back-pressure), hands a
    record's failure to the consumer at that batch's position, and goes away when the consumer stops early."""
    import threading
    import time
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "bayesian-yolov3_amd"))
    from lib_yolo import dataset_utils as du
    _records(tmp_path, 12)
    cfg = {"batch_size": 2, "full_img_size": [32, 32, 3], "cpu_thread_cnt": 2, "data": {"file_pattern": str(tmp_path / "val-*"), "prefetch": 2}}
    ds = du.TestingDataset(cfg)
    loaded = []
    orig = ds._load_block
    ds._load_block = lambda recs, buf, **kw: (loaded.append(len(recs)), orig(recs, buf, **kw))[1]
    it = ds.iter_shards_u8(0, 1, extra_buffers=1)
    first = next(it)
    time.sleep(0.5)
    # 1 with the consumer + 2 waiting in the queue + 1 being held by the feeder for the next put = 4 batches, not all 6
    assert 3 <= len(loaded) <= 4, loaded
    first.release()
    it.close()
    time.sleep(0.3)
    assert not [t for t in threading.enumerate() if t.name == "byolo-feeder"]
    # a corrupt record: batches before it arrive, then the error
    p = str(tmp_path / "val-00000-of-00001")
    raw = bytearray(open(p, "rb").read())
    recs = list(du._RecordFile(p, True))
    raw[recs[5][1] + 40] ^= 0xFF                       # inside the 6th record's payload (batch 2)
    open(p, "wb").write(bytes(raw))
    got = []
    with pytest.raises(IOError, match="CRC"):
        for sh in du.TestingDataset(cfg).iter_shards_u8(0, 1):
            got.append(sh.names); sh.release()
    assert got == [["f00.png", "f01.png"], ["f02.png", "f03.png"]]
    # ... which another rank's block does not see: rank 1 of 2 owns the odd records of every batch, record 5 is odd
    got = []
    for sh in du.TestingDataset(cfg).iter_shards_u8(0, 2):
        got.append(sh.names); sh.release()
    assert got == [["f%02d.png" % i] for i in range(0, 12, 2)]


def _run_infer(world, data_dir, out_path, timeout=300, extra=()):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_infer_worker.py"), str(r), str(world), str(port), str(data_dir), out_path] + [str(e) for e in extra],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    return [p.returncode for p in procs], outs


def test_inference_epistemic_world2_writes_the_single_process_json(tmp_path):
    """`torchrun --nproc-per-node 2 inference_epistemic.py` end to end up to the engine boundary, on CPU (gloo): the driver loop
    of the entry point with a stand-in engine whose rows depend on the pixels and on the image's position in the GLOBAL batch.
    11 images, global batch 5 (so the last batch is short and rank 1's block of it is EMPTY): the two-process job must write
    byte-identical ECP JSON files to the one-process run, every rank's engine option `device` is its LOCAL_RANK, and the
    blocks / first_image values tile every global batch (inference_epistemic.py:56-83 + byolo/dist.py)."""
    import json
    _records(tmp_path, 11)
    rc1, o1 = _run_infer(1, tmp_path, str(tmp_path / "one" / "run"))
    assert rc1 == [0], o1
    rc2, o2 = _run_infer(2, tmp_path, str(tmp_path / "two" / "run"))
    assert rc2 == [0, 0], o2
    a, b = str(tmp_path / "one" / "run_0"), str(tmp_path / "two" / "run_0")
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == ["f%02d.json" % i for i in range(11)]
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f
        assert json.load(open(os.path.join(a, f)))["children"]
    c = [json.load(open(tmp_path / ("calls_w2_r%d.json" % r))) for r in range(2)]
    assert c[0]["options"] == {"device": 0} and c[1]["options"] == {"device": 1}
    # (images in the block, first_image, seed) per global batch: 5 = 3 + 2, 5 = 3 + 2, 1 = 1 + 0 (rank 1 skips the call)
    assert c[0]["calls"] == [[3, 0, 4], [3, 0, 5], [1, 0, 6]] and c[1]["calls"] == [[2, 3, 4], [2, 3, 5]]
    # every rank wrote the files of ITS images, through the native formatter
    assert [c[r]["stats"]["images"] for r in range(2)] == [7, 4] and all(c[r]["stats"]["native_json"] for r in range(2))
    one = json.load(open(tmp_path / "calls_w1_r0.json"))
    assert one["calls"] == [[5, 0, 4], [5, 0, 5], [1, 0, 6]] and one["stats"]["images"] == 11 and one["stats"]["batches"] == 3


def test_inference_epistemic_world8_writes_the_single_process_json(tmp_path):
    """The shape of the driver's round-end 8-GPU run, on CPU (gloo, eight processes, the stand-in engine): 11 images, global batch
    5 -- blocks of 1+1+1+1+1+0+0+0, then the same, then 1+0+...+0: ranks 5 - 7 never own an image and still take part in every
    collective, rank 0 creates the directory, every rank writes only its own files; the job must leave byte-identical JSON to the
    one-process run."""
    import json
    _records(tmp_path, 11)
    rc1, o1 = _run_infer(1, tmp_path, str(tmp_path / "one" / "run"))
    assert rc1 == [0], o1
    rc8, o8 = _run_infer(8, tmp_path, str(tmp_path / "eight" / "run"), timeout=600)
    assert rc8 == [0] * 8, o8
    a, b = str(tmp_path / "one" / "run_0"), str(tmp_path / "eight" / "run_0")
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == ["f%02d.json" % i for i in range(11)]
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f
    c = [json.load(open(tmp_path / ("calls_w8_r%d.json" % r))) for r in range(8)]
    assert [c[r]["options"] for r in range(8)] == [{"device": r} for r in range(8)]
    assert [c[r]["stats"]["images"] for r in range(8)] == [3, 2, 2, 2, 2, 0, 0, 0]
    assert c[0]["calls"] == [[1, 0, 4], [1, 0, 5], [1, 0, 6]] and c[4]["calls"] == [[1, 4, 4], [1, 4, 5]] and c[7]["calls"] == []
    assert all(c[r]["stats"]["batches"] == 3 for r in range(8))


@pytest.mark.parametrize("raise_rank,raise_call", [(1, 1), (0, 2), (0, 3)])
def test_inference_world2_ranks_switch_to_fp32_together(tmp_path, raise_rank, raise_call):
    """BYOLO_ERR_RANGE on ONE rank (the stand-in raises its status words in one forward and returns garbage rows, as the device
    does): the words travel in the batch's all-gather, so BOTH ranks re-run THAT global batch in the fp32 mode (round 5: that batch
    only, `Model.run(precision='f32')` on the twin handle), re-run what was in flight behind it in the default precision, stay in
    split-f16 and write exactly the files of an undisturbed run -- no rank is left in another arithmetic, nothing hangs."""
    import json
    _records(tmp_path, 11)
    rc1, o1 = _run_infer(1, tmp_path, str(tmp_path / "one" / "run"))
    assert rc1 == [0], o1
    rc2, o2 = _run_infer(2, tmp_path, str(tmp_path / "two" / "run"), extra=(raise_rank, raise_call))
    assert rc2 == [0, 0], o2
    a, b = str(tmp_path / "one" / "run_0"), str(tmp_path / "two" / "run_0")
    assert sorted(os.listdir(a)) == sorted(os.listdir(b))
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f
    c = [json.load(open(tmp_path / ("calls_w2_r%d.json" % r))) for r in range(2)]
    bad = raise_call                                     # the global batch whose status words came back raised (seed = 3 + batch)
    for r in range(2):
        assert c[r]["precision"] == "split" and c[r]["stats"]["precision"] == "split"
        assert c[r]["stats"]["precision_switches"] == 2 and c[r]["stats"]["fp32_batches"] == [bad]
        assert "precision:f32" not in c[r]["log"] and "finalize:f32" not in c[r]["log"]            # nothing is re-packed in mid-stream
    # both ranks re-ran the SAME batch in fp32 (rank 1's block of the third, short batch is empty: nothing to run there)
    assert [l for l in c[0]["log"] if l.startswith("run:")] == ["run:f32:%d" % (3 + bad)]
    assert [l for l in c[1]["log"] if l.startswith("run:")] == (["run:f32:%d" % (3 + bad)] if bad < 3 else [])
    # ... and the batch in flight behind it once more in the default precision: the seeds of the repeated calls agree
    redo = [sorted({s for (_, _, s) in c[r]["calls"] if [x[2] for x in c[r]["calls"]].count(s) > 1}) for r in range(2)]
    assert redo[0] == [3 + b_ for b_ in range(bad, min(bad + 2, 4))]
    assert redo[1] == [3 + b_ for b_ in range(bad, min(bad + 2, 3))]


@pytest.mark.parametrize("world", [2, 8])
def test_t_sharded_loop_adds_up_over_the_ranks(tmp_path, world):
    """config['shard'] = 'T' on CPU (gloo, the stand-in engine whose per-sample contributions are small integers: their float32 sums
    are exact whatever the cut): every rank reads EVERY frame and runs ITS samples of the T = 3 -- [0,2) + [2,3) at world 2; one sample
    each on ranks 0 - 2 and NOTHING on ranks 3 - 7 at world 8, which still take part in every all-reduce and write their share of
    the files --, one all-reduce per image, image i written by rank i % world: byte-identical JSON to the one-process run."""
    import json
    _records(tmp_path, 7)
    rc1, o1 = _run_infer(1, tmp_path, str(tmp_path / "one" / "run"), extra=(-1, 0, 1))
    assert rc1 == [0], o1
    rcw, ow = _run_infer(world, tmp_path, str(tmp_path / "many" / "run"), extra=(-1, 0, 1), timeout=600)
    assert rcw == [0] * world, ow
    a, b = str(tmp_path / "one" / "run_0"), str(tmp_path / "many" / "run_0")
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == ["f%02d.json" % i for i in range(7)]
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f
    c = [json.load(open(tmp_path / ("calls_w%d_r%d.json" % (world, r)))) for r in range(world)]
    from byolo import dist as bdist
    for r in range(world):
        t0, t1 = bdist.shard_range(3, r, world)
        assert c[r]["stats"]["samples"] == [t0, t1] and c[r]["stats"]["shard"] == "T"
        # every frame, ONE image per call, its position in the batch as first_image, the rank's samples (none: no call at all)
        want = [[1, j, 3 + step, t0, t1] for step, n in ((1, 5), (2, 2)) for j in range(n)] if t1 > t0 else []
        assert c[r]["calls"] == want, (r, c[r]["calls"])
        assert c[r]["stats"]["images"] == len([i for i in range(7) if i % world == r])


def test_t_sharded_loop_reruns_an_image_in_fp32_on_every_rank(tmp_path):
    """... and BYOLO_ERR_RANGE on one rank's shard of one image: the flag rides in the all-reduced buffer, BOTH ranks re-run THAT image
    in fp32 (Model.run(precision='f32')), the files equal the undisturbed run's."""
    import json
    _records(tmp_path, 7)
    rc1, o1 = _run_infer(1, tmp_path, str(tmp_path / "one" / "run"), extra=(-1, 0, 1))
    rc2, o2 = _run_infer(2, tmp_path, str(tmp_path / "two" / "run"), extra=(1, 3, 1))          # rank 1's third forward = image 2 of batch 1
    assert rc1 == [0] and rc2 == [0, 0], (o1, o2)
    a, b = str(tmp_path / "one" / "run_0"), str(tmp_path / "two" / "run_0")
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f
    c = [json.load(open(tmp_path / ("calls_w2_r%d.json" % r))) for r in range(2)]
    for r in range(2):
        assert c[r]["stats"]["precision_switches"] == 2 and c[r]["stats"]["fp32_batches"] == [1] and c[r]["precision"] == "split"
        assert [l for l in c[r]["log"] if l.startswith("run:")] == ["run:f32:4"]
        assert c[r]["calls"].count([1, 2, 4, 0, 2] if r == 0 else [1, 2, 4, 2, 3]) == 2


def test_inference_world2_feed_failure_on_one_rank_stops_both(tmp_path):
    """every rank reads, CRC-checks and decodes only ITS block of a global batch, so a corrupt record raises on one rank
    only -- while the other is already queued in that batch's all-gather.  The failing rank takes part in the collective with a
    'feed failed' status word, the other rank reads it in the gathered buffer and raises too: both processes end within seconds
    instead of one waiting for the backend's timeout."""
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "bayesian-yolov3_amd"))
    from lib_yolo import dataset_utils as du
    _records(tmp_path, 11)
    p = str(tmp_path / "val-00000-of-00001")
    raw = bytearray(open(p, "rb").read())
    recs = list(du._RecordFile(p, True))
    raw[recs[8][1] + 40] ^= 0xFF                       # record 8 = batch 2 (records 5..9), rank 1's block [8, 10) at world 2
    open(p, "wb").write(bytes(raw))
    t0 = time.time()
    rc, outs = _run_infer(2, tmp_path, str(tmp_path / "out" / "run"), timeout=120)
    assert time.time() - t0 < 90
    assert rc[0] != 0 and rc[1] != 0, (rc, outs)
    assert "CRC" in outs[1], outs[1][-1500:]
    assert "rank 1 could not read its records of batch 2" in outs[0], outs[0][-1500:]
    # the batch before the corrupt one was written by both ranks
    assert sorted(os.listdir(tmp_path / "out" / "run_0")) == ["f%02d.json" % i for i in range(5)]


def test_inference_world2_ranks_stop_together_when_the_output_directory_exists(tmp_path):
    """The reference refuses to overwrite an existing run (os.makedirs, inference_epistemic.py:42).  Under torchrun only rank 0
    touches the directory -- and tells the others: every rank exits at once instead of waiting in its first collective for a
    peer that is gone."""
    _records(tmp_path, 3)
    os.makedirs(tmp_path / "out" / "run_0")
    rc, outs = _run_infer(2, tmp_path, str(tmp_path / "out" / "run"), timeout=120)
    assert rc == [7, 7], (rc, outs)
    assert "OSERROR" in outs[0] and "AGREED" in outs[1] and "rank 0 failed" in outs[1]


def test_shard_range_covers_batch():
    from byolo import dist as bdist
    for n in (1, 7, 8, 64):
        for w in (1, 2, 4, 8):
            spans = [bdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))


@pytest.mark.gpu
def test_bench_line_through_rccl_one_rank():
    """bench.py as the driver launches it (torch.distributed.run, one rank per GPU), on the only GPU of the box:
    a forced one-rank RCCL group runs every collective of the N > 1 path (barriers, the all-gather of the packed box
    lists incl. the int32 bit-casts, the MAX all-reduce of the times) and the JSON line keeps its contract."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BYOLO_DIST_FORCE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(os.path.dirname(HERE), "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak" and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_line_two_ranks_with_real_engines_on_one_device():
    """`bench.py --gpus 2` with TWO real engines: the box has one GPU and RCCL refuses two ranks on one device, so both ranks run
    on cuda:0 (BYOLO_DIST_SHARE_DEVICE=1) and the collectives go through gloo (the packed all-gather staged through host memory,
    byolo/dist.py all_gather_flat).  What the line must show: two ranks with DIFFERENT first images of the global batch, the
    global batch = 2 x the per-rank one, one kept-index checksum per image of the gathered list.  (That the N-rank job computes what
    one process computes is tests/test_entry_points.py::test_two_ranks_with_real_engines_write_the_single_process_json -- byte-identical
    files; the benchmark's position-weighted checksums of 1000 kept boxes per image are not comparable across per-call batch sizes:
    other tile / stream-K schedules reorder boxes whose scores differ in the last bit.)"""
    import json
    base = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--steps", "2", "--warmup", "1", "--config", "3",
            "--no-cpu-baseline", "--fp32-steps", "0", "--entry-frames", "0", "--no-other-configs", "--no-profile"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BYOLO_DIST_BACKEND="gloo", BYOLO_DIST_SHARE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + base[1:] + ["--gpus", "2", "--batch", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    two = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and [r["first_image"] for r in two["ranks"]] == [0, 2] and [r["images"] for r in two["ranks"]] == [2, 2]
    assert two["config"]["images_per_gpu"] == 2 and len(two["kept_checksum_per_image_of_one_more_step"]) == 4
    assert two["value"] > 0 and two["scaling"] == "weak" and two["range_status"].startswith("ok")
    # the preflight of the N > 1 path: what the BACKEND says the job is, before anything was built
    assert [r["nccl_world"] for r in two["ranks"]] == [2, 2] and [r["preflight_sum"] for r in two["ranks"]] == [2, 2] and {r["backend"] for r in two["ranks"]} == {"gloo"}
