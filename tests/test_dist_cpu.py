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
    """TestingDataset.iter_shards: batch_size is the GLOBAL batch (as in the reference); rank r gets the contiguous
    block shard_range(len(batch), r, world) of every global batch, its offset `lo` (-> first_image: the dropout stream
    position) and ALL file names; the blocks of the ranks tile the batch, also the last short one (blocks may be empty)."""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "bayesian-yolov3_amd"))
    from lib_yolo import dataset_utils as du
    from byolo import dist as bdist
    imgs = _records(tmp_path, 11)
    cfg = {"batch_size": 4, "full_img_size": [32, 32, 3], "data": {"file_pattern": str(tmp_path / "val-*")}}
    per_rank = [list(du.TestingDataset(cfg).iter_shards(r, world)) for r in range(world)]
    plain = list(du.TestingDataset(cfg))
    assert [len(n) for _, n in plain] == [4, 4, 3]
    for step in range(3):
        names = per_rank[0][step][1]
        assert names == plain[step][1] == ["f%02d.png" % (4 * step + i) for i in range(len(names))]
        pos = 0
        for r in range(world):
            x, n_r, lo = per_rank[r][step]
            assert n_r == names and lo == pos == bdist.shard_range(len(names), r, world)[0]
            assert x.dtype == np.float32 and x.shape[1:] == (32, 32, 3)
            for j in range(x.shape[0]):
                assert np.array_equal(x[j], imgs[4 * step + lo + j])
            pos += x.shape[0]
        assert pos == len(names)


def _run_infer(world, data_dir, out_path, timeout=300):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_infer_worker.py"), str(r), str(world), str(port), str(data_dir), out_path],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    return [p.returncode for p in procs], outs


def test_inference_epistemic_world2_writes_the_single_process_json(tmp_path):
    """`torchrun --nproc-per-node 2 inference_epistemic.py` end to end up to the engine boundary, on CPU (gloo): the driver loop
    of the entry point with a stand-in engine whose rows depend on the pixels and on the image's position in the GLOBAL batch.
    11 images, global batch 5 (so the last batch is short and rank 1's block of it has ONE image): the two-process job must write
    byte-identical ECP JSON files to the one-process run, rank 1 writes nothing, every rank's engine option `device` is its
    LOCAL_RANK, and the blocks / first_image values tile every global batch (inference_epistemic.py:56-83 + byolo/dist.py)."""
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
    one = json.load(open(tmp_path / "calls_w1_r0.json"))["calls"]
    assert one == [[5, 0, 4], [5, 0, 5], [1, 0, 6]]


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
