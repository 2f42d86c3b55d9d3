"""N > 1 path on CPU: two processes, gloo backend (rendezvous on 127.0.0.1)."""
import os
import socket
import subprocess
import sys

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


def test_shard_range_covers_batch():
    from byolo import dist as bdist
    for n in (1, 7, 8, 64):
        for w in (1, 2, 4, 8):
            spans = [bdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))


import pytest


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
