"""Multi-GPU: one process per GPU (torchrun), images sharded on the batch axis, weights replicated,
no collective inside the network; ONE all-gather (RCCL over xGMI; `gloo` in CPU tests) of the
fixed-size padded NMS outputs assembles the final box list on every rank.

The reference has no counterpart: it pins every session to one device
(`inference_epistemic.py:57`: tf.ConfigProto(device_count={'GPU': 1})).
"""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, force=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1 unless `force`, or
    BYOLO_DIST_FORCE=1, asks for a one-rank group: that is how the RCCL calls of the N > 1 path are exercised on a
    one-GPU box).  Returns (rank, local_rank, world)."""
    import torch
    import torch.distributed as dist
    rank, local, world = env_rank()
    if force is None:
        force = os.environ.get("BYOLO_DIST_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" IS RCCL on ROCm.  BYOLO_DIST_BACKEND=gloo: the N > 1 path with REAL engines on a box with fewer GPUs than
            # ranks (RCCL refuses two ranks on one device) -- the collectives then go through host memory (host_staged())
            backend = os.environ.get("BYOLO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def local_device(local_rank):
    """The GPU of this process: LOCAL_RANK, as torchrun numbers them -- unless BYOLO_DIST_SHARE_DEVICE=1 puts every rank on
    cuda:0 (tests and bench.py --gpus 2 on a one-GPU box, together with BYOLO_DIST_BACKEND=gloo)."""
    return 0 if os.environ.get("BYOLO_DIST_SHARE_DEVICE", "0") == "1" else local_rank


def host_staged(tensor):
    """True if a collective on `tensor` has to go through host memory: a device tensor under the gloo backend."""
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend() == "gloo" and tensor.is_cuda


def all_gather_flat(recv, send):
    """dist.all_gather_into_tensor(recv, send) for flat float32 buffers; under gloo with device tensors the exchange is staged
    through host memory (this WAITS for the current stream -- a test / small-box mode, never the RCCL path)."""
    import torch
    import torch.distributed as dist
    if not host_staged(send):
        dist.all_gather_into_tensor(recv, send)
        return
    h_send = send.cpu()                                   # synchronises with the current stream
    h_recv = torch.empty(recv.numel(), dtype=recv.dtype)
    dist.all_gather_into_tensor(h_recv, h_send)
    recv.copy_(h_recv)


def all_reduce_flat(buf):
    """dist.all_reduce(buf, SUM) in place for a flat float32 buffer -- the ONE collective per image of the T-sharded path
    (byolo/inference.py _run_t_sharded); staged through host memory under gloo with device tensors, like all_gather_flat."""
    import torch.distributed as dist
    if not host_staged(buf):
        dist.all_reduce(buf)
        return
    h = buf.cpu()
    dist.all_reduce(h)
    buf.copy_(h)


def agree_on_error(err, src=0):
    """Rank `src` passes an exception (or None); every rank gets it back -- so that all ranks of a job raise together
    instead of the healthy ones blocking in their next collective until the backend's timeout.  No-op without a
    process group."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return err
    box = [repr(err) if err is not None else None]
    dist.broadcast_object_list(box, src=src)
    if box[0] is None:
        return None
    return err if err is not None else RuntimeError("rank %d failed: %s" % (src, box[0]))


def agree_on_any_error(err):
    """Every rank passes an exception (or None); if ANY rank has one, every rank gets one back (its own, or a RuntimeError naming the
    first failing rank) -- for a step every rank takes at the same point of the job and any of them may fail at (building the fp32
    twin handle: a second weight pack and workspace on a nearly full GPU).  No-op without a process group."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return err
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, repr(err) if err is not None else None)
    bad = [(r, m) for r, m in enumerate(box) if m is not None]
    if not bad:
        return None
    return err if err is not None else RuntimeError("rank %d failed: %s" % bad[0])


def shard_range(n_items, rank, world):
    """Contiguous block of the batch axis owned by `rank` (first ranks get the remainder)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allgather_boxes(rows, kept, count, world=None):
    """rows [Bl,cap,D] f32, kept [Bl,cap] i32, count [Bl,2] i32  ->  the same with Bl*world images,
    rank-major (rank r's images at [r*Bl, (r+1)*Bl)).  Every rank must pass the same Bl.
    The three tensors are packed into one int32/float32-agnostic byte buffer so that exactly ONE
    collective is issued per batch (<= 0.74 MB per rank at BASELINE config 4)."""
    import torch
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not dist.is_initialized():
        return rows, kept, count
    Bl, cap, D = rows.shape
    n_r, n_k, n_c = rows.numel(), kept.numel(), count.numel()
    send = torch.empty(n_r + n_k + n_c, dtype=torch.float32, device=rows.device)
    send[:n_r] = rows.reshape(-1)
    send[n_r:n_r + n_k] = kept.reshape(-1).view(torch.float32)          # bit-cast, no conversion
    send[n_r + n_k:] = count.reshape(-1).view(torch.float32)
    recv = torch.empty(world * send.numel(), dtype=torch.float32, device=rows.device)
    all_gather_flat(recv, send)
    recv = recv.view(world, -1)
    g_rows = recv[:, :n_r].reshape(world * Bl, cap, D)
    g_kept = recv[:, n_r:n_r + n_k].contiguous().view(torch.int32).reshape(world * Bl, cap)
    g_count = recv[:, n_r + n_k:].contiguous().view(torch.int32).reshape(world * Bl, 2)
    return g_rows, g_kept, g_count


def padded_block(n_global, world):
    """Images per rank in the gathered buffer: every rank contributes the same number (the collective needs equal
    shapes), ranks whose block of the global batch is shorter pad with empty images (count 0)."""
    return (n_global + world - 1) // world


def unpack_global(g_rows, g_kept, g_count, n_global, world):
    """The gathered, rank-major, padded buffers of allgather_boxes -> ([rows_0, ..., rows_{n_global-1}], [kept_0, ...])
    in GLOBAL image order, image g trimmed to its kept count: the final box list of the batch.  Reads the counts on
    the host (one small copy): call it on the rank that consumes the list (rank 0 of the entry points)."""
    bl = padded_block(n_global, world)
    assert g_rows.shape[0] == world * bl, "every rank pads its block to padded_block(n_global, world) images"
    counts = g_count[:, 0].cpu().tolist()
    out_rows, out_kept = [], []
    for r in range(world):
        lo, hi = shard_range(n_global, r, world)
        for j in range(hi - lo):
            out_rows.append(g_rows[r * bl + j, :counts[r * bl + j]])
            out_kept.append(g_kept[r * bl + j, :counts[r * bl + j]])
    return out_rows, out_kept
