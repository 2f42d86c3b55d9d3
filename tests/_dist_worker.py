"""Worker for tests/test_dist_cpu.py: world_size-2 gloo run of the N>1 path (sharding + the single
all-gather of the padded box lists)."""
import os
import sys


def main(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "bayesian-yolov3_amd"))
    import torch
    import torch.distributed as dist
    from byolo import dist as bdist
    r, l, w = bdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    Bl, cap, D = 3, 5, 23
    g = torch.Generator().manual_seed(100 + rank)
    rows = torch.rand((Bl, cap, D), generator=g)
    kept = torch.randint(-1, 22743, (Bl, cap), generator=g, dtype=torch.int32)
    count = torch.randint(0, cap, (Bl, 2), generator=g, dtype=torch.int32)
    g_rows, g_kept, g_count = bdist.allgather_boxes(rows, kept, count, world)
    # a short global batch of 3 images over 2 ranks (blocks of 2 and 1): every rank pads its block to
    # padded_block(3, 2) = 2 images, the final box list comes back in GLOBAL image order, trimmed to the kept counts
    n_glob = 3
    lo, hi = bdist.shard_range(n_glob, rank, world)
    bl = bdist.padded_block(n_glob, world)
    p_rows = torch.zeros((bl, cap, D)); p_kept = torch.full((bl, cap), -1, dtype=torch.int32)
    p_count = torch.zeros((bl, 2), dtype=torch.int32)
    for j in range(hi - lo):
        k = 1 + (lo + j) % cap                       # image g keeps 1 + g boxes, all valued 100 * g + column index
        p_rows[j, :k] = 100.0 * (lo + j) + torch.arange(D, dtype=torch.float32)
        p_kept[j, :k] = torch.arange(k, dtype=torch.int32) + 1000 * (lo + j)
        p_count[j] = k
    u_rows, u_kept = bdist.unpack_global(*bdist.allgather_boxes(p_rows, p_kept, p_count, world), n_glob, world)
    # a step any rank may fail at (byolo/inference.py: building the fp32 twin handle): rank 1 fails, BOTH ranks get an exception back;
    # nobody fails, nobody gets one
    e_none = bdist.agree_on_any_error(None)
    e_any = bdist.agree_on_any_error(MemoryError("out of device memory") if rank == world - 1 else None)
    torch.save({"rows": rows, "kept": kept, "count": count, "g_rows": g_rows, "g_kept": g_kept, "g_count": g_count,
                "shard": bdist.shard_range(7, rank, world), "u_rows": u_rows, "u_kept": u_kept,
                "agree": (e_none is None, type(e_any).__name__, str(e_any))},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
