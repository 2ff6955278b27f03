"""One-process-per-GPU plumbing for bench.py and multi-image jobs (torch.distributed;
backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).

The path shards by independent objects (image pairs): rank r of W solves the objects
r, r+W, r+2W, ... with no data-path collective; the only collectives are the barrier and
the max-over-ranks of the wall time that the measurement contract asks for."""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Returns (rank, local_rank, world, dist_or_None).  Rendezvous on 127.0.0.1 unless
    MASTER_ADDR says otherwise."""
    rank, local_rank, world = env_rank()
    if world <= 1 and not os.environ.get("STEREO_AMD_FORCE_DIST"):  # (forcing: exercise RCCL on a 1-GPU box)
        return rank, local_rank, world, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return rank, local_rank, world, dist


def shard(n_objects, rank, world):
    """Object indices owned by `rank` (round robin)."""
    return list(range(rank, n_objects, world))


def barrier(dist, device=None):
    if device is not None and str(device).startswith("cuda"):
        import torch
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if device is not None and str(device).startswith("cuda"):
        import torch
        torch.cuda.synchronize()


def max_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def throughput(dist, units_local, seconds_local, device="cpu"):
    """Whole-job rate: all ranks' units / the slowest rank's time."""
    total = sum_over_ranks(dist, units_local, device)
    t = max_over_ranks(dist, seconds_local, device)
    return total / t, t
