"""Multi-GPU rendering: one process per GPU, the scene replicated, samples-per-pixel partitioned across ranks and the
per-rank float films summed with ONE reduce (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

The path has no other exchange step: every sample is independent and each contribution is already weighted by
1/spp_total, so the final film is the plain sum of the rank films.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torch.distributed.run.
    Returns (rank, world_size, local_rank); (0, 1, 0) when not launched that way."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def spp_share(spp_total, world_size, rank):
    begin = spp_total * rank // world_size
    end = spp_total * (rank + 1) // world_size
    return begin, end - begin


def render_and_reduce(render_fn, film, spp_total, seed=0, dst=0):
    """render_fn(spp_begin, spp_count, spp_total, seed) accumulates this rank's share into `film` (a torch tensor the
    renderer writes through its data pointer, or returns nothing for in-place renderers).  Afterwards the films of all
    ranks are summed onto rank `dst`.  Returns whatever render_fn returned (its statistics)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    begin, count = spp_share(spp_total, world, rank)
    stats = render_fn(begin, count, spp_total, seed)
    if world > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return stats
