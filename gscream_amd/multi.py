"""Multi-GPU plumbing for the one place the path shards: across SCENES.

GScream trains each SPIn-NeRF scene in its own process, sequentially on GPU 0 (reference: scripts/run.py:14-80,
no distributed call is ever executed).  Scenes are independent, so the MI355X build runs one process per GPU with
one scene (or a static share of the scene list) each.  Nothing inside the raster path crosses xGMI: the only
collectives are barriers around the timed region and the reductions of the timing report below.  With the `nccl`
backend these are RCCL calls; the same code runs on `gloo` for the CPU tests (tests/test_multi.py).
"""
import os

import torch


def dist_env():
    """(rank, local_rank, world_size) from the torchrun / torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def init(backend, device=None):
    """Initialises the default process group when WORLD_SIZE > 1; returns torch.distributed or None."""
    rank, _, world = dist_env()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def assign_scenes(num_scenes, world_size):
    """Static round-robin share of the scene list per rank: BASELINE.json config 5 (10 scenes on 8 GPUs) gives
    ranks 0 and 1 two scenes each.  Every scene appears exactly once."""
    return [list(range(r, num_scenes, world_size)) for r in range(world_size)]


def scene_seed(base_seed, rank, world_size):
    """Each rank rasterizes its own synthetic scene; a single process keeps the workload's canonical seed."""
    return base_seed if world_size <= 1 else base_seed + 10 * rank


def barrier(dist, device=None):
    if dist is not None:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def aggregate_throughput(dist, units_done, elapsed_s, device=None):
    """Whole-job throughput = units processed by ALL ranks / the SLOWEST rank's time (bench.py contract).
    Returns (total_units, max_elapsed, units_per_second) on every rank."""
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_done)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item()), float(u.item()) / float(t.item())
