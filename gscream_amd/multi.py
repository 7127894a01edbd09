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


def init(backend, device=None, force=False):
    """Initialises the default process group when WORLD_SIZE > 1; returns torch.distributed or None.
    force=True brings the backend up for a world of ONE as well (the RCCL bring-up test on a 1-GPU box: communicator
    creation, barrier and all-reduce all run through librccl; bench.py never forces it)."""
    rank, _, world = dist_env()
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def launch_ranks(script, argv, n_gpus, python=None):
    """`bench.py --gpus N` started WITHOUT torchrun: re-execute the script as N ranks of one node
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), one process per
    GPU, and return the launcher's exit code.  Rank 0 prints the JSON line; stdout/stderr pass straight through."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    return subprocess.call(cmd, env=env)


def pick_device(local_rank, oversubscribe=False):
    """The GPU of this rank: one rank per GPU.  More ranks than GPUs is an error unless `oversubscribe` (a test mode
    for boxes with a single GPU: ranks share devices round-robin; never a measurement)."""
    n = torch.cuda.device_count()
    if local_rank < n:
        return local_rank
    if not oversubscribe:
        raise RuntimeError(f"rank with LOCAL_RANK={local_rank} has no GPU of its own: only {n} visible "
                           "(one process per GPU; pass --oversubscribe only to test the launch path)")
    return local_rank % max(n, 1)


def pin_to_gpu_numa(device_index):
    """Keep this rank's host threads on the NUMA node its GPU hangs off (one process per GPU: the launch path -- Python, ctypes, the
    pinned read-back word -- then never crosses the socket interconnect).  Best effort and Linux only: the node comes from the
    device's PCI address in sysfs, the CPU list from /sys/devices/system/node; without either (containers often hide them, or report
    node -1) nothing is changed.  Returns what was done, for the bench line."""
    info = {"pinned": False, "numa_node": None, "cpus": None, "why": None}
    try:
        if not hasattr(os, "sched_setaffinity"):
            info["why"] = "os.sched_setaffinity not available"
            return info
        bdf = None
        try:
            from torch.cuda import get_device_properties
            pr = get_device_properties(device_index)
            dom, bus, dv = getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None)
            if bus is not None and dv is not None:
                bdf = f"{int(dom):04x}:{int(bus):02x}:{int(dv):02x}.0"
        except Exception as e:  # noqa: BLE001
            info["why"] = f"no PCI address for device {device_index}: {e!r}"
        node = None
        if bdf is not None:
            path = f"/sys/bus/pci/devices/{bdf}/numa_node"
            if os.path.exists(path):
                node = int(open(path).read().strip())
            else:
                info["why"] = f"{path} not present"
        if node is None or node < 0:
            info["why"] = info["why"] or f"sysfs reports numa_node {node} for {bdf}"
            return info
        cl = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
        cpus = set()
        for part in cl.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            info["why"] = f"node {node} has no CPU this process may run on"
            return info
        os.sched_setaffinity(0, allowed)
        info.update(pinned=True, numa_node=node, cpus=len(allowed), why=f"GPU {device_index} ({bdf}) is on NUMA node {node}")
    except Exception as e:  # noqa: BLE001
        info["why"] = repr(e)
    return info


class SceneQueue:
    """Shared work queue for more scenes than GPUs (BASELINE.json config 5: ten SPIn-NeRF scenes on eight GPUs; the
    reference trains them one after the other, scripts/run.py:14-80).  A rank that finishes a scene pulls the next
    index, so the two left-over scenes go to whichever GPUs are free first.  The queue head is one atomic counter in
    the process group's TCPStore (host side, rank 0 serves it): no collective, nothing on xGMI.  Single process: a
    local counter."""

    _generation = 0  # queues are created collectively, in the same order on every rank: the n-th queue of a process
    #                  group gets its own counter key, so a second queue does not start at the first one's count

    def __init__(self, dist, num_scenes, name="gsr_scene_queue"):
        self.num_scenes, self._local, self.name, self.store = int(num_scenes), 0, name, None
        if dist is not None:
            SceneQueue._generation += 1
            self.name = f"{name}/{SceneQueue._generation}"
            self.store = _default_store(dist)

    def pull(self):
        """-> next scene index, or None when the list is exhausted.  Every index is handed out exactly once."""
        if self.store is None:
            i, self._local = self._local, self._local + 1
        else:
            i = int(self.store.add(self.name, 1)) - 1
        return i if i < self.num_scenes else None


def _default_store(dist):
    """The process group's rendezvous store (TCPStore served by rank 0).  torch exposes it only under a private name."""
    get = getattr(dist.distributed_c10d, "_get_default_store", None)
    if get is None:
        raise RuntimeError("this torch build does not expose the default process-group store; SceneQueue needs it")
    return get()


def config5_scene(index):
    """(seed, P) of scene `index` of config 5: seeds 10..19 of the config-2 generator,
    P in [0.6, 1.4] x 10^6, fixed per seed."""
    import numpy as np
    seed = 10 + int(index)
    frac = np.random.default_rng(1000 + seed).uniform(0.6, 1.4)
    return seed, int(round(frac * 1_000_000 / 1000.0)) * 1000


def gather_objects(dist, obj):
    """Every rank's small report object on every rank (host side; gloo or RCCL's object path)."""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


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
    if dist is not None and dist.get_backend() == "gloo":
        device = None  # gloo reduces host tensors
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_done)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item()), float(u.item()) / float(t.item())
