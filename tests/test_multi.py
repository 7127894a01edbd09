"""world_size-2 `gloo` tests of the multi-GPU plumbing (CPU): scene sharding and the throughput reduction that
bench.py prints.  The raster path itself has no collective (one scene per GPU)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscream_amd import multi  # noqa: E402


def test_scene_assignment_covers_every_scene_once():
    shares = multi.assign_scenes(10, 8)  # BASELINE.json config 5
    assert len(shares) == 8 and sorted(sum(shares, [])) == list(range(10))
    assert [len(s) for s in shares] == [2, 2, 1, 1, 1, 1, 1, 1]
    assert multi.assign_scenes(3, 1) == [[0, 1, 2]] and multi.assign_scenes(1, 4) == [[0], [], [], []]
    assert multi.scene_seed(1, 0, 1) == 1 and len({multi.scene_seed(1, r, 8) for r in range(8)}) == 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = multi.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    multi.barrier(dist)
    # rank r "rasterizes" its share of 5 scenes, 10 steps each, and is slower the higher its rank
    my_scenes = multi.assign_scenes(5, world)[rank]
    units, elapsed = 10 * len(my_scenes), 1.0 + 0.5 * rank
    total, tmax, rate = multi.aggregate_throughput(dist, units, elapsed)
    multi.barrier(dist)
    pulled = [i for i in iter(multi.SceneQueue(dist, 10).pull, None)]  # config 5: shared work queue
    multi.barrier(dist)
    # a SECOND queue in the same process group starts at zero again (its own counter key), and a local queue in between
    # (dist=None) does not shift the key the ranks agree on
    if rank == 0:
        multi.SceneQueue(None, 2).pull()
    pulled2 = [i for i in iter(multi.SceneQueue(dist, 4).pull, None)]
    multi.barrier(dist)
    out[rank] = (my_scenes, total, tmax, rate, multi.scene_seed(7, rank, world), pulled, multi.gather_objects(dist, pulled), pulled2)
    dist.destroy_process_group()


def test_two_rank_gloo_throughput_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out[0][0] == [0, 2, 4] and out[1][0] == [1, 3]
    for r in range(world):
        _, total, tmax, rate, seed = out[r][:5]
        assert total == 50.0 and tmax == 1.5 and abs(rate - 50.0 / 1.5) < 1e-9  # all units / slowest rank
    assert out[0][4] != out[1][4]
    # the work queue hands every one of the ten scenes to exactly one rank, and the report gather sees both shares
    assert sorted(out[0][5] + out[1][5]) == list(range(10))
    assert out[0][6] == out[1][6] == [out[0][5], out[1][5]]
    assert sorted(out[0][7] + out[1][7]) == list(range(4)), "second queue of the same group hands out 0..3 again"


def test_single_process_needs_no_process_group():
    os.environ.pop("WORLD_SIZE", None)
    assert multi.init("gloo") is None
    total, tmax, rate = multi.aggregate_throughput(None, 20, 0.5)
    assert (total, tmax, rate) == (20.0, 0.5, 40.0)


def test_scene_queue_single_process_and_config5_scene_list():
    q = multi.SceneQueue(None, 3)
    assert [q.pull(), q.pull(), q.pull(), q.pull(), q.pull()] == [0, 1, 2, None, None]
    scenes = [multi.config5_scene(i) for i in range(10)]
    assert [s for s, _ in scenes] == list(range(10, 20))          # seeds 10..19
    assert all(600_000 <= p <= 1_400_000 for _, p in scenes) and len({p for _, p in scenes}) > 5
    assert scenes == [multi.config5_scene(i) for i in range(10)]  # fixed per seed


def test_pick_device_refuses_more_ranks_than_gpus(monkeypatch):
    import pytest
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert multi.pick_device(1) == 1 and multi.pick_device(3, oversubscribe=True) == 1
    with pytest.raises(RuntimeError, match="one process per GPU"):
        multi.pick_device(2)


def test_bench_refuses_gpus_without_devices():
    """`bench.py --gpus 8` on a box without 8 GPUs must fail loudly instead of measuring fewer GPUs."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # at most one device visible to the subprocess: the refusal must not depend on how many GPUs the box has
    # (on the 8-GPU target node `--gpus 8` would otherwise launch a real 8-rank bench)
    env.update(HIP_VISIBLE_DEVICES="0", CUDA_VISIBLE_DEVICES="0", ROCR_VISIBLE_DEVICES="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "one process per GPU" in (p.stderr + p.stdout)
    # and a launcher/flag mismatch is an error too
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)
