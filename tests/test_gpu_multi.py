"""-m gpu tests of the multi-GPU path on the ONE GPU the test box has: two ranks (one process each) are started on the
same device, initialise the collective backend (`nccl` == RCCL first), meet at the barriers bench.py uses, all-reduce a
report value, and every rank's images / gradients must equal the single-process result of the same scene.  The raster
path has no collective (one scene per GPU), so sharing a device changes nothing but the speed.

RCCL refuses two ranks on one physical GPU ("Duplicate GPU detected") on some builds; then the RCCL leg is reported as
unavailable-on-this-box and the same two-rank test runs on `gloo` (host-side collectives, device-side rasterizer).  The
8-GPU RCCL run itself is the driver's (SCALE_rNN.json)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene(rank):
    from gscream_amd import synthetic as S
    s = S.scene_config1(seed=40 + rank, P=3000, W=160, H=96)
    return s, S.upstream_grads(40 + rank, s["W"], s["H"])


def _worker(rank, world, port, backend, outdir):
    import helpers as Hh
    from gscream_amd import multi
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", multi.pick_device(rank, oversubscribe=True))
    torch.cuda.set_device(dev)
    status = {"rank": rank, "backend": backend, "ok": False}
    try:
        dist = multi.init(backend, dev)
        assert dist is not None and dist.get_world_size() == world
        multi.barrier(dist, dev)
        s, grads = _scene(rank)
        got = Hh.hip_run(s, grads, device=dev)
        multi.barrier(dist, dev)
        # the reduction bench.py prints: units of all ranks / slowest rank
        total, tmax, rate = multi.aggregate_throughput(dist, 10 * (rank + 1), 1.0 + rank, dev)
        # and a device-side all-reduce of something every rank computed on its own scene
        v = torch.tensor([float(got["radii"].sum())], dtype=torch.float64, device=None if backend == "gloo" else dev)
        dist.all_reduce(v)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **{k: got[k] for k in got})
        status.update(ok=True, total=total, tmax=tmax, rate=rate, radii_sum=float(v.item()),
                      pulled=[i for i in iter(multi.SceneQueue(dist, 5).pull, None)])
        multi.barrier(dist, dev)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        status["error"] = repr(e)
    json.dump(status, open(os.path.join(outdir, f"status{rank}.json"), "w"))


def _run_two_ranks(backend, outdir):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, str(outdir))) for r in range(world)]
    [p.start() for p in procs]
    [p.join(240) for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    out = []
    for r in range(world):
        f = os.path.join(outdir, f"status{r}.json")
        out.append(json.load(open(f)) if os.path.exists(f) else {"rank": r, "ok": False, "error": f"exit code {procs[r].exitcode}"})
    return out


def _check(status, outdir):
    import helpers as Hh
    assert all(s["ok"] for s in status), status
    for s in status:
        assert s["total"] == 30.0 and s["tmax"] == 2.0 and abs(s["rate"] - 15.0) < 1e-9
    # the queue handed every index out exactly once across the two ranks
    assert sorted(status[0]["pulled"] + status[1]["pulled"]) == list(range(5))
    expect_sum = 0.0
    for r in range(2):
        s, grads = _scene(r)
        ref = Hh.hip_run(s, grads)  # single process, same scene
        got = np.load(os.path.join(outdir, f"rank{r}.npz"))
        assert (got["radii"] == ref["radii"]).all()
        for k in ("out_color", "out_depth", "out_unc"):
            assert np.array_equal(got[k], ref[k]), f"rank {r} {k}: the forward is bit-reproducible"
        Hh.assert_grads_nearly_equal({k: got[k] for k in got.files}, ref, context=f"rank {r}")
        expect_sum += float(ref["radii"].sum())
    assert status[0]["radii_sum"] == status[1]["radii_sum"] == expect_sum


def test_two_ranks_one_gpu_collectives_and_images(native_lib, tmp_path):
    d = tmp_path / "nccl"
    d.mkdir()
    status = _run_two_ranks("nccl", d)
    if all(s["ok"] for s in status):
        _check(status, d)
        print("two ranks on one GPU over RCCL (nccl backend): OK")
        return
    err = " | ".join(s.get("error", "") for s in status)
    print("RCCL refused two ranks on one GPU:", err)
    assert "uplicate" in err or "invalid usage" in err.lower() or "ncclInvalidUsage" in err, \
        f"nccl backend failed for another reason than the duplicate-GPU check: {err}"
    d = tmp_path / "gloo"
    d.mkdir()
    status = _run_two_ranks("gloo", d)
    _check(status, d)
    pytest.xfail("RCCL rejects two ranks on the single GPU of this box (duplicate-GPU check); the same two-rank test "
                 "passed on gloo.  The N-GPU RCCL run is the driver's scaling bench.")


def _rccl_world1_worker(port, outdir):
    from gscream_amd import multi
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    status = {"ok": False}
    try:
        dev = torch.device("cuda", multi.pick_device(0))
        torch.cuda.set_device(dev)
        assert multi.init("nccl", dev) is None, "a world of one needs no process group unless forced"
        dist = multi.init("nccl", dev, force=True)  # communicator creation goes through librccl
        assert dist is not None and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        multi.barrier(dist, dev)                     # ncclAllReduce under the hood + device sync
        total, tmax, rate = multi.aggregate_throughput(dist, 40, 2.0, dev)  # the two reductions of bench.py's report, on the device
        x = torch.arange(1 << 20, dtype=torch.float32, device=dev)
        y = x.clone()
        dist.all_reduce(y)                           # a real RCCL kernel on a megabyte-sized device buffer
        torch.cuda.synchronize(dev)
        q = multi.SceneQueue(dist, 3)                # TCPStore counter of the process group
        pulled = [i for i in iter(q.pull, None)]
        gathered = multi.gather_objects(dist, {"rank": 0, "pulled": pulled})
        status.update(ok=True, total=total, tmax=tmax, rate=rate, same=bool(torch.equal(x, y)), pulled=pulled, gathered=gathered,
                      backend=dist.get_backend(), nccl_version=list(torch.cuda.nccl.version()))
        multi.barrier(dist, dev)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        status["error"] = repr(e)
    json.dump(status, open(os.path.join(outdir, "status.json"), "w"))


def test_rccl_comes_up_with_a_world_of_one(native_lib, tmp_path):
    """The only RCCL evidence a 1-GPU box can give: the `nccl` backend (== RCCL on ROCm) initialises, and the barrier,
    the report reductions, a device all-reduce, the work queue and the report gather all run on it."""
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), str(tmp_path)))
    p.start()
    p.join(240)
    if p.is_alive():
        p.kill()
        pytest.fail("RCCL world-size-1 bring-up hung")
    f = tmp_path / "status.json"
    assert f.exists(), f"worker died with exit code {p.exitcode}"
    st = json.load(open(f))
    assert st["ok"], st
    assert st["backend"] == "nccl" and (st["total"], st["tmax"], st["rate"]) == (40.0, 2.0, 20.0) and st["same"]
    assert st["pulled"] == [0, 1, 2] and st["gathered"] == [{"rank": 0, "pulled": [0, 1, 2]}]
    print("RCCL (nccl backend) world size 1: OK, version", st["nccl_version"])


def _bench(*extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--no-next-rows", *extra]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    return p


def test_bench_gpus_flag_launches_the_ranks_itself(native_lib):
    # one GPU here: --gpus 2 without the test-mode flag must refuse loudly, not silently measure one GPU
    p = _bench("--gpus", "2")
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0 and "one process per GPU" in (p.stderr + p.stdout)
    # with the test-mode flag both ranks share the device; the line must say n_gpus 2 and a world of 2
    p = _bench("--gpus", "2", "--oversubscribe", "--backend", "gloo")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["collective_world_size"] == 2 and line["scaling"] == "weak"
    assert line["config"]["oversubscribed_test_mode"] is True
    # round 5: every rank's own figures travel on the line (gathered with multi.gather_objects)
    pr = line["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["iters_per_s"] > 0 and r["num_rendered"] > 0 and r["host_ms"] > 0 for r in pr)
    assert all("numa" in r and "pinned" in r["numa"] for r in pr)
    assert abs(line["value"] - 2 * min(r["iters_per_s"] for r in pr)) <= 0.02 * line["value"], "value = all steps / the slowest rank"
    one = json.loads([l for l in _bench("--gpus", "1").stdout.splitlines() if l.startswith("{")][-1])
    assert one["n_gpus"] == 1 and one["value"] > 0


def test_config5_work_queue_runs_every_scene_once(native_lib):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config5", "--steps", "2", "--warmup", "1",
           "--gpus", "2", "--oversubscribe", "--backend", "gloo"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=290)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert [s["scene"] for s in line["scenes"]] == list(range(10)) and line["scaling"] == "strong"
    assert {s["rank"] for s in line["scenes"]} == {0, 1}, "both ranks pulled work"
    assert all(600_000 <= s["P"] <= 1_400_000 for s in line["scenes"])
    assert line["checks_passed"] == 10 and all(s["check_ok"] for s in line["scenes"]), "every scene's outputs are checked on its rank"
    assert line["wall_clock"]["seconds_first_to_last_barrier"] >= line["slowest_rank_busy_s"]
    assert sorted(sum((r["scenes"] for r in line["per_rank"]), [])) == list(range(10)) and all(r["build_all_scenes_s"] > 0 for r in line["per_rank"])
