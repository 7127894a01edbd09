"""Where does the host time of one fwd+bwd step go?  (GPU box.)

A P = 1k scene makes the kernels negligible, so the wall time of a step loop IS the host path: Python wrapper,
ctypes marshalling, allocator, autograd engine, HIP launches and the wait for num_rendered.  Prints
  * ms per step for P = 1k and for a few larger scenes (the step time must scale with P once the host is off the path),
  * micro-costs of the primitives the wrapper is built from,
  * a cProfile of the step loop (top entries by cumulative time).
usage: python tools/host_profile.py [--profile]
"""
import cProfile
import ctypes
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench as B  # noqa: E402
from gscream_amd import _native  # noqa: E402


def loop_ms(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_host / n * 1e3, (time.perf_counter() - t0) / n * 1e3


def micro(name, fn, n=20000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print(f"  {name:58s} {(time.perf_counter() - t0) / n * 1e6:8.2f} us")


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _native.load()
    B.gpu_spin_up(dev)
    print("== step loop, ms per step (host-enqueue time / wall incl. final sync) ==")
    for P in (1000, 100_000, 400_000, 676_000, 1_000_000):
        sb = B.SceneBench(dev, P, 1008, 567, 1, 1, (True, False, False))
        for _ in range(20):
            sb.step()
        n = 2000 if P <= 100_000 else 300
        h, w = loop_ms(sb.step, n)
        # GPU time of the same steps, stage by stage (HIP events on the launch stream): is the step host- or GPU-bound?
        _native.profile_begin()
        for _ in range(30):
            sb.step()
        torch.cuda.synchronize()
        prof = _native.profile_end()
        stages = {k: v[0] / max(v[1], 1) * 1e3 for k, v in prof.items() if v[1]}
        from gscream_amd import rasterizer as RZ
        print(f"  P={P:8d}: host {h:.4f}  wall {w:.4f}  -> {1e3 / w:.0f} it/s | kernel sum {sum(stages.values()) / 1e3:.4f} ms "
              f"R={RZ._last_stage1['num_rendered']} longest={RZ._last_stage1['max_tile_count']} | "
              + " ".join(f"{k.split('_')[0][:5]}{k.split('_')[-1][:3]}={v:.0f}" for k, v in stages.items()))
        if P == 1000:
            small = sb
    print("== primitives ==")
    x = torch.empty(1000, device=dev)
    e = torch.Tensor([])
    micro("torch.empty((3,567,1008)) cuda", lambda: torch.empty((3, 567, 1008), dtype=torch.float32, device=dev))
    micro("torch.empty(1000) uint8 cuda", lambda: torch.empty((1000,), dtype=torch.uint8, device=dev))
    micro("torch.Tensor([])", lambda: torch.Tensor([]))
    micro("x.contiguous()", lambda: x.contiguous())
    micro("x.is_contiguous()", lambda: x.is_contiguous())
    micro("x.data_ptr()", lambda: x.data_ptr())
    micro("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
    micro("torch.cuda.current_device()", lambda: torch.cuda.current_device())

    def ctx():
        with torch.cuda.device(dev):
            pass
    micro("with torch.cuda.device(dev)", ctx)
    micro("ctypes call gsr_version()", lambda: lib.gsr_version())
    micro("ctypes call gsr_geom_bytes(P)", lambda: lib.gsr_geom_bytes(1000))
    micro("_native.ptr(x)", lambda: _native.ptr(x))
    pres = torch.zeros(1000, dtype=torch.bool, device=dev)
    view = small.rs.viewmatrix
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pp, pv, ppr = _native.ptr(small.leaves[0]), _native.ptr(view), _native.ptr(pres)

    def launch():
        lib.gsr_mark_visible(1000, pp, pv, pv, ppr, st)
    micro("one tiny kernel launch through ctypes (gsr_mark_visible)", launch, 5000)
    torch.cuda.synchronize()
    micro("nn.Module construct GaussianRasterizer", lambda: type(small.rast)(small.rs))

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b, c, d, e_, f, g, h, i, j):
            ctx.save_for_backward(a, b)
            return a.detach(), b.detach()

        @staticmethod
        def backward(ctx, ga, gb):
            return (ga, gb, None, None, None, None, None, None, None, None)
    a = torch.zeros(10, device=dev, requires_grad=True)
    b = torch.zeros(10, device=dev, requires_grad=True)
    ga = torch.zeros(10, device=dev)

    def af():
        o = F.apply(a, b, e, e, e, e, e, e, e, None)
        torch.autograd.grad(o, (a, b), (ga, ga))
    micro("autograd.Function apply(10 args) + autograd.grad (no work)", af, 5000)
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(2000):
            small.step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
