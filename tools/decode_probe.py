"""Times the pieces of the fused neural-Gaussian decode (diagnostic)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decode_oracle as DO
from gscream_amd import neural_gaussians as NG

dev = "cuda"
N, K = 200_000, 10
model = DO.Model(N, K, seed=11, dtype=torch.float32, spread=1.5).to(dev)
cam = torch.tensor([0.0, 0.0, -6.0], device=dev)
feat = model._anchor_feat.detach().clone().requires_grad_(True)
anchor = model._anchor.detach().clone().requires_grad_(True)
off = model._offset.detach().clone().requires_grad_(True)
gs = torch.exp(model._scaling.detach()).requires_grad_(True)
mlps = (model.mlp_opacity, model.mlp_uncertainty, model.mlp_color, model.mlp_cov)

def timed(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def fwd():
    return NG.decode(feat, anchor, off, gs, cam, *mlps)
def fwdbwd():
    out = fwd()
    params = [feat, anchor, off, gs] + [p for m in mlps for p in m.parameters()]
    torch.autograd.grad(sum(o.sum() for o in out[:6]), params)
print("decode fwd ms", timed(fwd), " fwd+bwd ms", timed(fwdbwd))
