import torch
N=200_000
d2=torch.randn(120,N,device="cuda"); H=torch.randn(128,N,device="cuda"); X=torch.randn(36,N,device="cuda"); d1=torch.randn(128,N,device="cuda")
def timed(fn,n=10):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
print("direct d2[70]@H[32].t", timed(lambda: d2[50:120] @ H[96:128].t()))
print("direct d1[32]@X.t", timed(lambda: d1[0:32] @ X.t()))
print("all-at-once d2 @ H.t (120x128)", timed(lambda: d2 @ H.t()))
print("all-at-once d1 @ X.t (128x36)", timed(lambda: d1 @ X.t()))
for S in (50, 200, 1000):
    def split():
        a=d2.view(120,S,N//S).permute(1,0,2); b=H.view(128,S,N//S).permute(1,2,0)
        return torch.bmm(a,b).sum(0)
    print("split-K bmm S=%d (120x128)"%S, timed(split))
print("sum rows", timed(lambda: d2.sum(1)))
ref=(d2.double()@H.double().t()); got=d2@H.t(); print("err", ((got.double()-ref).abs().max()/ref.abs().max()).item())
