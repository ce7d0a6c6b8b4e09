import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, K, N = 3000, 1024, 512
x = torch.randn(M, K, generator=g) * 2.0; w = torch.randn(N, K, generator=g) * 0.04; b = torch.randn(N, generator=g)
ref = x.double() @ w.double().t() + b.double()
sab = (x.double().abs() @ w.double().abs().t())
y = ops.linear_act_fwd(x.to(dev), w.to(dev), b.to(dev), 0).cpu().double()
err = (y - ref).abs()
print(f"{os.environ.get('TAG','?'):10s} fwd max|err| {err.max():.3e}  rms {err.pow(2).mean().sqrt():.3e}  max err/sum|ab| {(err/sab).max():.3e}  mean signed {((y-ref)).mean():+.3e}")
dy = torch.randn(M, N, generator=g); wt = w.t().contiguous()   # dgrad: dX = dY W
refd = dy.double() @ w.double()
dx = ops.linear_dgrad(dy.to(dev), ops.transpose(w.to(dev))).cpu().double()
errd = (dx - refd).abs()
print(f"{'':10s} dgrad max|err| {errd.max():.3e}  rms {errd.pow(2).mean().sqrt():.3e}")
