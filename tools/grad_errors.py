import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat, functional as F_
from oracle import toad_oracle as orc
from tests.helpers import SLOT2KEY
dev = torch.device("cuda:0")
n, c = int(sys.argv[1]) if len(sys.argv) > 1 else 1337, 18
params = orc.xavier_params(c, seed=1)
gen = torch.Generator().manual_seed(7)
for k, v in params.items():
    if v.dim() == 1: v.normal_(0, 0.05, generator=gen)
x = torch.randn(n, 1024, generator=gen); sex = torch.tensor([1.0]); label = torch.tensor([4]); site = torch.tensor([1])
w = {s: params[k].to(dev) for s, k in SLOT2KEY.items()}
outs, sv = F_.mil_forward(w, x.to(dev), sex.to(dev))
dl, ds = orc.loss_grad(outs["logits"].cpu(), label, outs["site_logits"].cpu(), site)
g, _ = F_.mil_backward(w, sv, dl.to(dev), ds.to(dev))
p64 = {k: v.double() for k, v in params.items()}
sv64 = orc.Saved(x=x.double(), h1=sv.h1.cpu().double(), h=sv.h.cpu().double(), p=sv.p.cpu().double(), a_raw=sv.a_raw.cpu().double(),
                 m=sv.m.cpu().double(), mcat=sv.mcat.cpu().double(), sex=sex.double())
og = orc.backward(p64, sv64, dl.double(), ds.double())
o64, s64 = orc.forward(p64, x.double(), sex.double())
print(os.environ.get("TAG"), "fwd: h1 %.2e h %.2e p %.2e logits %.2e" % tuple((a.cpu().double() - b).abs().max().item() for a, b in ((sv.h1, s64.h1), (sv.h, s64.h), (sv.p, s64.p), (outs["logits"], o64["logits"]))))
for sl, k in SLOT2KEY.items():
    ref = og[k]; e = (g[sl].cpu().double() - ref)
    print(f"   {sl:6s} scale {ref.abs().max():.2e}  max err {e.abs().max():.2e}  rel {e.abs().max()/max(ref.abs().max().item(),1e-3):.2e}  mean signed {e.mean():+.2e}")
