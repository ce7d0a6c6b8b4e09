import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import toad_oracle as orc
from toad_amd import functional as F_, ops
from tests.helpers import case_inputs, SLOT2KEY
name = sys.argv[1] if len(sys.argv) > 1 else "n777"
g = np.load("tests/golden/toad_golden.npz")
ci = case_inputs(g, name)
dev = torch.device("cuda:0")
w = {s: ci["params"][k].to(dev) for s, k in SLOT2KEY.items()}
x = ci["x"].to(dev); sex = ci["sex"].to(dev)
outs, s = F_.mil_forward(w, x, sex)
o_out, o_s = orc.forward(ci["params"], ci["x"], ci["sex"])
def st(nm, a, b):
    a = a.detach().cpu().double(); b = b.detach().double()
    e = (a - b).abs()
    print(f"{nm:10s} max|ref| {b.abs().max():.3e}  max err {e.max():.3e}  mean err {e.mean():.3e}  #>1e-5: {(e > 1e-5).sum().item()}")
st("h1", s.h1, o_s.h1); st("h", s.h, o_s.h); st("p", s.p, o_s.p); st("a_raw", s.a_raw, o_s.a_raw); st("m", s.m, o_s.m)
print("relu mask flips h1:", ((s.h1.cpu() > 0) != (o_s.h1 > 0)).sum().item(), " h:", ((s.h.cpu() > 0) != (o_s.h > 0)).sum().item())
print("tiny h1 (0<h1<1e-5):", ((o_s.h1 > 0) & (o_s.h1 < 1e-5)).sum().item(), "of", o_s.h1.numel())
dl, ds = orc.loss_grad(o_out["logits"], ci["label"], o_out["site_logits"], ci["site"])
gg, _ = F_.mil_backward(w, s, dl.to(dev), ds.to(dev))
og = orc.backward(ci["params"], o_s, dl, ds)
for sl, k in SLOT2KEY.items():
    st(sl, gg[sl], og[k])
# stage by stage with ORACLE inputs
d = 384
dm = orc.heads_bwd(o_s.mcat, dl, ds, ci["params"]["classifier.weight"], ci["params"]["site_classifier.weight"])[4]
rpa, rpb, rdh, rdwc, rdbc = orc.gated_pool_bwd(o_s.p[:, :d], o_s.p[:, d:], o_s.h, ci["params"][SLOT2KEY["wc"]], o_s.a_raw, o_s.m, dm)
dp, dh, dwc, dbc = ops.gated_pool_bwd(s.p, d, s.h, w["wc"], s.a_raw, s.stats, s.m, dm.to(dev))
st("dp", dp, torch.cat([rpa, rpb], 1)); st("dh_pool", dh, rdh)
wab = torch.cat([ci["params"][SLOT2KEY["wa"]], ci["params"][SLOT2KEY["wb"]]], 0)
rdp = torch.cat([rpa, rpb], 1)
rdz2 = (rdp @ wab + rdh) * (o_s.h > 0).float()
dz2 = ops.linear_dgrad(rdp.to(dev), ops.transpose(wab.to(dev)), rdh.to(dev), o_s.h.to(dev))
st("dz2", dz2, rdz2)
rdz1 = (rdz2 @ ci["params"][SLOT2KEY["w2"]]) * (o_s.h1 > 0).float()
dz1 = ops.linear_dgrad(rdz2.to(dev), ops.transpose(w["w2"]), None, o_s.h1.to(dev))
st("dz1", dz1, rdz1)
dw1, db1 = ops.linear_wgrad(rdz1.to(dev), x)
st("dw1", dw1, rdz1.t() @ ci["x"]); st("db1", db1, rdz1.sum(0))
dw1d = (rdz1.double().t() @ ci["x"].double())
st("dw1 vs f64", dw1, dw1d); st("cpu32 vs f64", (rdz1.t() @ ci["x"]), dw1d)
# ---- fp64 yardstick: how far is the fp32 CPU path itself from exact arithmetic?
p64 = {k: v.double() for k, v in ci["params"].items()}
o64, l64, g64 = orc.fwd_bwd(p64, ci["x"].double(), ci["sex"].double(), ci["label"], ci["site"])
print("---- vs fp64 oracle:   |cpu32-f64|   |gpu-f64|   |gpu-cpu32|   (max abs; rel-L2 gpu-f64)")
for sl, k in SLOT2KEY.items():
    a = (og[k].double() - g64[k]).abs().max().item()
    b = (gg[sl].cpu().double() - g64[k]).abs().max().item()
    c = (gg[sl].cpu().double() - og[k].double()).abs().max().item()
    r = (gg[sl].cpu().double() - g64[k]).norm().item() / max(g64[k].norm().item(), 1e-30)
    r32 = (og[k].double() - g64[k]).norm().item() / max(g64[k].norm().item(), 1e-30)
    print(f"{sl:6s} absmax {g64[k].abs().max():.2e}  {a:.2e}  {b:.2e}  {c:.2e}   relL2 gpu {r:.2e} cpu32 {r32:.2e}")
for k in ("logits", "site_logits", "A"):
    print(k, "cpu32-f64 %.2e gpu-f64 %.2e" % ((o_out[k].double() - o64[k]).abs().max().item(),
          (outs["A_nt"].t().cpu().double() - o64[k]).abs().max().item() if k == "A" else (outs[k].cpu().double() - o64[k]).abs().max().item()))
