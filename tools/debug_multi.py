"""debug: multi-slide step vs per-slide steps vs fp64 oracle, per gradient slot"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat, ops
from oracle import toad_oracle as orc
from tests.helpers import SLOT2KEY
dev = torch.device("cuda:0")
def run(lens):
    torch.manual_seed(0)
    m = TOAD_fc_mtl_concat(n_classes=18)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.relocate()
    w = {k: v.detach() for k, v in m._weights().items()}
    B = len(lens)
    slides = []
    for i, n in enumerate(lens):
        g = torch.Generator().manual_seed(i)
        slides.append((torch.randn(n, 1024, generator=g), torch.tensor([float(i % 2)]), torch.tensor([(7 * i) % 18]), torch.tensor([(i // 2) % 2])))
    ds = [tuple(t.to(dev) for t in s) for s in slides]
    g1 = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    for i, s in enumerate(ds):
        ops.mil_step(w, g1, 0.0 if i == 0 else 1.0, s[0], s[1], s[2], s[3], 0.75 / B, 0.25 / B)
    g2 = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    sex = torch.cat([s[1] for s in ds]); label = torch.cat([s[2] for s in ds]); site = torch.cat([s[3] for s in ds])
    ops.mil_multi_step(w, g2, 0.0, [s[0] for s in ds], sex, label, site, 0.75 / B, 0.25 / B)
    p64 = {k: v.double() for k, v in params.items()}
    tot = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in params.items()}
    for (bag, sx, lb, st) in slides:
        _, _, gd = orc.fwd_bwd(p64, bag.double(), sx.double(), lb, st)
        for k in tot: tot[k] += gd[k] / B
    d = w["wc"].shape[1]
    def full(g):
        o = dict(g); o["wa"], o["wb"], o["ba"], o["bb"] = g["wab"][:d], g["wab"][d:], g["bab"][:d], g["bab"][d:]; return o
    f1, f2 = full(g1), full(g2)
    print("lens", lens)
    for slot, key in SLOT2KEY.items():
        r = tot[key]; sc = max(r.abs().max().item(), 1e-30)
        e1 = (f1[slot].cpu().double() - r).abs(); e2 = (f2[slot].cpu().double() - r).abs()
        print(f"  {slot:6s} scale {sc:9.3e} | per-slide max {e1.max().item()/sc:9.2e} frac>5e-5 {(e1 > 5e-5*sc).float().mean().item():6.3f} | multi max {e2.max().item()/sc:9.2e} frac {(e2 > 5e-5*sc).float().mean().item():6.3f}")
for lens in ([256] * 8, [2048], [256], [300, 500], [5000]):
    run(lens)
