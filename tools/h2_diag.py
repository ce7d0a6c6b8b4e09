"""Accuracy diagnostics of the GEMM arithmetic on the GPU (run with TOAD_GEMM_H2=1 and =0 to compare the fp16 two-piece path with
the bf16 three-piece path): (1) single products against fp64: max / rms error, error relative to sum|a.b|, mean SIGNED error
(a biased accumulation shows up there); (2) the n256 golden case: every intermediate of the backward chain against the fp64
oracle evaluated on the SAME saved activations (identical ReLU masks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from toad_amd import ops, functional as F_
from oracle import toad_oracle as orc
from tests.helpers import case_inputs, SLOT2KEY
dev = torch.device("cuda:0")
tag = os.environ.get("TOAD_GEMM_H2", "1")
g = torch.Generator().manual_seed(0)


def report(name, got, ref, sab=None):
    err = got.double() - ref
    s = f"[h2={tag}] {name:28s} max|err| {err.abs().max():.3e} rms {err.pow(2).mean().sqrt():.3e} rel-to-absmax {err.abs().max() / ref.abs().max():.3e} mean signed {err.mean():+.3e}"
    if sab is not None:
        s += f" max err/sum|ab| {(err.abs() / sab).max():.3e} mean signed/mean sum|ab| {err.mean() / sab.mean():+.3e}"
    print(s, flush=True)


for (M, K, N, sx) in ((3000, 1024, 512, 2.0), (3000, 512, 768, 1.0), (256, 512, 512, 1e-4)):
    x = torch.randn(M, K, generator=g) * sx; w = torch.randn(N, K, generator=g) * 0.04
    ref = x.double() @ w.double().t(); sab = x.double().abs() @ w.double().abs().t()
    report(f"fwd {M}x{K}x{N}", ops.linear_act_fwd(x.to(dev), w.to(dev), None, 0).cpu(), ref, sab)
    xr = x.relu()                                    # non-negative operand (post-ReLU activations): a bias cannot hide in sign symmetry
    report(f"fwd relu-x {M}x{K}x{N}", ops.linear_act_fwd(xr.to(dev), w.abs().to(dev), None, 0).cpu(), xr.double() @ w.abs().double().t(),
           xr.double() @ w.abs().double().t())
    dy = torch.randn(M, N, generator=g) * 1e-3
    report(f"wgrad {M}", ops.linear_wgrad(dy.to(dev), x.to(dev))[0].cpu(), dy.double().t() @ x.double(), dy.double().abs().t() @ x.double().abs())

golden = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "toad_golden.npz"))
for name in ("n256", "n777"):
    ci = case_inputs(golden, name)
    w = {s_: ci["params"][k].to(dev) for s_, k in SLOT2KEY.items()}
    outs, sv = F_.mil_forward(w, ci["x"].to(dev), ci["sex"].to(dev))
    dl, ds = orc.loss_grad(outs["logits"].cpu(), ci["label"], outs["site_logits"].cpu(), ci["site"])
    p64 = {k: v.double() for k, v in ci["params"].items()}
    d = 384
    # fp64 chain on the device's saved activations
    s64 = {k: getattr(sv, k).cpu().double() for k in ("h1", "h", "p", "a_raw", "m", "mcat")}
    x64 = ci["x"].double()
    _, _, _, _, dm64 = orc.heads_bwd(s64["mcat"], dl.double(), ds.double(), p64["classifier.weight"], p64["site_classifier.weight"])
    dpa, dpb, dhp, dwc, dbc = orc.gated_pool_bwd(s64["p"][:, :d], s64["p"][:, d:], s64["h"], p64["attention_net.4.attention_c.weight"], s64["a_raw"], s64["m"], dm64)
    dp64 = torch.cat([dpa, dpb], 1)
    wab64 = torch.cat([p64["attention_net.4.attention_a.0.weight"], p64["attention_net.4.attention_b.0.weight"]], 0)
    dz2_64 = (dp64 @ wab64 + dhp) * (s64["h"] > 0)
    dz1_64 = (dz2_64 @ p64["attention_net.2.weight"]) * (s64["h1"] > 0)
    # device chain, op by op
    hb = ops.heads_bwd(sv.mcat, dl.to(dev), ds.to(dev), w["wcls"], w["wsite"])
    dm = hb[4]
    dp, _, _, _, dp_amax = ops.gated_pool_bwd(sv.p, d, sv.h, w["wc"], sv.a_raw, sv.stats, sv.m, dm, want_dh=False, want_amax=True)
    wab = torch.cat([w["wa"], w["wb"]], 0)
    dz2, dz2_amax = ops.linear_dgrad(dp, ops.transpose(wab), relu_src=sv.h, pool=(sv.a_raw, sv.stats, dm), dy_amax=dp_amax, want_amax=True)
    dz1, dz1_amax = ops.linear_dgrad(dz2, ops.transpose(w["w2"]), relu_src=sv.h1, dy_amax=dz2_amax, want_amax=True)
    dw1, db1 = ops.linear_wgrad(dz1, sv.x, dy_amax=dz1_amax, x_amax=sv.x_amax)
    print(f"--- {name}: dp_amax bound / true max = {(dp_amax.cpu() / dp.abs().max().cpu()).tolist()[:2]}")
    report(name + " dM", dm.cpu(), dm64)
    report(name + " dP", dp.cpu(), dp64)
    report(name + " dZ2", dz2.cpu(), dz2_64)
    report(name + " dZ1", dz1.cpu(), dz1_64)
    # dZ1 from the EXACT dZ2 (isolates the second dgrad)
    dz1b = ops.linear_dgrad(dz2_64.float().to(dev), ops.transpose(w["w2"]), relu_src=sv.h1)
    report(name + " dZ1 | exact dZ2", dz1b.cpu(), dz1_64)
    report(name + " db1", db1.cpu(), dz1_64.sum(0))
    report(name + " db1 = colsum(dev dZ1)", dz1.cpu().double().sum(0).float(), dz1_64.sum(0))
    report(name + " dW1", dw1.cpu(), dz1_64.t() @ x64)
    cond = dz1_64.abs().sum(0) / dz1_64.sum(0).abs().clamp_min(1e-300)
    print(f"    conditioning of db1 columns (sum|terms| / |sum|): median {cond.median():.1f} max {cond.max():.1f}; row-scale spread of dZ1: max|row| min/max = {(dz1_64.abs().max(1).values.min() / dz1_64.abs().max()).item():.2e}")
