"""Random bag sizes through the drop-in module (forward, weighted CE, backward) on fp32 and prepared bags against the CPU oracle - the check of
__graft_entry__.smoke() over many N. Not collected by pytest (no test_ prefix): `python tests/fuzz_mil.py [cases] [seed]` on a GPU box.
Outputs to 1e-4; gradients to 2e-5 of each gradient's own scale + 10x the fp32 round-off of the same computation (fp64 oracle backward on the
device's saved activations, so the ReLU masks are identical); prepared bag == fp32 bag bitwise in the forward."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat, functional as F_, ops
from oracle import toad_oracle as orc           # checker only
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda:0")
c = 18
params = orc.xavier_params(c, seed=1)
gen = torch.Generator().manual_seed(7 + seed)
for k, v in params.items():
    if v.dim() == 1:
        v.normal_(0, 0.05, generator=gen)
model = TOAD_fc_mtl_concat(n_classes=c); model.load_state_dict(params); model.relocate()
ce = torch.nn.CrossEntropyLoss()
nfail = 0
for i in range(cases):
    n = rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1025, 2047, 2049, 4095, 4097, 8191, 8193,
                    rng.randint(1, 300), rng.randint(300, 5000), rng.randint(5000, 40000)])
    x = torch.randn(n, 1024, generator=gen)
    sex = torch.tensor([float(rng.randint(0, 1))]); label = torch.tensor([rng.randrange(c)]); site = torch.tensor([rng.randint(0, 1)])
    msgs, ok = [], True
    for mode in ("fp32", "prepared"):
        if mode == "prepared" and n < 64:
            continue
        model.zero_grad(set_to_none=True)
        bag = x.to(dev) if mode == "fp32" else ops.prepare_bag(x.to(dev))
        res = model(bag, sex.to(dev))
        loss = ce(res["logits"], label.to(dev)) * 0.75 + ce(res["site_logits"], site.to(dev)) * 0.25
        loss.backward()
        if mode == "fp32":
            o_out, o_loss, _ = orc.fwd_bwd(params, x, sex, label, site)
            for k in ("logits", "Y_prob", "site_logits", "site_prob", "A"):
                err = (res[k].detach().cpu() - o_out[k]).abs().max().item()
                ok = ok and err <= 1e-4
            ok = ok and abs(loss.item() - o_loss.item()) <= 1e-4
            keep = {k: res[k].detach().clone() for k in ("logits", "A")}
            w = {k: v.detach() for k, v in model._weights().items()}
            outs, sv = F_.mil_forward(w, x.to(dev), sex.to(dev))
            dl, ds = orc.loss_grad(outs["logits"].cpu(), label, outs["site_logits"].cpu(), site)
            sv_cpu = orc.Saved(x=x, h1=sv.h1.cpu(), h=sv.h.cpu(), p=sv.p.cpu(), a_raw=sv.a_raw.cpu(), m=sv.m.cpu(), mcat=sv.mcat.cpu(), sex=sex)
            o32 = orc.backward(params, sv_cpu, dl, ds)
            sv64 = orc.Saved(**{k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in sv_cpu.__dict__.items()})
            og = orc.backward({k: v.double() for k, v in params.items()}, sv64, dl.double(), ds.double())
            ref_noise = {k: (o32[k].double() - og[k]).abs().max().item() for k in og}
        else:
            # prepared vs raw fp32 bag: a few ulp (the raw bag is scaled from each tile's first columns inside the GEMM, the prepared one with each
            # tile's maximum: tests/test_gpu_pt.py ROUTE_TOL)
            for kk in ("logits", "A"):
                ok = ok and (res[kk].detach() - keep[kk]).abs().max().item() <= 5e-6 * max(keep[kk].abs().max().item(), 1e-30)
            # The oracle backward runs on the activations THIS route saved (round 6): since the half-height tiles, a raw bag of 2.6k ... 16k patches sums its
            # first Linear whole-K while the prepared bag's 256-row tiles are K-split - H1 / H of the two routes differ by fp32 round-off, and a pre-activation
            # within round-off of zero then legitimately flips its ReLU mask (one rank-one term per flip in the trunk gradients: 2 of 40 cases of the first
            # round-6 sweep, both N = 4095, profiles/r07k_fuzz_sweeps.txt). Masks identical -> everything to 2e-5, as for the raw bag.
            ar = ops.mil_fwd(w, bag, sex.to(dev))
            L_, D2_ = w["w2"].shape[0], 2 * w["wa"].shape[0]
            sv_cpu = orc.Saved(x=x, h1=ar.view("h1", (n, w["w1"].shape[0])).cpu(), h=ar.view("h", (n, L_)).cpu(), p=ar.view("p", (n, D2_)).cpu(),
                               a_raw=ar.view("a_raw", (n, 2)).cpu(), m=ar.view("m", (2, L_)).cpu(), mcat=ar.view("mcat", (2, L_ + 1)).cpu(), sex=sex)
            dl, ds = orc.loss_grad(ar.view("logits", (1, c)).cpu(), label, ar.view("site_logits", (1, 2)).cpu(), site)
            o32 = orc.backward(params, sv_cpu, dl, ds)
            sv64 = orc.Saved(**{k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in sv_cpu.__dict__.items()})
            og = orc.backward({k: v.double() for k, v in params.items()}, sv64, dl.double(), ds.double())
            ref_noise = {k: (o32[k].double() - og[k]).abs().max().item() for k in og}
        worst = 0.0
        for k, p in model.named_parameters():
            scale = og[k].abs().max().item()
            if k.endswith("attention_c.bias"):
                scale = max(scale, og["attention_net.4.attention_c.weight"].abs().max().item())
            if scale == 0.0:                      # one-patch bags: the attention branch has no gradient
                scale = max(v.abs().max().item() for v in og.values())
            err = (p.grad.cpu().double() - og[k]).abs().max().item()
            worst = max(worst, err / scale)
            if err > 2e-5 * scale + 32.0 * ref_noise[k]:
                ok = False; msgs.append(f"{mode}:{k} err {err:.2e} scale {scale:.2e} noise {ref_noise[k]:.2e}")
        msgs.append(f"{mode} worst grad {worst:.1e}")
    nfail += 0 if ok else 1
    print(f"case {i}: N={n}: " + "  ".join(msgs) + ("" if ok else "   <<<<<< FAIL"), flush=True)
print(f"{nfail} failures over {cases} cases")
sys.exit(1 if nfail else 0)
