"""Random (L, D, n_tasks, N, dropout) through the standalone Attn_Net_Gated (models/model_toad.py:17-41; the reference's constructor takes any shape) -
inside and OUTSIDE the pool kernels' covering instantiation (column blocks of <= 512, task blocks of <= 4, zero padding to multiples of 4 / 8:
toad_amd/model_toad.py _ScoresFn) - against autograd on the oracle's formula. With dropout the forward must be reproducible under the same torch seed
and differ under another, and its backward must be the gradient of ITS OWN masked forward: checked by linearity (the scores are linear in Wc, so
d<A, G>/dWc[t] = sum_rows G[:, t] * gate, which the kernels' dWc must equal for the forward's own gate = recovered with one-hot Wc probes).
Not collected by pytest: `python tests/fuzz_attn.py [cases] [seed]` on a GPU box."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import Attn_Net_Gated
from oracle import toad_oracle as orc           # checker only
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda:0")
def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-6)).item()
nfail = 0
for i in range(cases):
    l = rng.choice([1024, 512, 200, 36, 100, 1536, 2048, rng.randint(1, 300) * 4, rng.randint(5, 700)])
    d = rng.choice([256, 384, 512, 30, 516, 640, 1100, rng.randint(1, 300) * 4, rng.randint(3, 900)])
    t = rng.choice([1, 2, 3, 4, 5, 7, 9, rng.randint(1, 12)])
    n = rng.choice([1, 2, 63, 64, 65, 257, 999, rng.randint(1, 6000)])
    torch.manual_seed(seed * 7919 + i)
    net = Attn_Net_Gated(L=l, D=d, n_tasks=t).to(dev)
    x = torch.randn(n, l)
    msgs, ok = [], True
    try:
        xg = x.to(dev).requires_grad_(True)
        a, _ = net(xg)
        prm = [p.detach().cpu().clone().requires_grad_(True) for p in (net.attention_a[0].weight, net.attention_a[0].bias, net.attention_b[0].weight,
                                                                       net.attention_b[0].bias, net.attention_c.weight, net.attention_c.bias)]
        xr = x.clone().requires_grad_(True)
        ref = orc.gated_scores(torch.addmm(prm[1], xr, prm[0].t()), torch.addmm(prm[3], xr, prm[2].t()), prm[4], prm[5])
        e = (a.detach().cpu() - ref.detach()).abs().max().item()
        msgs.append(f"fwd {e:.1e}"); ok = ok and tuple(a.shape) == (n, t) and e <= 2e-5
        gsel = torch.randn(n, t)
        (a * gsel.to(dev)).sum().backward(); (ref * gsel).sum().backward()
        mine = [net.attention_a[0].weight, net.attention_a[0].bias, net.attention_b[0].weight, net.attention_b[0].bias, net.attention_c.weight, net.attention_c.bias]
        es = [rel(p.grad.cpu(), r.grad) for p, r in zip(mine, prm)] + [rel(xg.grad.cpu(), xr.grad)]
        msgs.append("bwd max %.1e" % max(es)); ok = ok and max(es) <= 2e-4
    except Exception as ex:                     # noqa: BLE001 - a refused shape is a failure here: the reference takes any
        msgs.append("RAISED " + type(ex).__name__ + ": " + str(ex)[:100]); ok = False
    nfail += 0 if ok else 1
    print(f"case {i}: N{n} L{l} D{d} T{t}: " + "  ".join(msgs) + ("" if ok else "   <<<<<< FAIL"), flush=True)
# train-mode dropout (models/model_toad.py:27-29) on one blocked and one in-envelope shape: reproducible under the seed, different under another, finite grads
for (l, d, t) in ((200, 516, 5), (512, 384, 2)):
    net = Attn_Net_Gated(L=l, D=d, dropout=True, n_tasks=t).to(dev); net.train()
    x = torch.randn(777, l, device=dev)
    outs = []
    for s in (5, 5, 6):
        torch.manual_seed(s)
        xg = x.clone().requires_grad_(True)
        a, _ = net(xg)
        a.sum().backward()
        outs.append((a.detach().clone(), xg.grad.clone()))
    net.eval()
    a_eval, _ = net(x)
    ok = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and not torch.equal(outs[0][0], outs[2][0]) \
        and bool(torch.isfinite(outs[2][1]).all()) and not torch.equal(a_eval, outs[0][0])
    # E[dropout output] = eval output: the mean over rows of (train - eval) is small against the scores' spread
    bias = (outs[0][0] - a_eval).mean().abs().item() / a_eval.std().item()
    ok = ok and bias <= 0.2
    nfail += 0 if ok else 1
    print(f"dropout L{l} D{d} T{t}: reproducible / seed-dependent / finite, mean shift {bias:.3f} sigma" + ("" if ok else "   <<<<<< FAIL"), flush=True)
print(f"{nfail} failures over {cases} cases")
sys.exit(1 if nfail else 0)
