"""Random ragged batches through toad_mil_multi_step_f32 against the sum of the oracle's per-slide fp64 gradients (the third check of
__graft_entry__.smoke() over many batch compositions). Not collected by pytest: `python tests/fuzz_multi.py [cases] [seed]` on a GPU box.
Tolerance 1e-3 of each gradient's scale (a ReLU-boundary flip may move a trunk gradient by one patch's contribution) for the trunk, 2e-5 + noise
for the mask-free head / attention-c gradients; per-slide losses and logits to 1e-4."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat, ops
from oracle import toad_oracle as orc           # checker only
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda:0")
c = 18
params = orc.xavier_params(c, seed=1)
gen = torch.Generator().manual_seed(11 + seed)
for k, v in params.items():
    if v.dim() == 1:
        v.normal_(0, 0.05, generator=gen)
model = TOAD_fc_mtl_concat(n_classes=c); model.load_state_dict(params); model.relocate()
w = {k: v.detach() for k, v in model._weights().items()}
p64 = {k: v.double() for k, v in params.items()}
KEYS = (("wcls", "classifier.weight", 2e-5), ("wsite", "site_classifier.weight", 2e-5), ("wc", "attention_net.4.attention_c.weight", 1e-4),
        ("w2", "attention_net.2.weight", 1e-3), ("w1", "attention_net.0.weight", 1e-3), ("wab", None, 1e-3))
nfail = 0
for i in range(cases):
    B = rng.randint(1, 12)
    lens = [rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 300, 777, rng.randint(1, 2500)]) for _ in range(B)]
    bags = [torch.randn(m, 1024, generator=gen) for m in lens]
    labels = torch.tensor([rng.randrange(c) for _ in range(B)]); sites = torch.tensor([rng.randint(0, 1) for _ in range(B)])
    sexes = torch.tensor([float(rng.randint(0, 1)) for _ in range(B)])
    g = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    out = ops.mil_multi_step(w, g, 0.0, [b.to(dev) for b in bags], sexes.to(dev), labels.to(dev), sites.to(dev), 0.75 / B, 0.25 / B)
    tot = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in params.items()}
    ok, msgs = True, []
    for j, (b, lb, st, sx) in enumerate(zip(bags, labels, sites, sexes)):
        o_out, o_loss, gd = orc.fwd_bwd(p64, b.double(), sx.reshape(1).double(), lb.reshape(1), st.reshape(1))
        for k in tot:
            tot[k] += gd[k] / B
    for slot, key, tol in KEYS:
        if key is None:
            continue
        sc = tot[key].abs().max().item()
        err = (g[slot].cpu().double() - tot[key]).abs().max().item()
        if err > tol * sc + 1e-12:
            # a trunk gradient may differ by a few LEGITIMATE ReLU-boundary flips: rank-one terms of one patch's size (tests/helpers.py)
            try:
                if tol < 1e-3:
                    raise AssertionError("mask-free gradient")
                nflip = helpers.assert_grad_close_or_few_flips(g[slot], tot[key], 1e-4, sc, what=slot, max_flips=8, flip_size=3e-2)
                msgs.append(f"{slot}: {nflip} flip(s)")
            except AssertionError as ex:
                ok = False; msgs.append(f"{slot} err {err:.2e} scale {sc:.2e} ({str(ex)[:60]})")
    nfail += 0 if ok else 1
    print(f"case {i}: B={B} lens={lens}: " + ("ok " + " ".join(msgs) if ok else "  ".join(msgs) + "   <<<<<< FAIL"), flush=True)
print(f"{nfail} failures over {cases} cases")
sys.exit(1 if nfail else 0)
