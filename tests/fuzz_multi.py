"""Random ragged batches through toad_mil_multi_step_f32 against the sum of the oracle's per-slide fp64 gradients (the third check of
__graft_entry__.smoke() over many batch compositions). Not collected by pytest: `python tests/fuzz_multi.py [cases] [seed] [big]` on a GPU box.
Checked per case: all 14 parameter gradients (the stacked attention_a / attention_b weights and biases through the `wab` / `bab` slots, every
bias slot) - 1e-3 of each gradient's scale for the trunk / attention weights (a ReLU-boundary flip may move them by one patch's contribution;
then the rank-one test decides), 2e-5 .. 1e-4 for the mask-free head / attention-c gradients - and the per-slide losses and logits the call
returns, to 1e-4 against the oracle's fp64 values."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat, ops
from oracle import toad_oracle as orc           # checker only
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"      # up to 40 slides of up to 30,000 patches: many slide boundaries inside 256-row tiles, >= 512-tile launches
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
A, B_ = "attention_net.4.attention_a.0.", "attention_net.4.attention_b.0."
KEYS = (("wcls", "classifier.weight", 2e-5), ("bcls", "classifier.bias", 2e-5), ("wsite", "site_classifier.weight", 2e-5),
        ("bsite", "site_classifier.bias", 2e-5), ("wc", "attention_net.4.attention_c.weight", 1e-4), ("bc", "attention_net.4.attention_c.bias", 1e-4),
        ("w2", "attention_net.2.weight", 1e-3), ("b2", "attention_net.2.bias", 1e-3), ("w1", "attention_net.0.weight", 1e-3),
        ("b1", "attention_net.0.bias", 1e-3), ("wab", (A + "weight", B_ + "weight"), 1e-3), ("bab", (A + "bias", B_ + "bias"), 1e-3))
nfail = 0
for i in range(cases):
    B = rng.randint(1, 40 if big else 12)
    lens = [rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 300, 777, rng.randint(1, 2500)] + ([511, 513, 1023, 4097, rng.randint(1, 30000), rng.randint(1, 30000)] if big else []))
            for _ in range(B)]
    bags = [torch.randn(m, 1024, generator=gen) for m in lens]
    labels = torch.tensor([rng.randrange(c) for _ in range(B)]); sites = torch.tensor([rng.randint(0, 1) for _ in range(B)])
    sexes = torch.tensor([float(rng.randint(0, 1)) for _ in range(B)])
    g = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    out = ops.mil_multi_step(w, g, 0.0, [b.to(dev) for b in bags], sexes.to(dev), labels.to(dev), sites.to(dev), 0.75 / B, 0.25 / B, want_logits=True)
    tot = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in params.items()}
    ok, msgs = True, []
    for j, (b, lb, st, sx) in enumerate(zip(bags, labels, sites, sexes)):
        o_out, o_loss, gd = orc.fwd_bwd(p64, b.double(), sx.reshape(1).double(), lb.reshape(1), st.reshape(1))
        for k in tot:
            tot[k] += gd[k] / B
        # per-slide results of the call: loss [B, 3] = (weighted loss / B, class CE, site CE), logits [B, C], site logits [B, 2]
        cls_ce = torch.nn.functional.cross_entropy(o_out["logits"], lb.reshape(1)).item()
        site_ce = torch.nn.functional.cross_entropy(o_out["site_logits"], st.reshape(1)).item()
        got = out[0][j].cpu().double()
        e_loss = max(abs(got[0].item() - float(o_loss) / B), abs(got[1].item() - cls_ce), abs(got[2].item() - site_ce))
        e_log = max((out[1][j].cpu().double() - o_out["logits"][0]).abs().max().item(), (out[2][j].cpu().double() - o_out["site_logits"][0]).abs().max().item())
        if e_loss > 1e-4 or e_log > 1e-4:
            ok = False; msgs.append(f"slide {j}: loss err {e_loss:.2e} logits err {e_log:.2e}")
    flips = {}
    gmax = max(v.abs().max().item() for v in tot.values())
    for slot, key, tol in sorted(KEYS, key=lambda t: t[0].startswith("b")):          # weights first: their flip counts license the bias slots
        ref = torch.cat([tot[key[0]], tot[key[1]]]) if isinstance(key, tuple) else tot[key]
        sc = ref.abs().max().item()
        if slot == "bc":                 # exactly zero by the softmax's shift invariance: what is returned is round-off of terms of dWc's size
            sc = max(sc, tot["attention_net.4.attention_c.weight"].abs().max().item())
        err = (g[slot].cpu().double() - ref).abs().max().item()
        # head biases are SUMS over the batch of (softmax - onehot) * w / B: with many slides they cancel to far below the size of their terms, and what
        # any fp32 summation returns is round-off of those terms (|term| <= w / B), not of the result: floor of 2e-6 of one term's bound per sqrt(B)
        floor = {"bcls": 2e-6 * 0.75, "bsite": 2e-6 * 0.25}.get(slot, 0.0) / max(B, 1) ** 0.5
        # gradients that are identically zero (a batch of one-patch bags: the softmax over one patch has no gradient, so the attention and trunk slots
        # hold round-off of the other slots' size): 1e-6 of the largest gradient of the step, as tests/helpers.py does for the all-equal bag
        floor += 1e-6 * gmax if sc < 1e-9 * gmax else 0.0
        if err > tol * sc + floor + 1e-12:
            # a trunk gradient may differ by a few LEGITIMATE ReLU-boundary flips: rank-one terms of one patch's size (tests/helpers.py);
            # the bias of a layer then moves by that patch's dZ element, allowed only when the layer's weight gradient showed the flip
            try:
                if tol < 1e-3:
                    raise AssertionError("mask-free gradient")
                if ref.dim() < 2:
                    nf = flips.get("w" + slot[1:], 0)
                    if nf == 0 or err > 3e-2 * sc * nf:
                        raise AssertionError(f"bias gradient off with {nf} flip(s) in its layer")
                    msgs.append(f"{slot}: moved by its layer's flip(s)")
                else:
                    nflip = helpers.assert_grad_close_or_few_flips(g[slot], ref, 1e-4, sc, what=slot, max_flips=8, flip_size=3e-2)
                    flips[slot] = nflip
                    msgs.append(f"{slot}: {nflip} flip(s)")
            except AssertionError as ex:
                ok = False; msgs.append(f"{slot} err {err:.2e} scale {sc:.2e} ({str(ex)[:60]})")
    nfail += 0 if ok else 1
    print(f"case {i}: B={B} lens={lens}: " + ("ok " + " ".join(msgs) if ok else "  ".join(msgs) + "   <<<<<< FAIL"), flush=True)
print(f"{nfail} failures over {cases} cases")
sys.exit(1 if nfail else 0)
