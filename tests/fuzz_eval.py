"""Random ragged batches through the forward-only multi-slide path (model.forward_many) against one model(data, sex) call per slide.
Not collected by pytest: `python tests/fuzz_eval.py [cases] [seed]` on a GPU box. The two paths scale their GEMM operands per 256-row block of
different row sets, so they agree to fp32 round-off (2e-5 here), not bitwise; predictions must be identical unless the top-2 gap is at round-off."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda:0")
torch.manual_seed(seed)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.eval()
gen = torch.Generator().manual_seed(5 + seed)
nfail = 0
for i in range(cases):
    B = rng.randint(1, 16)
    lens = [rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 300, 777, 2049, rng.randint(1, 6000)]) for _ in range(B)]
    bags = [torch.randn(m, 1024, generator=gen).to(dev) for m in lens]
    sexes = [torch.tensor([float(rng.randint(0, 1))], device=dev) for _ in range(B)]
    with torch.no_grad():
        many = model.forward_many(bags, sexes, return_features=True)
        ok, worst = True, 0.0
        for b, sx, r in zip(bags, sexes, many):
            one = model(b, sx, return_features=True)
            for k in ("logits", "Y_prob", "site_logits", "site_prob", "A", "features"):
                e = (r[k] - one[k]).abs().max().item()
                worst = max(worst, e)
                ok = ok and e <= 2e-5
            if int(r["Y_hat"]) != int(one["Y_hat"]):
                top2 = one["logits"].flatten().topk(2).values
                ok = ok and float(top2[0] - top2[1]) <= 1e-5
    nfail += 0 if ok else 1
    print(f"case {i}: B={B} lens={lens}: worst |diff| {worst:.1e}" + ("" if ok else "   <<<<<< FAIL"), flush=True)
print(f"{nfail} failures over {cases} cases")
sys.exit(1 if nfail else 0)
