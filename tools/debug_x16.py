import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from toad_amd import TOAD_fc_mtl_concat
torch.manual_seed(5)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
for n in (256, 777):
    g = torch.Generator().manual_seed(40 + n)
    x16 = (torch.randn(n, 1024, generator=g) * 0.7).half().cuda(); x32 = x16.float()
    sex = torch.tensor([1.0]).cuda(); label = torch.tensor([3]).cuda(); site = torch.tensor([1]).cuda()
    lf = torch.nn.CrossEntropyLoss()
    def run(x):
        model.zero_grad(set_to_none=True)
        r = model(x, sex)
        (lf(r["logits"], label) * 0.75 + lf(r["site_logits"], site) * 0.25).backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    a, b = run(x32), run(x16)
    for k in a:
        d = (a[k] - b[k]).abs()
        print(n, k, tuple(a[k].shape), "max|ref| %.3e maxdiff %.3e" % (a[k].abs().max().item(), d.max().item()))
    d = (a["attention_net.0.weight"] - b["attention_net.0.weight"]).abs()
    print("rows with error:", (d.max(1).values > 1e-6).nonzero().flatten()[:20].tolist(), "count", int((d.max(1).values > 1e-6).sum()))
    cols = (d.max(0).values > 1e-6).nonzero().flatten()
    print("cols with error count", len(cols), cols[:40].tolist())
    r = (b["attention_net.0.weight"] / a["attention_net.0.weight"])
    print("ratio sample", r[0, :8].tolist())
