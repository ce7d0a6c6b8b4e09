import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import toad_oracle as orc
from toad_amd import TOAD_fc_mtl_concat
dev = torch.device("cuda:0")
params = orc.xavier_params(18, seed=2)
m = TOAD_fc_mtl_concat(n_classes=18); m.load_state_dict(params); m.relocate(); m.train()
g = torch.Generator().manual_seed(3)
small = torch.randn(1100, 1024, generator=g)
r = 1000
big = small.to(dev).repeat(r, 1)            # 1.1M rows, 4.5 GB: beyond the 32-bit-offset fast path (M*K*4 >= 2^32)
sex = torch.ones(1, device=dev)
t0 = time.time()
out_b = m(big, sex); 
ce = torch.nn.CrossEntropyLoss()
loss = ce(out_b["logits"], torch.tensor([3], device=dev)) * 0.75 + ce(out_b["site_logits"], torch.tensor([1], device=dev)) * 0.25
loss.backward(); torch.cuda.synchronize()
gb = {k: p.grad.clone() for k, p in m.named_parameters()}
print("big fwd+bwd", time.time() - t0, "s")
m.zero_grad()
out_s = m(small.to(dev), sex)
loss_s = ce(out_s["logits"], torch.tensor([3], device=dev)) * 0.75 + ce(out_s["site_logits"], torch.tensor([1], device=dev)) * 0.25
loss_s.backward()
print("logits diff", (out_b["logits"] - out_s["logits"]).abs().max().item(), "A diff", (out_b["A"][:, :1100] - out_s["A"]).abs().max().item())
for k, p in m.named_parameters():
    e = (gb[k] - p.grad).abs().max().item(); s = p.grad.abs().max().item()
    print(f"{k:45s} grad diff {e:.2e} scale {s:.2e}")
# ---- warm timing of the fused step on the 1.1 M-row bag (round 5: the NT GEMMs run the fp16 two-piece kernels over row chunks, csrc/step.hip nt_rows)
from toad_amd.dp import SlideShardedDP
dp = SlideShardedDP(m, {"lr": 1e-4, "weight_decay": 1e-5})
slide = (big, sex, torch.tensor([3], device=dev), torch.tensor([1], device=dev))
for _ in range(2):
    dp.step([slide], 1)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5):
    dp.step([slide], 1)
torch.cuda.synchronize()
ms = (time.time() - t0) / 5 * 1e3
print(f"fused step on {big.shape[0]} patches: {ms:.2f} ms = {ms / big.shape[0] * 1e5:.3f} ms per 100k patches ({6029312.0 * big.shape[0] / ms / 1e9:.0f} TF-eq whole step)")
