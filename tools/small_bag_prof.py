import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from toad_amd import TOAD_fc_mtl_concat
from toad_amd.dp import SlideShardedDP
n = int(sys.argv[1]); steps = int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
dp = SlideShardedDP(model, {"lr": 1e-4, "weight_decay": 1e-5})
bag = torch.randn(n, 1024, device=dev)
sl = [(bag, torch.tensor([1.0], device=dev), torch.tensor([3], device=dev), torch.tensor([1], device=dev))]
for _ in range(5): dp.step(sl, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): dp.step(sl, 1)
torch.cuda.synchronize()
print(f"N={n}: {(time.perf_counter()-t0)/steps*1e3:.3f} ms/step wall")
