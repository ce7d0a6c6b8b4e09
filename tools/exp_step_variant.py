"""Experiment helper: the per-slide training step on a tagged library variant (TOAD_HIP_LIB=...), for rocprofv3 --kernel-trace --stats.
python tools/exp_step_variant.py N steps"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.ab.select_lib as sl
import torch
from toad_amd import TOAD_fc_mtl_concat
from toad_amd.dp import SlideShardedDP
n = int(sys.argv[1]); steps = int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
dp = SlideShardedDP(model, {"lr": 1e-4, "weight_decay": 1e-5})
bags = [torch.randn(n, 1024, device=dev) for _ in range(2)]
sl_ = [[(b, torch.tensor([1.0], device=dev), torch.tensor([3], device=dev), torch.tensor([1], device=dev))] for b in bags]
for i in range(10): dp.step(sl_[i % 2], 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): dp.step(sl_[i % 2], 1)
torch.cuda.synchronize()
print(f"{sl.TAG} N={n}: {(time.perf_counter()-t0)/steps*1e3:.4f} ms/step wall", flush=True)
