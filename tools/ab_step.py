"""Per-call timing of the fused training step (HIP events around the pool forward and each of the eight GEMM calls), for A/B runs
of library variants on ONE box: TOAD_HIP_LIB=<variant.so> python tools/ab_step.py [N] [steps]. Prints mean microseconds per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import tools.ab.select_lib as _sel      # TOAD_HIP_LIB=<variant.so> is honoured HERE, not by the product's loader
from toad_amd import TOAD_fc_mtl_concat, ops
from toad_amd.dp import SlideShardedDP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
dp = SlideShardedDP(model, {"lr": 1e-4, "weight_decay": 1e-5})
slides = []
for b in range(2):
    g = torch.Generator(device=dev).manual_seed(1000 + b)
    slides.append((torch.randn(n, 1024, device=dev, generator=g), torch.tensor([1.0], device=dev), torch.tensor([3], device=dev), torch.tensor([1], device=dev)))
if os.environ.get("TOAD_BAG", "fp32") == "prepared":         # the ingest format (ops.prepare_bag); TOAD_BAG=fp32 for the fp32 bag
    slides = [(ops.prepare_bag(s[0]),) + s[1:] for s in slides]
for i in range(5):
    dp.step([slides[i % 2]], 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    dp.step([slides[i % 2]], 1)
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / steps * 1e3
ops.enable_timing(True, level=2, prealloc=18 * steps)
for i in range(steps):
    dp.step([slides[i % 2]], 1)
torch.cuda.synchronize()
T = ops._TIMING
def mean(lst): return sum(a.elapsed_time(b) for a, b in lst) / len(lst) * 1e3
f, w, d = T["gemm_fwd"], T["gemm_wgrad"], T["gemm_dgrad"]
row = {"fwd1": mean(f[0::3]), "fwd2": mean(f[1::3]), "fwd_ab": mean(f[2::3]), "wgrad_ab": mean(w[0::3]), "dgrad_ab": mean(d[0::2]),
       "wgrad2": mean(w[1::3]), "dgrad2": mean(d[1::2]), "wgrad1": mean(w[2::3]), "pool_fwd": mean(T["pool_fwd"])}
tag = _sel.TAG + "/" + os.environ.get("TOAD_BAG", "fp32")
print(f"{tag:20s} step {plain:6.3f} ms | " + " ".join(f"{k} {v:6.1f}" for k, v in row.items()) + f" | gemm sum {sum(v for k, v in row.items() if k != 'pool_fwd'):7.1f}", flush=True)
