"""Per-slide training step (toad_mil_step_f32 + flat Adam: one optimiser step per slide, the reference's semantics) over a sweep of bag sizes:
ms per step, slides/s, us per 1,000 patches. Shows where the tile plans switch (half-height NT tiles: 2.6k ... 16k patches at N = 512;
batched weight gradients: <= 262,144 rows).   python tools/size_sweep.py [sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat
from toad_amd.dp import SlideShardedDP
sizes = [int(a) for a in sys.argv[1:]] or [256, 1000, 2000, 2500, 2700, 4000, 6000, 8000, 10000, 12000, 16384, 16500, 20000, 30000, 50000, 100000]
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
dp = SlideShardedDP(model, {"lr": 1e-4, "weight_decay": 1e-5})
print(f"{'patches':>8} {'ms/step':>9} {'slides/s':>10} {'us per 1k patches':>18}")
for n in sizes:
    slides = []
    for b in range(2):
        g = torch.Generator(device=dev).manual_seed(1000 + b)
        slides.append((torch.randn(n, 1024, device=dev, generator=g), torch.tensor([1.0], device=dev), torch.tensor([3], device=dev), torch.tensor([1], device=dev)))
    steps = max(30, min(3000, int(1.0 / max(n * 2.2e-8, 2e-4))))
    for i in range(10):
        dp.step([slides[i % 2]], 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        dp.step([slides[i % 2]], 1)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{n:8d} {ms:9.4f} {1e3 / ms:10.1f} {ms * 1e3 / (n / 1000):18.2f}", flush=True)
    del slides
