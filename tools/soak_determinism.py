"""Soak test: the fused step on the same inputs must give bitwise identical gradients every time (an LDS hazard or a barrier-protocol
slip in the persistent GEMM kernels would show up as a rare mismatch). usage: soak_determinism.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat
from toad_amd.dp import hip_slide_grad

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
torch.manual_seed(11)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
w = {k: v.detach() for k, v in model._weights().items()}
bad = 0
for n in (100000, 20000, 9999, 777, 300000):
    g = torch.Generator(device=dev).manual_seed(n)
    for dtype in (torch.float32, torch.float16):
        bag = torch.randn(n, 1024, device=dev, generator=g).to(dtype)
        slide = (bag, torch.tensor([1.0], device=dev), torch.tensor([3], device=dev), torch.tensor([1], device=dev))
        ref = {k: torch.zeros_like(v) for k, v in w.items()}
        l0 = hip_slide_grad(model, ref, slide, beta=0.0).clone()
        ref = {k: v.clone() for k, v in ref.items()}
        cur = {k: torch.zeros_like(v) for k, v in w.items()}
        r = max(20, reps * 20000 // n) if n > 20000 else reps
        mism = 0
        for i in range(r):
            l = hip_slide_grad(model, cur, slide, beta=0.0)
            if not torch.equal(l, l0) or any(not torch.equal(cur[k], ref[k]) for k in ref):
                mism += 1
        torch.cuda.synchronize()
        print(f"N={n} {str(dtype).split('.')[-1]}: {r} repeats, {mism} mismatches, loss {l0[0].item():.6f}", flush=True)
        bad += mism
print("SOAK", "FAILED" if bad else "OK")
