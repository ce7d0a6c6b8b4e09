"""A/B of the pooling backward at N = 100k: dH written or not, abs-max bound emitted or not (HIP events, 20 launches each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
dev = torch.device("cuda:0")
n, d, l, t = 100000, 384, 512, 2
g = torch.Generator(device=dev).manual_seed(0)
p = torch.randn(n, 2 * d, device=dev, generator=g); h = torch.randn(n, l, device=dev, generator=g).relu_()
wc = torch.randn(t, d, device=dev, generator=g) * 0.05; bc = torch.zeros(t, device=dev)
a_raw, m, stats = ops.gated_pool_fwd(p, d, h, wc, bc)
dm = torch.randn(t, l, device=dev, generator=g) * 1e-3
dp = torch.empty_like(p); dh = torch.empty_like(h)
for want_dh in (True, False):
    for want_amax in (False, True):
        for _ in range(3):
            ops.gated_pool_bwd(p, d, h, wc, a_raw, stats, m, dm, dp=dp, dh=dh if want_dh else None, want_dh=want_dh, want_amax=want_amax)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gated_pool_bwd(p, d, h, wc, a_raw, stats, m, dm, dp=dp, dh=dh if want_dh else None, want_dh=want_dh, want_amax=want_amax)
        e1.record(); torch.cuda.synchronize()
        print(f"dH {'written' if want_dh else 'skipped'}, amax {'on ' if want_amax else 'off'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. partial-reduce launch{' + memset' if want_amax else ''})", flush=True)
