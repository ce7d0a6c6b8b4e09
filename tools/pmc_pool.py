"""Launch the fused pool forward and the pool backward of the whole-slide path (no dH_pool output, abs-max bound on) a few times at
N = 100k, for separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
h = rn(N, 512).relu(); p = rn(N, 768); wc = rn(2, 384) * 0.1; bc = rn(2); dm = rn(2, 512)
for _ in range(3):
    a_raw, m, stats = ops.gated_pool_fwd(p, 384, h, wc, bc)
    ops.gated_pool_bwd(p, 384, h, wc, a_raw, stats, m, dm, want_dh=False, want_amax=True)
torch.cuda.synchronize()
