"""Launch each hot kernel a few times at N=100k (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x = rn(N, 1024); w1 = rn(512, 1024) * 0.03; b1 = rn(512)
h = rn(N, 512).relu(); wab = rn(768, 512) * 0.04; bab = rn(768)
p = rn(N, 768); wc = rn(2, 384) * 0.1; bc = rn(2); dp = rn(N, 768); dh = rn(N, 512); dm = rn(2, 512)
wabt = ops.transpose(wab)
for _ in range(3):
    ops.linear_act_fwd(x, w1, b1, 1)
    ops.linear_act_fwd(h, wab, bab, 0)
    ops.linear_dgrad(dp, wabt, dh, h)
    ops.linear_wgrad(dp, h)
    ops.linear_wgrad(dh, x)
    a_raw, m, stats = ops.gated_pool_fwd(p, 384, h, wc, bc)
    ops.gated_pool_bwd(p, 384, h, wc, a_raw, stats, m, dm)
torch.cuda.synchronize()
