import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
N = 100000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x = rn(N, 1024); w1 = rn(512, 1024) * 0.03; b1 = rn(512)
dp = rn(N, 768); dh = rn(N, 512); h = rn(N, 512); wabt = rn(512, 768)
for _ in range(3):
    ops.linear_act_fwd(x, w1, b1, 1)
    ops.linear_dgrad(dp, wabt, dh, h)
    ops.linear_wgrad(dh, x)
torch.cuda.synchronize()
