"""Times toad_linear_act_res_fwd_f32 on given shapes. usage: python tools/gemm_shape_bench.py M,K,N,res [M,K,N,res ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    m, k, n, res = (int(v) for v in spec.split(","))
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    for _ in range(3):
        ops.linear_act_res_fwd(x, w, b, r, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.linear_act_res_fwd(x, w, b, r, 1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    byts = (m * k + m * n * (2 if res else 1)) * 4
    print(f"M={m} K={k} N={n} res={res}: {us:8.1f} us  {2*m*k*n/us/1e6:6.1f} TF-eq  {byts/us/1e3:6.0f} GB/s algorithmic")
