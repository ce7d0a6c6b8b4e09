"""One convolution layer of the extractor, a few launches: for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes and kernel traces.
usage: python tools/conv_traffic.py B H W Cin Cout k stride pad [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
b, h, w, cin, cout, k, s, p = (int(v) for v in sys.argv[1:9])
iters = int(sys.argv[9]) if len(sys.argv) > 9 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(b, h, w, cin, device=dev, generator=g)
wf = torch.randn(cout, k * k * cin, device=dev, generator=g) / (k * k * cin) ** 0.5
bias = torch.randn(cout, device=dev, generator=g)
for _ in range(2):
    ops.conv_nhwc(x, wf, bias, None, k, k, s, p, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = ops.conv_nhwc(x, wf, bias, None, k, k, s, p, 1)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
m = y.numel() // cout
print(f"conv {b}x{h}x{w}x{cin} -> {cout}, {k}x{k}/{s}: {us:.1f} us per call (incl. weight split + abs-max)  "
      f"{2 * m * k * k * cin * cout / us / 1e6:.1f} TF-eq  in {x.numel() * 4 / 1e6:.0f} MB out {y.numel() * 4 / 1e6:.0f} MB")
