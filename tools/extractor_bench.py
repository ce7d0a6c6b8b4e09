"""Times the ResNet-50-trunc extractor on the GPU: tiles/s and effective TFLOP/s (8.556 GFLOP per 256x256 tile).
usage: python tools/extractor_bench.py [batch] [iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd.resnet_custom import resnet50_baseline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = resnet50_baseline().relocate().eval()
x = torch.randn(B, 3, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        m(x)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"B={B}: {ms:.3f} ms/batch  {B / ms * 1e3:.0f} tiles/s  {8.556e9 * B / ms / 1e9:.1f} TFLOP/s effective")
