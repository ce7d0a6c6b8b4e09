"""Forward-only throughput of the validate / summary loop on small ragged bags: the reference's one model(data, sex) per slide
(eval_utils_mtl_concat.py:88-91) against the grouped pass (toad_amd.eval.forward_grouped -> TOAD_fc_mtl_concat.forward_many).
usage: eval_bench.py [n_slides] [min_patches] [max_patches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat
from toad_amd.eval import forward_grouped

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 128
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev = torch.device("cuda:0")
torch.manual_seed(3)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.eval()
g = torch.Generator().manual_seed(5)
lens = torch.randint(lo, hi + 1, (ns,), generator=g).tolist()
slides = [(torch.randn(n, 1024, device=dev), torch.tensor([1], device=dev), torch.tensor([0], device=dev), torch.tensor([1.0], device=dev)) for n in lens]


def run(group_rows):
    with torch.no_grad():
        for _ in forward_grouped(model, slides, group_rows):
            pass


for gr in (0, 8192, 32768, 131072):
    run(gr); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run(gr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"group_rows={gr:7d}: {ns / dt:9.1f} slides/s  {sum(lens) / dt / 1e6:7.2f} M patches/s  ({dt * 1e3 / ns:.3f} ms/slide, {ns} slides of {lo}..{hi} patches)")
