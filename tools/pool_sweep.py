import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
N = 100000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
h = rn(N, 512).relu(); p = rn(N, 768); wc = rn(2, 384) * 0.1; bc = rn(2); dm = rn(2, 512)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
a_raw, m, stats = ops.gated_pool_fwd(p, 384, h, wc, bc)
tf = timeit(lambda: ops.gated_pool_fwd(p, 384, h, wc, bc))
tb = timeit(lambda: ops.gated_pool_bwd(p, 384, h, wc, a_raw, stats, m, dm))
print(f"TOAD_POOL_GRID={os.environ.get('TOAD_POOL_GRID','default')}: fwd {tf:.1f} us ({512.8e6/tf/1e6:.2f} TB/s)  bwd {tb:.1f} us ({1024.8e6/tb/1e6:.2f} TB/s)")
