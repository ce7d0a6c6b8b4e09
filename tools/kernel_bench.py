"""Per-kernel timing on one MI355X with HIP events (torch.cuda.Event on the current stream, which
is the stream the kernels are launched on). Prints achieved TF/s / GB/s against the rooflines.

    python tools/kernel_bench.py [N]
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from toad_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


x = rn(N, 1024); w1 = rn(512, 1024) * 0.03; b1 = rn(512)
h1 = rn(N, 512).relu(); w2 = rn(512, 512) * 0.04; wab = rn(768, 512) * 0.04; bab = rn(768)
h = rn(N, 512).relu(); p = rn(N, 768); wc = rn(2, 384) * 0.1; bc = rn(2)
dp = rn(N, 768); dh = rn(N, 512)

rows = []
def gemm(name, fn, flops):
    t = timeit(fn); rows.append((name, t * 1e6, f"{flops / t / 1e12:7.1f} TF/s  ({flops / t / 157.3e12 * 100:5.1f}% of 157.3)"))
def mem(name, fn, nbytes):
    t = timeit(fn); rows.append((name, t * 1e6, f"{nbytes / t / 1e9:7.0f} GB/s  ({nbytes / t / 8e12 * 100:5.1f}% of 8 TB/s)"))

gemm("fwd  X[N,1024]->512 +relu", lambda: ops.linear_act_fwd(x, w1, b1, 1), 2 * N * 1024 * 512)
gemm("fwd  H1[N,512]->512 +relu", lambda: ops.linear_act_fwd(h1, w2, b1, 1), 2 * N * 512 * 512)
gemm("fwd  H[N,512]->768", lambda: ops.linear_act_fwd(h, wab, bab, 0), 2 * N * 512 * 768)
wabt = ops.transpose(wab); w2t = ops.transpose(w2)
gemm("dgrad dP[N,768]->512 +add+mask", lambda: ops.linear_dgrad(dp, wabt, dh, h), 2 * N * 768 * 512)
gemm("dgrad dZ2[N,512]->512 +mask", lambda: ops.linear_dgrad(dh, w2t, None, h1), 2 * N * 512 * 512)
gemm("wgrad dWab[768,512]", lambda: ops.linear_wgrad(dp, h), 2 * N * 768 * 512)
gemm("wgrad dW2[512,512]", lambda: ops.linear_wgrad(dh, h1), 2 * N * 512 * 512)
gemm("wgrad dW1[512,1024]", lambda: ops.linear_wgrad(dh, x), 2 * N * 512 * 1024)
a_raw, m, stats = ops.gated_pool_fwd(p, 384, h, wc, bc)
mem("gated_pool_fwd (fused)", lambda: ops.gated_pool_fwd(p, 384, h, wc, bc), 4 * (N * (768 + 512 + 2) + 2 * 384 + 2 + 2 * 512))
mem("gated scores only", lambda: ops.gated_pool_fwd(p, 384, None, wc, bc), 4 * N * (768 + 2))
dm = rn(2, 512)
mem("gated_pool_bwd", lambda: ops.gated_pool_bwd(p, 384, h, wc, a_raw, stats, m, dm), 4 * N * (2 * 512 + 4 * 384 + 2))
cp = torch.empty_like(x)
mem("torch copy_ X (HBM ceiling ref)", lambda: cp.copy_(x), 2 * x.numel() * 4)
print(f"N = {N}")
for name, us, s in rows:
    print(f"{name:34s} {us:10.1f} us   {s}")
