"""Random-shape parity sweep of the per-op GEMM entry points (forward + ReLU bit image + abs-max, dgrad with the bit mask, wgrad on fp32 and
prepared operands) against fp64. Not collected by pytest (no test_ prefix): run on a GPU box, `python tests/fuzz_gemm.py [cases] [seed]`.
Shapes straddle the 256-row / 256-column tile edges, the K-split remainders of the persistent kernels and the generic fallback (K % 32 != 0)."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda:0")
worst = 0.0
nfail = 0
def rel(a, ref):
    return ((a.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
for i in range(cases):
    m = rng.choice([1, 7, 64, 255, 256, 257, 511, 513, 1000, 2049, 4097, 10000, 33000, 70001, rng.randint(1, 5000)])
    k = rng.choice([32, 64, 96, 128, 512, 1024, 40, 100, 1000, rng.randint(1, 40) * 32])
    n = rng.choice([4, 64, 128, 256, 384, 512, 768, 1024, 20, 300, rng.randint(1, 200) * 4])
    g = torch.Generator().manual_seed(seed * 7919 + i)
    x = torch.randn(m, k, generator=g); w = torch.randn(n, k, generator=g) / k ** 0.5; b = torch.randn(n, generator=g) * 0.1
    dy = torch.randn(m, n, generator=g)
    xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    msgs = []
    try:
        h2 = ops.h2_ok(m, n, k)
        if h2:
            y, ya, bits = ops.linear_act_fwd(xd, wd, bd, 1, want_bits=True)
        else:
            y, bits, ya = ops.linear_act_fwd(xd, wd, bd, 1), None, None
        ref = (x.double() @ w.double().t() + b.double()).clamp_min(0)
        e = rel(y, ref); msgs.append(f"fwd {e:.1e}")
        ok = e <= 2e-5
        if ya is not None:
            blk = torch.stack([y[r:r + 256].abs().max() for r in range(0, m, 256)])
            ok = ok and torch.equal(ya[: blk.numel()].cpu(), blk.cpu())
            msgs.append("amax " + ("ok" if torch.equal(ya[: blk.numel()].cpu(), blk.cpu()) else "BAD"))
        wt = ops.transpose(wd)                                   # [K, N]: dX = dY W
        # dgrad of THIS layer's input needs dY [M,N] . W [N,K]: wt = W^T is [K,N]
        dxr = (dy.double() @ w.double())
        dx = ops.linear_dgrad(dyd, wt)
        e = rel(dx, dxr); msgs.append(f"dgrad {e:.1e}"); ok = ok and e <= 2e-5
        dw, db = ops.linear_wgrad(dyd, xd)
        dwr = dy.double().t() @ x.double()
        e = rel(dw, dwr); msgs.append(f"wgrad {e:.1e}"); ok = ok and e <= 2e-5
        e = rel(db, dy.double().sum(0)); ok = ok and e <= 2e-5
        if k % 8 == 0 and m >= 64:
            bag = ops.prepare_bag(xd)
            dw2, _ = ops.linear_wgrad(dyd, bag)
            e = rel(dw2, dwr); msgs.append(f"wgrad(prepared) {e:.1e}"); ok = ok and e <= 2e-5
        # dgrad with this layer's ReLU mask as bits: mask rows of a [M,N] gradient need a weight with N as its OUTPUT dimension
        if bits is not None and ops.h2_ok(m, n, n):
            w2 = torch.randn(n, n, generator=g) / n ** 0.5
            dz = torch.randn(m, n, generator=g)
            dm = ops.linear_dgrad(dz.to(dev), ops.transpose(w2.to(dev)), relu_src=y, relu_bits=bits)
            dmr = (dz.double() @ w2.double()) * (y.cpu() > 0)
            e = rel(dm, dmr); msgs.append(f"dgrad(bits) {e:.1e}"); ok = ok and e <= 2e-5
    except RuntimeError as ex:
        msgs.append("REFUSED " + str(ex)[:90]); ok = True
    if not ok:
        nfail += 1
    print(f"case {i}: M{m} K{k} N{n}: " + "  ".join(msgs) + ("" if ok else "   <<<<<< FAIL"))
print(f"{nfail} failures over {cases} cases")
sys.exit(1 if nfail else 0)
