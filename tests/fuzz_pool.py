"""Random shapes through the fused gated-attention pool forward / backward against the CPU oracle (the checks of tests/test_gpu_kernels.py over many
(N, D, L, n_tasks)). Not collected by pytest: `python tests/fuzz_pool.py [cases] [seed]` on a GPU box."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops
from oracle import toad_oracle as orc           # checker only
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda:0")
def rel(a, b):        # relative to the reference's scale; a reference that is exactly zero (one-patch bags: softmax of a single score has no gradient) -> absolute, per 1e-2
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-2)).item()
nfail = 0
for i in range(cases):
    d = rng.choice([384, 256, 128, 512, 4, 100, 260, rng.randint(1, 128) * 4])
    l = rng.choice([512, 1024, 640, 8, 200, 520, rng.randint(1, 256) * 4])
    t = rng.randint(1, 4)
    n = rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 4097, rng.randint(1, 20000)])
    g = torch.Generator().manual_seed(seed * 104729 + i)
    p = torch.randn(n, 2 * d, generator=g) * rng.choice([1.0, 1.0, 3.0])
    h = torch.randn(n, l, generator=g).relu()
    wc = torch.randn(t, d, generator=g) * 0.1; bc = torch.randn(t, generator=g) * 0.1
    msgs, ok = [], True
    try:
        a_raw, m, stats = ops.gated_pool_fwd(p.to(dev), d, h.to(dev), wc.to(dev), bc.to(dev))
        ra, rm = orc.gated_pool_fwd(p[:, :d], p[:, d:], h, wc, bc)
        e1, e2 = (a_raw.cpu() - ra).abs().max().item(), (m.cpu() - rm).abs().max().item()
        msgs.append(f"fwd A {e1:.1e} M {e2:.1e}"); ok = ok and e1 <= 2e-5 and e2 <= 2e-5
        dm = torch.randn(t, l, generator=g)
        dp, dh, dwc, dbc = ops.gated_pool_bwd(p.to(dev), d, h.to(dev), wc.to(dev), a_raw, stats, m, dm.to(dev))
        rpa, rpb, rdh, rdwc, rdbc = orc.gated_pool_bwd(p[:, :d], p[:, d:], h, wc, ra, rm, dm)
        es = (rel(dp.cpu(), torch.cat([rpa, rpb], 1)), rel(dh.cpu(), rdh), rel(dwc.cpu(), rdwc))
        # one-patch bags: the exact attention gradient is zero (softmax of a single score); what a kernel returns is the round-off of
        # H.dM - M.dM, two fp32 sums of ~|H||dM| sqrt(L) in different orders (tests/helpers.py gives those cases a scale-based floor)
        msgs.append("bwd dP %.1e dH %.1e dWc %.1e" % es); ok = ok and max(es) <= (1e-2 if n == 1 else 1e-4)
    except RuntimeError as ex:
        msgs.append("REFUSED " + str(ex)[:80])
    nfail += 0 if ok else 1
    print(f"case {i}: N{n} D{d} L{l} T{t}: " + "  ".join(msgs) + ("" if ok else "   <<<<<< FAIL"), flush=True)
print(f"{nfail} failures over {cases} cases")
sys.exit(1 if nfail else 0)
