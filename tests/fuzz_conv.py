"""Random-shape parity sweep of the extractor's convolution / residual-GEMM entry points against fp64 (run on a GPU box (tests/test_gpu_fuzz.py runs a short fixed-seed sweep),
`python tests/fuzz_conv.py [cases] [seed]`). Exercises the kernel selection of gemm_f32.hip: halo (stride-1 3x3 on power-of-two widths), streamed
(everything else narrow, odd stage counts, strides), 256x256 (wide), with ragged M / Cout, several images per tile and tiles per image."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from toad_amd import ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda:0")
worst = 0.0
for i in range(cases):
    k = rng.choice([1, 3, 3, 3])
    s = rng.choice([1, 1, 1, 2])
    p = 0 if k == 1 else 1
    if rng.random() < 0.6:
        w = rng.choice([8, 16, 32, 64, 128]); th = max(1, 256 // w); h = th * rng.randint(1, 3)
    else:
        h, w = rng.randint(1, 40), rng.randint(1, 40)
    b = rng.randint(1, 5)
    cin = rng.choice([32, 64, 96, 128, 160, 256])
    cout = rng.choice([4, 32, 64, 100, 128, 192, 256, 300])
    res = rng.random() < 0.4
    act = rng.choice([0, 1, 1])
    if (h + 2 * p - k) // s + 1 < 1 or (w + 2 * p - k) // s + 1 < 1:
        continue
    g = torch.Generator().manual_seed(seed * 1000 + i)
    x = torch.randn(b, h, w, cin, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    r = torch.randn(b, ho, wo, cout, generator=g) if res else None
    wf = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    try:
        y = ops.conv_nhwc(x.to(dev), wf.to(dev), bias.to(dev), None if r is None else r.to(dev), k, k, s, p, act).cpu()
    except RuntimeError as e:
        print(f"case {i}: b{b} {h}x{w} cin{cin} cout{cout} k{k} s{s} res{int(res)}: REFUSED ({str(e)[:80]})")
        continue
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), stride=s, padding=p).permute(0, 2, 3, 1)
    if r is not None:
        ref = ref + r.double()
    if act:
        ref = ref.clamp_min(0)
    err = (y.double() - ref).abs().max().item()
    worst = max(worst, err)
    flag = "" if err <= 2e-5 else "   <<<<<< FAIL"
    print(f"case {i}: b{b} {h}x{w} cin{cin} cout{cout} k{k} s{s} res{int(res)} act{act}: max err {err:.2e}{flag}")
print(f"worst {worst:.2e} over {cases} cases (bound 2e-5)")
sys.exit(0 if worst <= 2e-5 else 1)
