"""Phase timing inside gemm_nt_h2_big_kernel (library built with -DTOAD_H2_TRACE; TOAD_HIP_LIB points at it): runs the 1024 -> 512 forward once
on a 100k-patch bag and prints, for waves 0 and 4 of workgroup 0, the cycles each of steps 8..23 spent in LOAD(.,0) work, barrier 1, COMPUTE(.,0),
barrier 2, LOAD(.,1) work, barrier 3, COMPUTE(.,1), barrier 4 (s_memtime, shader clock)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, K, device=dev, generator=g); w = torch.randn(512, K, device=dev, generator=g) * 0.03; b = torch.zeros(512, device=dev)
lib = _lib.load()
ws = ops._ws(lib.toad_linear_ws_bytes(N, 512, K), dev)
for _ in range(3):
    ops.linear_act_fwd(x, w, b, 1)
torch.cuda.synchronize()
off = 256 * 256 * 256 * 4 + 65536 * 4
tr = ws[off: off + 16 * 2 * 9 * 8].view(torch.int64).cpu().view(16, 2, 9)
names = ["L0 work", "bar1", "C0", "bar2", "L1 work", "bar3", "C1", "bar4"]
for wv in (0, 1):
    d = (tr[:, wv, 1:] - tr[:, wv, :-1]).double()
    print(f"wave {wv * 4}: mean cycles per phase over 16 steps:", "  ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(0).tolist())), f" | step {float((tr[:, wv, 8] - tr[:, wv, 0]).double().mean()):.0f}")
    print("          per step total:", [int(v) for v in (tr[:, wv, 8] - tr[:, wv, 0]).tolist()])
print("step period from the per-step start stamps (wave 0):", [int(tr[i + 1, 0, 0] - tr[i, 0, 0]) for i in range(15)])
base = int(tr[2, 0, 0])
for st in (2, 3):
    for wv in (0, 1):
        print(f"step {8 + st} wave {wv * 4}: stamps relative to wave 0's start of step 10:", [int(v) - base for v in tr[st, wv].tolist()])
