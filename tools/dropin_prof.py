"""Host-side cost of the drop-in training step (reference loop body, utils/core_utils_mtl_concat.py:201-234) at a small bag size:
wall time per phase WITHOUT device syncs in between (the GPU work is ~0.2 ms, the host path is what limits small bags), then cProfile.
usage: dropin_prof.py [patches] [steps]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import TOAD_fc_mtl_concat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
loss_fn = torch.nn.CrossEntropyLoss()
data = torch.randn(n, 1024, device=dev); sex = torch.tensor([1.0], device=dev)
label = torch.tensor([3], device=dev); site = torch.tensor([1], device=dev)


def loop(opt, steps, acc=None):
    for _ in range(steps):
        t0 = time.perf_counter()
        res = model(data, sex)
        t1 = time.perf_counter()
        loss = loss_fn(res["logits"], label) * 0.75 + loss_fn(res["site_logits"], site) * 0.25
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        opt.step()
        t4 = time.perf_counter()
        opt.zero_grad()
        t5 = time.perf_counter()
        if acc is not None:
            for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                acc[i] += d


for name, mk in (("torch.optim.Adam (default foreach)", lambda: torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)),
                 ("torch.optim.Adam(fused=True)", lambda: torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True))):
    opt = mk()
    loop(opt, 20); torch.cuda.synchronize()
    acc = [0.0] * 5
    t0 = time.perf_counter(); loop(opt, steps, acc); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name}: {dt * 1e3:.3f} ms/step; host us: forward {acc[0] / steps * 1e6:.0f}  losses {acc[1] / steps * 1e6:.0f}  backward {acc[2] / steps * 1e6:.0f}  "
          f"optimizer {acc[3] / steps * 1e6:.0f}  zero_grad {acc[4] / steps * 1e6:.0f}")
from types import SimpleNamespace
from toad_amd.optim import get_optim
from toad_amd.train import train_loop
batches = [(data, label, site, sex)] * steps
for flat in (False, True):
    o = get_optim(model, SimpleNamespace(opt="adam", lr=1e-4, reg=1e-5), flat=flat)
    train_loop(0, model, batches[:20], o, 18); torch.cuda.synchronize()
    t0 = time.perf_counter(); train_loop(0, model, batches, o, 18); torch.cuda.synchronize()
    print(f"toad_amd.train.train_loop (fused slide step, {'FlatAdam' if flat else 'torch.optim.Adam'}; incl. the loggers): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
    t0 = time.perf_counter(); train_loop(0, model, batches, o if not flat else get_optim(model, SimpleNamespace(opt="adam", lr=1e-4, reg=1e-5), flat=False), 18, fused=False); torch.cuda.synchronize()
    print(f"toad_amd.train.train_loop (fused=False, torch.optim.Adam; incl. the loggers): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
loop(opt, 20)
pr = cProfile.Profile(); pr.enable(); loop(opt, 200); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
