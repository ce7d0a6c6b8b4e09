"""TEST INFRASTRUCTURE ONLY — captures a short run of the REAL reference harness as a golden fixture.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/pin_train_against_reference.py [--write]

What it drives, unmodified, on CPU:
  * ``utils/core_utils_mtl_concat.py:189-259``  ``train_loop``  (2 epochs, batch size 1, ``get_optim``'s SGD branch,
    ``utils/utils.py:63-70``: momentum 0.9, lr / weight decay below),
  * ``utils/eval_utils_mtl_concat.py:65-177``   ``summary``     (per-slide probabilities, errors, AUCs, top-k),
on SLIDES seeded bags, with the reference model initialised from the closed-form parameters of the other
goldens. It then replays the same schedule with the oracle (oracle/toad_oracle.py fwd_bwd + torch Adam) and
asserts the two agree, repeats it with the reference model in fp64 (the yardstick: how far two correct fp32
implementations may drift over the schedule), and with --write stores the REFERENCE's results in tests/golden/toad_train_golden.npz
(data only: the bags are regenerated from their seeds by the tests).

Why SGD and not the reference's default Adam: Adam's update is lr * m / (sqrt(v) + eps) — sign-like for the first
steps — so roundoff-level gradient differences on near-zero elements become +-lr parameter differences. Measured
here on this schedule with lr 2e-4: the reference in fp32 vs the reference in fp64 differ by 3.0e-3 in parameters
(3.5e-2 in probabilities) after 60 Adam steps, i.e. an end-of-training comparison under Adam cannot separate a
correct implementation from an incorrect one. Under SGD the same comparison stays at 1e-5. The Adam update itself
is pinned step-wise (same gradients in -> same parameters out) by tests/test_gpu_model.py.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import toad_oracle as orc  # noqa: E402
from oracle.pin_against_reference import import_reference, strided_sample  # noqa: E402

# ---- the schedule (the tests rebuild it from these constants, stored in the fixture's `meta`) ----------------
N_CLASSES = 6
SLIDES = 30
EPOCHS = 2
LR = 5e-4            # 5x the reference default (main_mtl_concat.py:85): 60 steps move parameters by ~1e-2 >> tolerance
REG = 1e-5           # reference default (main_mtl_concat.py:87)
SEED0 = 7000


def slide(i: int):
    """(data [N,1024], label, site, sex) of slide i — pure function of i (torch CPU generator)."""
    n = 40 + (i * 37) % 260
    g = torch.Generator().manual_seed(SEED0 + i)
    return torch.randn(n, 1024, generator=g), i % N_CLASSES, (i // 3) % 2, i % 2


class _Slides(torch.utils.data.Dataset):
    def __init__(self):
        import pandas as pd
        self.slide_data = pd.DataFrame({"slide_id": ["slide_%d" % i for i in range(SLIDES)]})

    def __len__(self):
        return SLIDES

    def __getitem__(self, i):
        return slide(i)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    RefModel = import_reference()
    sys.modules["torchsummary"].summary = lambda *a, **k: None        # imported by models/resnet_custom.py:5, never called here
    from utils.core_utils_mtl_concat import train_loop            # type: ignore
    from utils.eval_utils_mtl_concat import summary               # type: ignore
    from utils.utils import collate_MIL_mtl_concat, get_optim     # type: ignore
    import utils.eval_utils_mtl_concat as ref_eval                # type: ignore

    # eval_utils:61 calls .view(-1) on a slice of a transposed tensor, which torch >= 1.7 rejects (the reference
    # pins torch 1.5.1). The pinning run swaps in the same formula with .reshape(-1); nothing else is touched.
    def accuracy_reshape(output, target, topk=(1,)):
        maxk = max(topk)
        pred = output.topk(maxk, 1, True, True)[1].t()
        correct = pred.eq(target.view(1, -1).expand_as(pred))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(1.0 / target.size(0)) for k in topk]
    ref_eval.accuracy = accuracy_reshape
    from torch.utils.data import DataLoader, SequentialSampler

    params0 = orc.closed_form_params(N_CLASSES)
    model = RefModel(dropout=False, n_classes=N_CLASSES)
    model.load_state_dict(params0, strict=True)
    ds = _Slides()
    loader = DataLoader(ds, batch_size=1, sampler=SequentialSampler(ds), collate_fn=collate_MIL_mtl_concat)
    hp = types.SimpleNamespace(opt="sgd", lr=LR, reg=REG, n_classes=N_CLASSES, micro_average=False)
    optimizer = get_optim(model, hp)
    loss_fn = torch.nn.CrossEntropyLoss()
    epoch_lines = []
    for epoch in range(EPOCHS):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            train_loop(epoch, model, loader, optimizer, N_CLASSES, None, loss_fn)
        m = re.search(r"Epoch: (\d+), cls train_loss: ([\d.]+), cls train_error: ([\d.]+)", buf.getvalue())
        epoch_lines.append((float(m.group(2)), float(m.group(3))))
        print("  reference epoch %d: cls train_loss %.4f  cls train_error %.4f" % (epoch, *epoch_lines[-1]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = summary(model, loader, hp)
    final = {k: v.detach().clone() for k, v in model.state_dict().items()}
    print("  reference summary: cls_err %.4f cls_auc %.4f site_err %.4f site_auc %.4f top1/3/5 %.3f %.3f %.3f" % (
        res["cls_test_error"], res["cls_auc"], res["site_test_error"], res["site_auc"],
        res["top1_acc"], res["top3_acc"], res["top5_acc"]))

    # ---- the oracle on the same schedule: fwd_bwd + torch Adam on plain tensors
    p = {k: v.clone().requires_grad_(False) for k, v in params0.items()}
    plist = [torch.nn.Parameter(p[k]) for k in orc.PARAM_KEYS]
    opt = torch.optim.SGD(plist, lr=LR, momentum=0.9, weight_decay=REG)
    for epoch in range(EPOCHS):
        for i in range(SLIDES):
            x, label, site, sex = slide(i)
            cur = {k: q.detach() for k, q in zip(orc.PARAM_KEYS, plist)}
            _, _, grads = orc.fwd_bwd(cur, x, torch.tensor([float(sex)]), torch.tensor([label]), torch.tensor([site]))
            for k, q in zip(orc.PARAM_KEYS, plist):
                q.grad = grads[k].clone()
            opt.step()
    cur = {k: q.detach() for k, q in zip(orc.PARAM_KEYS, plist)}
    worst_p = max((cur[k] - final[k]).abs().max().item() for k in orc.PARAM_KEYS)
    moved = max((final[k] - params0[k]).abs().max().item() for k in orc.PARAM_KEYS)
    probs = np.concatenate([res["patient_results"]["slide_%d" % i]["cls_prob"] for i in range(SLIDES)])
    site_probs = np.concatenate([res["patient_results"]["slide_%d" % i]["site_prob"] for i in range(SLIDES)])
    worst_o = 0.0
    for i in range(SLIDES):
        x, label, site, sex = slide(i)
        o, _ = orc.forward(cur, x, torch.tensor([float(sex)]))
        worst_o = max(worst_o, np.abs(o["Y_prob"].numpy() - probs[i]).max(), np.abs(o["site_prob"].numpy() - site_probs[i]).max())
    print("  oracle vs reference after %d SGD steps: params max-abs %.2e (moved by %.2e), probabilities max-abs %.2e" % (
        EPOCHS * SLIDES, worst_p, moved, worst_o))
    assert worst_p <= 1e-5 and worst_o <= 1e-5, (worst_p, worst_o)

    # ---- yardstick: the reference model in fp64 on the same schedule
    m64 = RefModel(dropout=False, n_classes=N_CLASSES).double()
    m64.load_state_dict({k: v.double() for k, v in params0.items()}, strict=True)
    o64 = torch.optim.SGD(m64.parameters(), lr=LR, momentum=0.9, weight_decay=REG)
    m64.train()
    for epoch in range(EPOCHS):
        for i in range(SLIDES):
            x, label, site, sex = slide(i)
            r = m64(x.double(), torch.tensor([float(sex)], dtype=torch.float64))
            loss = loss_fn(r["logits"], torch.tensor([label])) * 0.75 + loss_fn(r["site_logits"], torch.tensor([site])) * 0.25
            loss.backward(); o64.step(); o64.zero_grad()
    m64.eval()
    f64 = m64.state_dict()
    dev_p = max((final[k].double() - f64[k]).abs().max().item() for k in orc.PARAM_KEYS)
    dev_o = 0.0
    with torch.no_grad():
        for i in range(SLIDES):
            x, label, site, sex = slide(i)
            r = m64(x.double(), torch.tensor([float(sex)], dtype=torch.float64))
            dev_o = max(dev_o, np.abs(r["Y_prob"].numpy() - probs[i]).max(), np.abs(r["site_prob"].numpy() - site_probs[i]).max())
    print("  reference fp32 vs reference fp64: params max-abs %.2e, probabilities max-abs %.2e" % (dev_p, dev_o))

    if args.write:
        w = {"meta": np.array([N_CLASSES, SLIDES, EPOCHS, LR, REG, SEED0], dtype=np.float64),
             "epoch_cls_loss_err": np.array(epoch_lines, dtype=np.float64),              # printed with 4 decimals
             "cls_prob": probs.astype(np.float32), "site_prob": site_probs.astype(np.float32),
             "Y_hat": np.asarray(res["df"]["Y_hat"]).astype(np.int64), "site_hat": np.asarray(res["df"]["site_hat"]).astype(np.int64),
             "scalars": np.array([res["cls_test_error"], res["cls_auc"], res["site_test_error"], res["site_auc"],
                                  res["top1_acc"], res["top3_acc"], res["top5_acc"]], dtype=np.float64),
             "cls_aucs": np.array(res["cls_aucs"], dtype=np.float64),
             "moved": np.float64(moved), "dev64_params": np.float64(dev_p), "dev64_probs": np.float64(dev_o)}
        for k in orc.PARAM_KEYS:
            w["final_sample/" + k] = strided_sample(final[k])
            w["final_l2/" + k] = np.float64(final[k].double().norm().item())
        out = os.path.join(REPO, "tests", "golden", "toad_train_golden.npz")
        np.savez_compressed(out, **w)
        print("  wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
