"""TEST INFRASTRUCTURE ONLY — BASELINE config 1 as stated, pinned STEP BY STEP to the real reference harness.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/pin_config1_against_reference.py [--write]

BASELINE.json config 1 = the reference's own CPU-runnable case: 18 classes (main_mtl_concat.py:141), 256-patch x 1024-d random bags,
the harness defaults of main_mtl_concat.py:85-101 - Adam (`--opt adam`), lr 1e-4, weight decay 1e-5, no dropout, seed 1 - driven
through the unmodified ``utils/core_utils_mtl_concat.py:189-259 train_loop`` with ``utils/utils.py:63-70 get_optim`` and the model's
own ``initialize_weights`` start (utils/utils.py:150-154 under ``torch.manual_seed(1)``). The committed end-of-run pin
(pin_train_against_reference.py) uses SGD because an END-of-training comparison under Adam is meaningless (sign-like first updates);
this one pins Adam the only way that is meaningful - per step, for the first STEPS optimiser steps:

  * after every step: the two losses the loop computed for that slide, Y_hat / site_hat, and strided samples of all 14 parameters;
  * the same schedule with the reference model in fp64 (the yardstick: how far two correct fp32 runs may be apart at step j);
  * the oracle (oracle/toad_oracle.py fwd_bwd + torch.optim.Adam on plain tensors) replayed on the same schedule must agree with the
    reference at every step - asserted here, so the fixture pins the oracle too.

With --write the REFERENCE's per-step values go to tests/golden/toad_config1_golden.npz (data only; bags and the initial parameters are
regenerated from their seeds by the tests).
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import toad_oracle as orc  # noqa: E402
from oracle.pin_against_reference import import_reference, strided_sample  # noqa: E402

N_CLASSES = 18       # main_mtl_concat.py:141
PATCHES = 256        # BASELINE.json configs[0]
STEPS = 12           # optimiser steps pinned (one slide each, batch size 1: utils/utils.py:51-55)
LR, REG, SEED = 1e-4, 1e-5, 1    # main_mtl_concat.py:85-89 defaults
MAX_FRAC = 1e-3      # at most this fraction of the parameters may be further than TIGHT from the reference at any pinned step
TIGHT = 2e-6         # |parameter difference| that counts as "the same" (2 % of one Adam step of size lr)
BAG_SEED0 = 0        # slide i's bag = randn(256, 1024) under torch.Generator().manual_seed(BAG_SEED0 + i)


def slide(i: int):
    """(data [256,1024], label, site, sex) of slide i: SURVEY.md 8(d) synthetic labels, label = i mod 18, site = i mod 2, sex = (i//2) mod 2."""
    g = torch.Generator().manual_seed(BAG_SEED0 + i)
    return torch.randn(PATCHES, 1024, generator=g), i % N_CLASSES, i % 2, (i // 2) % 2


class _Slides(torch.utils.data.Dataset):
    def __len__(self):
        return STEPS

    def __getitem__(self, i):
        return slide(i)


def initial_params(RefModel=None):
    """The reference's own initialisation under seed 1 (xavier_normal_ weights, zero biases). Without the reference (tests): the oracle's
    restatement of it, which this script asserts to be bitwise the same."""
    if RefModel is None:
        return orc.xavier_params_like_reference(N_CLASSES, SEED)
    torch.manual_seed(SEED)
    m = RefModel(dropout=False, n_classes=N_CLASSES)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    RefModel = import_reference()
    sys.modules["torchsummary"].summary = lambda *a, **k: None
    from utils.core_utils_mtl_concat import train_loop            # type: ignore
    from utils.utils import collate_MIL_mtl_concat, get_optim     # type: ignore
    from torch.utils.data import DataLoader, SequentialSampler

    params0 = initial_params(RefModel)
    p_or = orc.xavier_params_like_reference(N_CLASSES, SEED)
    assert all(torch.equal(params0[k], p_or[k]) for k in orc.PARAM_KEYS), "oracle initialisation differs from the reference's under seed 1"
    hp = types.SimpleNamespace(opt="adam", lr=LR, reg=REG, n_classes=N_CLASSES)

    def run_reference(dtype):
        model = RefModel(dropout=False, n_classes=N_CLASSES)
        model.load_state_dict(params0, strict=True)
        model = model.to(dtype)
        optimizer = get_optim(model, hp)
        rec = {"cls": [], "site": [], "yhat": [], "shat": [], "params": []}
        ce = torch.nn.CrossEntropyLoss()
        calls = {"n": 0}

        def loss_fn(logits, target):                # the loop calls it twice per slide: class loss, then site loss (core_utils:213-214)
            v = ce(logits, target)
            rec["cls" if calls["n"] % 2 == 0 else "site"].append(float(v.item()))
            (rec["yhat"] if calls["n"] % 2 == 0 else rec["shat"]).append(int(logits.argmax(1).item()))
            calls["n"] += 1
            return v
        step0 = optimizer.step

        def step(*a, **k):
            r = step0(*a, **k)
            rec["params"].append({kk: v.detach().clone() for kk, v in model.state_dict().items()})
            return r
        optimizer.step = step
        if dtype == torch.float32:
            ds = _Slides()
            loader = DataLoader(ds, batch_size=1, sampler=SequentialSampler(ds), collate_fn=collate_MIL_mtl_concat)
            with contextlib.redirect_stdout(io.StringIO()):
                train_loop(0, model, loader, optimizer, N_CLASSES, None, loss_fn)
        else:                                       # the fp64 yardstick: same loop body on double tensors (train_loop casts sex to float32)
            model.train()
            for i in range(STEPS):
                x, label, site, sex = slide(i)
                r = model(x.double(), torch.tensor([float(sex)], dtype=dtype))
                loss = loss_fn(r["logits"], torch.tensor([label])) * 0.75 + loss_fn(r["site_logits"], torch.tensor([site])) * 0.25
                loss.backward(); optimizer.step(); optimizer.zero_grad()
        return rec

    ref = run_reference(torch.float32)
    ref64 = run_reference(torch.float64)
    assert len(ref["params"]) == STEPS and len(ref["cls"]) == STEPS

    # ---- the oracle on the same schedule
    plist = [torch.nn.Parameter(params0[k].clone()) for k in orc.PARAM_KEYS]
    opt = torch.optim.Adam(plist, lr=LR, weight_decay=REG)
    worst_p, worst_l, dev_p, frac_p, frac64 = [], [], [], [], []
    for i in range(STEPS):
        x, label, site, sex = slide(i)
        cur = {k: q.detach() for k, q in zip(orc.PARAM_KEYS, plist)}
        out, _, grads = orc.fwd_bwd(cur, x, torch.tensor([float(sex)]), torch.tensor([label]), torch.tensor([site]))
        cls = torch.nn.functional.cross_entropy(out["logits"], torch.tensor([label])).item()
        st = torch.nn.functional.cross_entropy(out["site_logits"], torch.tensor([site])).item()
        for k, q in zip(orc.PARAM_KEYS, plist):
            q.grad = grads[k].clone()
        opt.step()
        worst_l.append(max(abs(cls - ref["cls"][i]), abs(st - ref["site"][i])))
        worst_p.append(max((q.detach() - ref["params"][i][k]).abs().max().item() for k, q in zip(orc.PARAM_KEYS, plist)))
        dev_p.append(max((ref["params"][i][k].double() - ref64["params"][i][k]).abs().max().item() for k in orc.PARAM_KEYS))
        d_or = torch.cat([(q.detach() - ref["params"][i][k]).abs().reshape(-1) for k, q in zip(orc.PARAM_KEYS, plist)])
        d_64 = torch.cat([(ref["params"][i][k].double() - ref64["params"][i][k]).abs().reshape(-1) for k in orc.PARAM_KEYS])
        frac_p.append(float((d_or > TIGHT).float().mean())); frac64.append(float((d_64 > TIGHT).float().mean()))
    moved = max((ref["params"][-1][k] - params0[k]).abs().max().item() for k in orc.PARAM_KEYS)
    print("  reference (Adam, lr %g, wd %g, %d steps of %d-patch bags, %d classes): cls losses %s" % (LR, REG, STEPS, PATCHES, N_CLASSES,
          " ".join("%.4f" % v for v in ref["cls"])))
    print("  oracle vs reference per step: losses max-abs %.2e; parameters max-abs per step %s" % (max(worst_l), " ".join("%.1e" % v for v in worst_p)))
    print("  reference fp32 vs fp64 per step (parameters max-abs): %s   (moved by %.2e in all)" % (" ".join("%.1e" % v for v in dev_p), moved))
    print("  fraction of the 1.19 M parameters further than %.0e from the reference: oracle %s | reference fp64 %s" % (
        TIGHT, " ".join("%.1e" % v for v in frac_p), " ".join("%.1e" % v for v in frac64)))
    assert max(worst_l) <= 2e-5
    # Adam's update is lr * m / (sqrt(v) + eps): sign-like. An element whose true gradient is below the round-off of its computation gets an
    # arbitrary sign, so ANY two correct fp32 runs differ by up to 2 lr per step on those elements (the reference in fp32 vs itself in fp64
    # does: dev64) - a max-abs bound cannot be tight. What can: every element stays inside the sign-flip envelope 2 lr (j + 1), and all but
    # a small fraction (the near-zero-gradient elements) agree to TIGHT.
    for j in range(STEPS):
        assert worst_p[j] <= 2.0 * LR * (j + 1) + 1e-7, (j, worst_p[j])
        assert frac_p[j] <= MAX_FRAC, (j, frac_p[j])

    if args.write:
        w = {"meta": np.array([N_CLASSES, PATCHES, STEPS, LR, REG, SEED, BAG_SEED0], dtype=np.float64),
             "cls_loss": np.array(ref["cls"]), "site_loss": np.array(ref["site"]),
             "Y_hat": np.array(ref["yhat"], dtype=np.int64), "site_hat": np.array(ref["shat"], dtype=np.int64),
             "dev64_params": np.array(dev_p), "moved": np.float64(moved), "frac_far_oracle": np.array(frac_p), "frac_far_ref64": np.array(frac64),
             "tight": np.float64(TIGHT), "max_frac": np.float64(MAX_FRAC),
             "dev64_loss": np.array([max(abs(a - b), abs(c - d)) for a, b, c, d in zip(ref["cls"], ref64["cls"], ref["site"], ref64["site"])])}
        for k in orc.PARAM_KEYS:
            w["param_sample/" + k] = np.stack([strided_sample(ref["params"][j][k]) for j in range(STEPS)])
            w["param_l2/" + k] = np.array([ref["params"][j][k].double().norm().item() for j in range(STEPS)])
        out = os.path.join(REPO, "tests", "golden", "toad_config1_golden.npz")
        np.savez_compressed(out, **w)
        print("  wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
