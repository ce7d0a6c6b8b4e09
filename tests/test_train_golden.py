"""The captured run of the REAL reference harness (oracle/pin_train_against_reference.py: the reference's
train_loop for 2 epochs with its SGD optimiser, then its eval summary) as a golden:
  * CPU: the oracle replays the schedule and lands on the reference's parameters / probabilities; this repo's
    eval.summary, fed the reference's per-slide probabilities through a stub model, reproduces the reference's
    errors, AUCs and top-k accuracies exactly.
  * GPU: toad_amd.train.train_loop + toad_amd.eval.summary on the HIP kernels reproduce the run.
Tolerance: the north star's 1e-4, or 4x the reference's own fp32-vs-fp64 drift over the schedule where larger.
"""
import os
import types

import numpy as np
import pytest
import torch

from oracle import toad_oracle as orc
from helpers import strided_sample

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tg():
    g = np.load(os.path.join(REPO, "tests", "golden", "toad_train_golden.npz"), allow_pickle=False)
    c, slides, epochs, lr, reg, seed0 = g["meta"]
    return g, int(c), int(slides), int(epochs), float(lr), float(reg), int(seed0)


def make_slide(i, c, seed0):
    """Same pure function of i as oracle/pin_train_against_reference.py:slide (the fixture holds no inputs)."""
    n = 40 + (i * 37) % 260
    gen = torch.Generator().manual_seed(seed0 + i)
    return torch.randn(n, 1024, generator=gen), i % c, (i // 3) % 2, i % 2


def batches(c, slides, seed0):
    out = []
    for i in range(slides):
        x, label, site, sex = make_slide(i, c, seed0)
        out.append((x, torch.tensor([label]), torch.tensor([site]), torch.tensor([sex])))   # collate_MIL_mtl_concat: LongTensors
    return out


def check_run(g, final, probs, site_probs, res, tol_p, tol_o):
    for k in orc.PARAM_KEYS:
        assert np.abs(strided_sample(final[k]) - g["final_sample/" + k]).max() <= tol_p, k
        l2 = float(g["final_l2/" + k])
        assert abs(float(final[k].double().norm()) - l2) <= 1e-4 * l2 + 1e-6, k
    assert np.abs(probs - g["cls_prob"]).max() <= tol_o
    assert np.abs(site_probs - g["site_prob"]).max() <= tol_o
    if res is not None:
        # discrete outputs: equal wherever the reference's own top-2 margin exceeds the tolerance
        ref = g["cls_prob"]; srt = np.sort(ref, axis=1)
        firm = (srt[:, -1] - srt[:, -2]) > 4 * tol_o
        assert (np.asarray(res["df"]["Y_hat"])[firm] == g["Y_hat"][firm]).all()
        sc = g["scalars"]
        if firm.all():
            assert abs(res["cls_test_error"] - sc[0]) < 1e-12 and abs(res["top1_acc"] - sc[4]) < 1e-6
        assert abs(res["cls_auc"] - sc[1]) <= 0.02 and abs(res["site_auc"] - sc[3]) <= 0.02   # rank statistics of 30 slides


def test_oracle_replays_reference_training_run_cpu(tg):
    g, c, slides, epochs, lr, reg, seed0 = tg
    plist = [torch.nn.Parameter(v.clone()) for v in (orc.closed_form_params(c)[k] for k in orc.PARAM_KEYS)]
    opt = torch.optim.SGD(plist, lr=lr, momentum=0.9, weight_decay=reg)                      # utils/utils.py:66-67
    data = batches(c, slides, seed0)
    losses = []
    for _ in range(epochs):
        tot = 0.0
        for x, label, site, sex in data:
            cur = {k: q.detach() for k, q in zip(orc.PARAM_KEYS, plist)}
            out, _, grads = orc.fwd_bwd(cur, x, sex.float(), label, site)
            tot += torch.nn.functional.cross_entropy(out["logits"], label).item()
            for k, q in zip(orc.PARAM_KEYS, plist):
                q.grad = grads[k].clone()
            opt.step()
        losses.append(tot / slides)
    cur = {k: q.detach() for k, q in zip(orc.PARAM_KEYS, plist)}
    outs = [orc.forward(cur, x, sex.float())[0] for x, _, _, sex in data]
    probs = np.concatenate([o["Y_prob"].numpy() for o in outs]); sp = np.concatenate([o["site_prob"].numpy() for o in outs])
    check_run(g, cur, probs, sp, None, 1e-5, 1e-5)
    assert np.abs(np.array(losses) - g["epoch_cls_loss_err"][:, 0]).max() <= 1e-4 + 5e-5    # the reference prints 4 decimals
    assert float(g["moved"]) > 100 * 1e-4                                                    # the run is not a no-op


class _Replay(torch.nn.Module):
    """Plays back recorded probabilities as the model's outputs (for the metric code only)."""

    def __init__(self, probs, site_probs):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.p, self.s, self.i = probs, site_probs, 0

    def forward(self, data, sex):
        p = torch.from_numpy(self.p[self.i:self.i + 1]); s = torch.from_numpy(self.s[self.i:self.i + 1])
        self.i += 1
        return {"logits": p.log(), "site_logits": s.log(), "Y_prob": p, "site_prob": s,
                "Y_hat": p.argmax(1, keepdim=True), "site_hat": s.argmax(1, keepdim=True)}


def test_summary_metrics_equal_reference_summary_cpu(tg):
    """eval.summary over the reference's own per-slide probabilities -> the reference's scalars, to the last digit."""
    from toad_amd.eval import accuracy, summary
    g, c, slides, epochs, lr, reg, seed0 = tg
    data = [(torch.zeros(2, 4), b[1], b[2], b[3]) for b in batches(c, slides, seed0)]
    args = types.SimpleNamespace(n_classes=c, micro_average=False)
    res = summary(_Replay(g["cls_prob"], g["site_prob"]), data, args, slide_ids=["slide_%d" % i for i in range(slides)])
    sc = g["scalars"]
    got = [res["cls_test_error"], res["cls_auc"], res["site_test_error"], res["site_auc"], res["top1_acc"], res["top3_acc"], res["top5_acc"]]
    assert np.abs(np.array(got) - sc).max() <= 1e-7, (got, sc.tolist())
    assert np.allclose(np.array(res["cls_aucs"]), g["cls_aucs"], atol=1e-12, equal_nan=True)
    assert (np.asarray(res["df"]["Y_hat"]) == g["Y_hat"]).all() and (np.asarray(res["df"]["site_hat"]) == g["site_hat"]).all()
    assert list(res["df"].columns) == ["slide_id", "sex", "Y", "Y_hat", "site", "site_hat"] + ["p_%d" % k for k in range(c)] + ["site_p"]
    pr = res["patient_results"]["slide_3"]
    assert set(pr) == {"slide_id", "cls_prob", "cls_label", "site_prob", "site_label"} and pr["cls_prob"].shape == (1, c)
    assert sum(n for _, _, n in res["loggers"][0].summary()) == slides
    # micro-average branch and the binary branch (eval_utils:137-158)
    args.micro_average = True
    r2 = summary(_Replay(g["cls_prob"], g["site_prob"]), data, args, slide_ids=list(range(slides)))
    assert 0.0 <= r2["cls_auc"] <= 1.0 and abs(r2["cls_auc"] - res["cls_auc"]) < 0.2
    two = types.SimpleNamespace(n_classes=2, micro_average=False)
    d2 = [(b[0], b[2], b[2], b[3]) for b in data]
    r3 = summary(_Replay(g["site_prob"], g["site_prob"]), d2, two, slide_ids=list(range(slides)))
    assert abs(r3["cls_auc"] - sc[3]) <= 1e-12 and "top1_acc" not in r3 and r3["cls_aucs"] == []
    one = [(b[0], torch.tensor([0]), torch.tensor([1]), b[3]) for b in data]                  # a single class present -> -1
    r4 = summary(_Replay(g["cls_prob"], g["site_prob"]), one, types.SimpleNamespace(n_classes=c), slide_ids=list(range(slides)))
    assert r4["cls_auc"] == -1 and r4["site_auc"] == -1
    top = accuracy(torch.tensor([[.1, .7, .2], [.5, .3, .2]]), torch.tensor([2, 0]), topk=(1, 2))
    assert abs(top[0].item() - 0.5) < 1e-7 and abs(top[1].item() - 1.0) < 1e-7


@pytest.mark.gpu
def test_hip_training_run_reproduces_reference_run(cuda, tg, tmp_path):
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.eval import attention_heatmap_scores, eval as eval_ckpt, summary
    from toad_amd.train import train_loop
    g, c, slides, epochs, lr, reg, seed0 = tg
    model = TOAD_fc_mtl_concat(n_classes=c)
    model.load_state_dict(orc.closed_form_params(c), strict=True)
    model.relocate()
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, model.parameters()), lr=lr, momentum=0.9, weight_decay=reg)
    data = batches(c, slides, seed0)
    stats = [train_loop(e, model, data, opt, c) for e in range(epochs)]
    args = types.SimpleNamespace(n_classes=c, micro_average=False, drop_out=False)
    ids = ["slide_%d" % i for i in range(slides)]
    res = summary(model, data, args, slide_ids=ids)
    final = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    probs = np.concatenate([res["patient_results"][s]["cls_prob"] for s in ids])
    sp = np.concatenate([res["patient_results"][s]["site_prob"] for s in ids])
    tol_p = max(1e-4, 4 * float(g["dev64_params"])); tol_o = max(1e-4, 4 * float(g["dev64_probs"]))
    check_run(g, final, probs, sp, res, tol_p, tol_o)
    assert np.abs(np.array([s["cls_loss"] for s in stats]) - g["epoch_cls_loss_err"][:, 0]).max() <= 1e-4 + 5e-5
    assert abs(stats[-1]["cls_error"] - g["epoch_cls_loss_err"][-1, 1]) <= 1.0 / slides + 5e-5
    # checkpoint round trip through the reference's eval entry points (eval_utils:19-46)
    ck = str(tmp_path / "s_0_checkpoint.pt")
    torch.save(model.state_dict(), ck)
    m2, r2 = eval_ckpt(data, args, ck, slide_ids=ids)
    p2 = np.concatenate([r2["patient_results"][s]["cls_prob"] for s in ids])
    assert np.array_equal(p2, probs) and not m2.training
    # heat-map scores = row 0 of the raw attention the full forward returns
    x = data[5][0].to(cuda)
    full = model(x, torch.ones(1, device=cuda))["A"][0]
    assert torch.equal(attention_heatmap_scores(model, x), full)
    pct = attention_heatmap_scores(model, x, percentile=True)
    assert float(pct.min()) == 0.0 and float(pct.max()) == 1.0 and torch.equal(pct.argsort(), full.argsort())
