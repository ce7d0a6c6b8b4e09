"""The train/validate driver: bookkeeping on CPU with a stub model; the real loop on the GPU against the
oracle stepping the same slides with plain SGD."""
import numpy as np
import pytest
import torch

from oracle import toad_oracle as orc


class _Stub(torch.nn.Module):
    """Returns preset logits per call; has one parameter so loss.backward()/optimizer work (CPU)."""

    def __init__(self, table):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.table, self.i = table, 0

    def forward(self, data, sex):
        lg, sl = self.table[self.i % len(self.table)]
        self.i += 1
        lg = lg + self.w * 0
        return {"logits": lg, "site_logits": sl + self.w * 0, "Y_prob": torch.softmax(lg, 1), "site_prob": torch.softmax(sl, 1),
                "Y_hat": lg.argmax(1, keepdim=True), "site_hat": sl.argmax(1, keepdim=True)}


def test_bookkeeping_matches_reference_formulas_cpu():
    from toad_amd.train import train_loop, validate, _auc
    torch.manual_seed(0)
    table = [(torch.randn(1, 3), torch.randn(1, 2)) for _ in range(7)]
    labels = [0, 1, 2, 1, 0, 2, 2]; sites = [0, 1, 1, 0, 0, 1, 0]
    loader = [(torch.zeros(4, 8), torch.tensor([l]), torch.tensor([s]), torch.tensor([1.0])) for l, s in zip(labels, sites)]
    m = _Stub(table)
    r = train_loop(0, m, loader, torch.optim.SGD(m.parameters(), lr=0.0), 3)
    ce = torch.nn.functional.cross_entropy
    exp_cls = np.mean([ce(t[0], torch.tensor([l])).item() for t, l in zip(table, labels)])
    exp_err = np.mean([float(t[0].argmax().item() != l) for t, l in zip(table, labels)])
    assert abs(r["cls_loss"] - exp_cls) < 1e-6 and abs(r["cls_error"] - exp_err) < 1e-12 and r["slides"] == 7
    assert sum(c for _, _, c in r["cls_acc"]) == 7 and [c for _, _, c in r["cls_acc"]] == [2, 2, 3]
    m.i = 0
    v = validate(m, loader, 3)
    assert v["prob"].shape == (7, 3) and abs(v["cls_loss"] - exp_cls) < 1e-6
    from sklearn.metrics import roc_auc_score
    assert abs(v["site_auc"] - roc_auc_score(sites, v["site_prob"][:, 1])) < 1e-12
    assert 0.0 <= v["cls_auc"] <= 1.0
    assert abs(_auc(np.array([0, 1, 1, 0]), np.array([[.9, .1], [.2, .8], [.4, .6], [.7, .3]]), 2) - 1.0) < 1e-12


@pytest.mark.gpu
def test_train_loop_tracks_oracle_sgd(cuda):
    """6 slides, one SGD step each through the reference call sequence on the HIP kernels; the CPU oracle
    takes the same steps. Parameters after the epoch agree to 1e-4 (north-star tolerance)."""
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.train import train_loop, validate
    torch.manual_seed(8)
    model = TOAD_fc_mtl_concat(n_classes=18)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.relocate()
    slides = []
    for i, n in enumerate((120, 333, 64, 257, 90, 500)):
        g = torch.Generator().manual_seed(300 + i)
        slides.append((torch.randn(n, 1024, generator=g), torch.tensor([i % 18]), torch.tensor([i % 2]), torch.tensor([float(i % 2)])))
    lr = 0.05
    r = train_loop(0, model, slides, torch.optim.SGD(model.parameters(), lr=lr), 18)
    for data, label, site, sex in slides:
        out, loss, grads = orc.fwd_bwd(params, data, sex, label, site)
        for k in params:
            params[k] = params[k] - lr * grads[k]
    new = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for k in params:
        assert (new[k] - params[k]).abs().max().item() <= 1e-4, k
    assert r["slides"] == 6 and 0.0 <= r["cls_error"] <= 1.0 and np.isfinite(r["cls_loss"])
    v = validate(model, slides, 18)
    o = [orc.forward(params, d, s)[0] for d, _, _, s in slides]
    ref_prob = torch.cat([x["Y_prob"] for x in o]).numpy()
    assert np.abs(v["prob"] - ref_prob).max() <= 1e-4
    assert v["slides"] == 6


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_fused_train_loop_equals_the_reference_sequence(cuda, opt):
    """train_loop's fused route (one toad_mil_step_f32 per slide + the optimiser) against the reference's literal sequence
    (model(data, sex), two CE modules, backward, step, zero_grad; core_utils_mtl_concat.py:201-234): same epoch statistics and
    the same parameters after 7 slides - with the torch optimiser get_optim(flat=False) builds, and with the one-launch flat
    optimiser of get_optim (flat=True). Also checks get_optim's error branch (utils/utils.py:68-69)."""
    from types import SimpleNamespace
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.optim import get_optim, FlatAdam, FlatSGD
    from toad_amd.train import train_loop
    c = 18
    slides = []
    for i, n in enumerate((120, 333, 64, 1025, 90, 500, 256)):
        g = torch.Generator().manual_seed(700 + i)
        slides.append((torch.randn(n, 1024, generator=g), torch.tensor([(7 * i) % c]), torch.tensor([i % 2]), torch.tensor([float(i % 2)])))
    args = SimpleNamespace(opt=opt, lr=2e-3, reg=1e-5)

    def run(flat, fused):
        torch.manual_seed(21)
        model = TOAD_fc_mtl_concat(n_classes=c); model.relocate()
        o = get_optim(model, args, flat=flat)
        assert isinstance(o, (FlatAdam, FlatSGD)) == flat
        stats = [train_loop(e, model, slides, o, c, fused=fused) for e in range(2)]
        return stats, model.flat_parameters().detach().cpu().clone()

    ref_stats, ref_p = run(False, False)
    for flat in (False, True):
        stats, p = run(flat, True)
        diff = (p - ref_p).abs()
        if opt == "sgd":
            assert diff.max().item() <= 2e-2 * args.lr, (flat, diff.max().item())
        else:
            # Adam's update is lr * m / (sqrt(v) + eps): an element whose gradient is round-off (the cancellation-dominated
            # attention biases, never-hit class rows) steps by +-lr in a direction the last bit decides, so the bound on EVERY element
            # is the trivial one (steps x lr); the typical element must agree to a small fraction of one step
            assert diff.max().item() <= 2 * len(slides) * args.lr * 1.01, (flat, diff.max().item())
            assert diff.median().item() <= 1e-3 * args.lr and (diff > 0.1 * args.lr).double().mean().item() <= 0.02, \
                (flat, diff.median().item(), (diff > 0.1 * args.lr).double().mean().item())
        for e, (a, b) in enumerate(zip(stats, ref_stats)):
            assert a["slides"] == b["slides"] == len(slides)
            rel = 2e-5 if (opt == "sgd" or e == 0) else 1e-3      # (Adam: the round-off-driven elements above feed the second epoch)
            for k in ("cls_loss", "site_loss"):
                assert abs(a[k] - b[k]) <= rel * max(abs(b[k]), 1.0), (k, e)
            if opt == "sgd" or e == 0:
                assert a["cls_error"] == b["cls_error"] and a["site_error"] == b["site_error"]
                assert a["cls_acc"] == b["cls_acc"] and a["site_acc"] == b["site_acc"]
    with pytest.raises(NotImplementedError):
        get_optim(None, SimpleNamespace(opt="rmsprop", lr=1e-4, reg=0.0))
    with pytest.raises(ValueError):
        train_loop(0, TOAD_fc_mtl_concat(n_classes=c).cuda(), slides, None, c, loss_fn=torch.nn.CrossEntropyLoss(label_smoothing=0.1), fused=True)


def test_grouping_plan_and_get_optim_cpu():
    """Host logic that needs no GPU: the ragged grouping plan of forward_grouped (loader order preserved, groups cut at the row budget,
    oversized slides alone, models without forward_many called slide by slide), get_optim's torch branch and error branch
    (utils/utils.py:63-70), and train_loop's refusal to fuse for anything but the HIP module."""
    from types import SimpleNamespace
    from toad_amd.eval import forward_grouped
    from toad_amd.optim import get_optim
    from toad_amd.train import _fused_ok
    lens = [300, 5000, 40, 40, 1200, 7, 900, 2600, 0, 10]
    batches = [(torch.zeros(n, 4), torch.tensor([0]), torch.tensor([0]), torch.tensor([0.0])) for n in lens]
    calls = []

    class Spy:
        def __call__(self, data, sex):
            calls.append([int(data.shape[0])]); return {"n": int(data.shape[0])}

        def forward_many(self, bags, sexes):
            calls.append([int(b.shape[0]) for b in bags]); return [{"n": int(b.shape[0])} for b in bags]

    got = [(int(b[0].shape[0]), r["n"]) for b, r in forward_grouped(Spy(), batches, 4096)]
    assert got == [(n, n) for n in lens]
    assert calls == [[300], [5000], [40, 40, 1200, 7, 900], [2600], [0], [10]]
    calls.clear()
    assert [r["n"] for _, r in forward_grouped(Spy(), batches, 0)] == lens and calls == [[n] for n in lens]

    class Plain:                                            # a model without forward_many: one call per slide
        def __call__(self, data, sex):
            return {"n": int(data.shape[0])}

    assert [r["n"] for _, r in forward_grouped(Plain(), batches, 4096)] == lens
    lin = torch.nn.Linear(3, 2)
    o = get_optim(lin, SimpleNamespace(opt="adam", lr=1e-3, reg=1e-4), flat=False)
    assert isinstance(o, torch.optim.Adam) and o.defaults["lr"] == 1e-3 and o.defaults["weight_decay"] == 1e-4
    o = get_optim(lin, SimpleNamespace(opt="sgd", lr=1e-2, reg=0.0), flat=False)
    assert isinstance(o, torch.optim.SGD) and o.defaults["momentum"] == 0.9
    with pytest.raises(NotImplementedError):
        get_optim(lin, SimpleNamespace(opt="adagrad", lr=1e-2, reg=0.0))
    assert not _fused_ok(lin, o, None)
