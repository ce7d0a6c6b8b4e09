"""CPU: the oracle (oracle/toad_oracle.py) against the committed golden vectors, which are the
REFERENCE's outputs captured by oracle/pin_against_reference.py in the build container."""
import numpy as np
import pytest
import torch

from oracle import toad_oracle as orc
from tests.helpers import (case_inputs, check_activations_vs_golden, check_outputs_vs_golden, check_trunk_grads_vs_golden_blocks,
                           relu_flip_positions)

SMALL = ["n1", "n2", "n63", "n64", "n65", "n256", "n777", "n777_c2", "n1024_sat", "n300_equal", "n10000", "r256", "r10000"]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_reference_golden(golden, name):
    ci = case_inputs(golden, name)
    out, loss, grads = orc.fwd_bwd(ci["params"], ci["x"], ci["sex"], ci["label"], ci["site"])
    feat, _ = orc.forward(ci["params"], ci["x"], ci["sex"], return_features=True)
    out = {k: v.detach() for k, v in out.items()}
    out["features"] = feat["features"]
    check_outputs_vs_golden(golden, name, out, loss, grads, atol=2e-5)


@pytest.mark.parametrize("name", ["n256", "n777_c2", "n1024_sat", "n10000", "r256", "r10000"])
def test_trunk_gradient_blocks_keep_the_reference_as_comparand(golden, name):
    """The flip-tolerant comparison the GPU suite uses for dW1 / db1 / dW2 / db2 (tests/helpers.py), exercised here with the fp32 oracle
    standing in for the device: the oracle in fp32 is bitwise the reference in fp32, whose ReLU masks differ from the fp64 reference's
    at a few pre-activations within round-off of zero (n1024_sat: 2 in layer 2, n10000: 1 + 5). Everything those flips cannot explain
    must match the reference's fp64 gradient matrices; a gradient that is actually wrong must be caught."""
    ci = case_inputs(golden, name)
    _, saved = orc.forward(ci["params"], ci["x"], ci["sex"])
    _, _, grads = orc.fwd_bwd(ci["params"], ci["x"], ci["sex"], ci["label"], ci["site"])
    p1, p2 = relu_flip_positions(ci["params"], ci["x"], saved.h1, saved.h)
    assert check_trunk_grads_vs_golden_blocks(golden, name, grads, ci["x"], p1, p2) >= 50       # of the 2 x 32 strided block rows
    assert check_activations_vs_golden(golden, name, saved.h1, saved.h) == (name in ("n10000", "r10000"))
    # sensitivity: one wrong element in an unflipped row, or a rank-one error along a patch that did NOT flip, fails the check
    k1, k2 = "attention_net.0.weight", "attention_net.2.weight"
    flipped2 = set(int(j) for j in p2[:, 1].tolist())
    row = next(r for r in golden[name + "/grad_block_rows"].tolist() if r not in flipped2)
    for key, delta in ((k2, None), (k1, "rank1")):
        bad = {k: v.clone() for k, v in grads.items()}
        sc = float(golden[name + "/grad_absmax/" + key])
        if delta is None:
            bad[key][row, 7] += 1e-3 * sc
        else:
            other = next(n for n in range(ci["n"]) if n not in set(int(v) for v in p2[:, 0].tolist()))
            bad[key][:128] += 1e-3 * sc * torch.outer(torch.ones(128), ci["x"][other] / ci["x"][other].abs().max())
        with pytest.raises(AssertionError):
            check_trunk_grads_vs_golden_blocks(golden, name, bad, ci["x"], p1, p2)


def test_oracle_attention_only(golden):
    ci = case_inputs(golden, "n777")
    a = orc.forward(ci["params"], ci["x"], ci["sex"], attention_only=True)
    assert a.shape == (777,)
    from tests.helpers import strided_sample
    assert np.abs(strided_sample(a) - golden["n777/A_only_sample"]).max() <= 2e-5


def test_oracle_backward_matches_autograd():
    """The hand-written backward (the kernels' CPU twin) equals torch autograd on the same graph."""
    torch.manual_seed(3)
    params = {k: v.clone().requires_grad_(True) for k, v in orc.xavier_params(18, seed=5).items()}
    for k in params:
        if params[k].dim() == 1:
            params[k].data.normal_(0, 0.05)
    x = torch.randn(333, 1024)
    sex = torch.tensor([1.0]); label = torch.tensor([7]); site = torch.tensor([1])
    out, _ = orc.forward(params, x, sex)
    loss = orc.loss_fn(out["logits"], label, out["site_logits"], site) + 0.3 * out["A"].sin().sum()
    loss.backward()
    with torch.no_grad():
        p2 = {k: v.detach() for k, v in params.items()}
        out2, saved = orc.forward(p2, x, sex)
        dl, ds = orc.loss_grad(out2["logits"], label, out2["site_logits"], site)
        da = 0.3 * out2["A"].cos().t().contiguous()          # external gradient on A_raw [N,T]
        g = orc.backward(p2, saved, dl, ds, da_ext=da)
    for k in orc.PARAM_KEYS:
        ref = params[k].grad
        err = (g[k] - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1.0), (k, err)


def test_closed_form_inputs_are_deterministic():
    a = orc.closed_form_bag(17).numpy()
    b = orc.closed_form_bag(17).numpy()
    assert np.array_equal(a, b) and a.shape == (17, 1024) and np.isfinite(a).all()
    p = orc.closed_form_params(18)
    assert set(p) == set(orc.PARAM_KEYS)
    assert sum(v.numel() for v in p.values()) == 1192490


def test_oracle_dropout_backward_matches_autograd_and_reference_dropout_semantics():
    """With explicit Dropout(0.25) multipliers the oracle's manual backward equals autograd, and the
    multipliers are exactly what nn.Dropout applies (0 or 1/0.75), models/model_toad.py:27-29,61,64."""
    torch.manual_seed(7)
    params = {k: v.clone().requires_grad_(True) for k, v in orc.xavier_params(18, seed=3).items()}
    n = 211
    x = torch.randn(n, 1024)
    sex = torch.tensor([0.0]); label = torch.tensor([2]); site = torch.tensor([0])
    mk = {k: (torch.rand(n, w) >= 0.25).float() / 0.75 for k, w in (("h1", 512), ("h", 512), ("a", 384), ("b", 384))}
    d = torch.nn.Dropout(0.25); d.train()
    y = d(torch.ones(1000, 100))
    assert set(y.unique().tolist()) == {0.0, 1.0 / 0.75} or set(y.unique().tolist()) == {0.0, float(torch.tensor(1.0) / 0.75)}
    out, _ = orc.forward(params, x, sex, masks=mk)
    orc.loss_fn(out["logits"], label, out["site_logits"], site).backward()
    with torch.no_grad():
        p2 = {k: v.detach() for k, v in params.items()}
        _, _, g = orc.fwd_bwd(p2, x, sex, label, site, masks=mk)
    for k in orc.PARAM_KEYS:
        ref = params[k].grad
        assert (g[k] - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0), k


@pytest.mark.parametrize("drop", [False, True])
def test_batch_checker_accepts_a_correct_batch_and_rejects_a_wrong_gradient(drop):
    """Self-test of tests/helpers.py::check_batch_against_oracle (the checker of the config-4-shape GPU tests) without a GPU: fed the oracle's
    own fp32 forward activations and the autograd gradient of the summed per-slide losses (utils/core_utils_mtl_concat.py:200-234 semantics) it
    passes; with one gradient off by 1e-3 of its scale, or one slide's rows of H shifted, it fails."""
    from tests.helpers import SLOT2KEY, check_batch_against_oracle
    torch.manual_seed(11)
    lens = [300, 77, 513]
    B = len(lens)
    base = orc.xavier_params(18, seed=9)
    for k in base:
        if base[k].dim() == 1:
            base[k].normal_(0, 0.05)
    params = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    slides, masks, offs = [], ([] if drop else None), [0]
    for i, n in enumerate(lens):
        slides.append((torch.randn(n, 1024), torch.tensor([float(i % 2)]), torch.tensor([(5 * i) % 18]), torch.tensor([i % 2])))
        offs.append(offs[-1] + n)
        if drop:
            masks.append({k: (torch.rand(n, w) >= 0.25).float() / 0.75 for k, w in (("h1", 512), ("h", 512), ("a", 384), ("b", 384))})
    dev = {k: [] for k in ("h1", "h", "p", "a_raw", "logits", "site_logits", "loss")}
    total = 0.0
    for b, (x, sx, lb, st) in enumerate(slides):
        out, sv = orc.forward(params, x, sx, masks=None if masks is None else masks[b])
        loss = orc.loss_fn(out["logits"], lb, out["site_logits"], st)
        total = total + loss / B
        for k, v in (("h1", sv.h1), ("h", sv.h), ("p", sv.p), ("a_raw", sv.a_raw), ("logits", out["logits"]), ("site_logits", out["site_logits"])):
            dev[k].append(v.detach())
        dev["loss"].append(torch.stack([loss.detach() / B] * 3)[None])
    total.backward()
    dev = {k: torch.cat(v, 0) for k, v in dev.items()}
    grads = {s: params[k].grad.clone() for s, k in SLOT2KEY.items()}
    _, _, flips = check_batch_against_oracle("selftest", base, slides, offs, dev, grads, masks)
    assert flips[0] + flips[1] <= 2
    bad = dict(grads); bad["w2"] = grads["w2"] + 1e-3 * grads["w2"].abs().max()
    with pytest.raises(AssertionError):
        check_batch_against_oracle("selftest", base, slides, offs, dev, bad, masks)
    bad_dev = dict(dev); bad_dev["h"] = dev["h"].clone(); bad_dev["h"][offs[1]:offs[2]] += 1e-2
    with pytest.raises(AssertionError):
        check_batch_against_oracle("selftest", base, slides, offs, bad_dev, grads, masks)
