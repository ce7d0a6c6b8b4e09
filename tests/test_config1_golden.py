"""BASELINE config 1 as stated (18 classes, 256-patch x 1024-d bags, the harness's default Adam: lr 1e-4, weight decay 1e-5, seed 1) pinned
STEP BY STEP to the reference's own train_loop (oracle/pin_config1_against_reference.py -> tests/golden/toad_config1_golden.npz).
CPU: the oracle + torch.optim.Adam replay the fixture. GPU: toad_amd.train.train_loop on the HIP module replays it - fused step + FlatAdam
(what get_optim returns) and the literal sequence with torch.optim.Adam.
Adam's update is sign-like, so two correct fp32 runs differ by up to 2 lr per step on the few elements whose gradient is below its own
round-off (the reference in fp32 vs itself in fp64 does, on 7e-4 of the parameters after 12 steps): the parameter check is therefore (a) the
sign-flip envelope 2 lr (j + 1) for every sampled element, (b) agreement to 2e-6 on all but a handful of the 896 sampled elements, and the
per-step losses - which see every parameter - to 2e-5 (CPU) / 1e-4 (GPU)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import toad_oracle as orc
from tests.helpers import strided_sample

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c1():
    g = np.load(os.path.join(REPO, "tests", "golden", "toad_config1_golden.npz"), allow_pickle=False)
    c, patches, steps, lr, reg, seed, bag_seed0 = g["meta"]
    return g, int(c), int(patches), int(steps), float(lr), float(reg), int(seed), int(bag_seed0)


def make_slide(i, c, patches, bag_seed0):
    """Same pure function of i as oracle/pin_config1_against_reference.py:slide (the fixture holds no inputs)."""
    gen = torch.Generator().manual_seed(bag_seed0 + i)
    return torch.randn(patches, 1024, generator=gen), torch.tensor([i % c]), torch.tensor([i % 2]), torch.tensor([(i // 2) % 2])


def check_step(g, j, lr, params, cls_loss, site_loss, y_hat, s_hat, loss_tol, max_far):
    assert abs(cls_loss - float(g["cls_loss"][j])) <= loss_tol and abs(site_loss - float(g["site_loss"][j])) <= loss_tol, (j, cls_loss, site_loss)
    assert int(y_hat) == int(g["Y_hat"][j]) and int(s_hat) == int(g["site_hat"][j]), j
    far = 0
    for k in orc.PARAM_KEYS:
        d = np.abs(strided_sample(params[k]) - g["param_sample/" + k][j])
        assert d.max() <= 2.0 * lr * (j + 1) + 1e-7, (j, k, float(d.max()))            # inside the sign-flip envelope
        far += int((d[:min(d.size, params[k].numel())] > float(g["tight"])).sum())     # (a strided sample of a short vector repeats its elements)
        l2 = float(g["param_l2/" + k][j])
        # (absolute slack: a few sign-flip elements - attention_c.bias, whose exact gradient is zero, consists of nothing else)
        assert abs(float(params[k].double().norm()) - l2) <= 2e-5 * l2 + 4.0 * lr * (j + 1), (j, k)
    assert far <= max_far, (j, far)


def test_initialisation_is_the_references_under_seed_1(c1):
    """The drop-in's constructor draws the same parameters as the reference's under torch.manual_seed(1) (same layer construction order,
    then xavier_normal_ in modules() order; models/model_toad.py:54-75, utils/utils.py:150-154): step 0 of the fixture depends on it."""
    g, c, patches, steps, lr, reg, seed, b0 = c1
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(seed)
    sd = TOAD_fc_mtl_concat(dropout=False, n_classes=c).state_dict()
    ref = orc.xavier_params_like_reference(c, seed)
    assert all(torch.equal(sd[k], ref[k]) for k in orc.PARAM_KEYS)


def test_oracle_replays_the_references_adam_steps(c1):
    g, c, patches, steps, lr, reg, seed, b0 = c1
    p0 = orc.xavier_params_like_reference(c, seed)
    plist = [torch.nn.Parameter(p0[k].clone()) for k in orc.PARAM_KEYS]
    opt = torch.optim.Adam(plist, lr=lr, weight_decay=reg)                       # get_optim's adam branch (utils/utils.py:64-65)
    for j in range(steps):
        x, label, site, sex = make_slide(j, c, patches, b0)
        cur = {k: q.detach() for k, q in zip(orc.PARAM_KEYS, plist)}
        out, _, grads = orc.fwd_bwd(cur, x, sex.float(), label, site)
        for k, q in zip(orc.PARAM_KEYS, plist):
            q.grad = grads[k].clone()
        opt.step()
        check_step(g, j, lr, {k: q.detach() for k, q in zip(orc.PARAM_KEYS, plist)},
                   torch.nn.functional.cross_entropy(out["logits"], label).item(), torch.nn.functional.cross_entropy(out["site_logits"], site).item(),
                   out["Y_hat"], out["site_hat"], loss_tol=2e-5, max_far=3)


@pytest.mark.gpu
@pytest.mark.parametrize("flat", [True, False])
def test_hip_train_loop_replays_the_references_adam_steps(cuda, c1, flat):
    """toad_amd.train.train_loop, one slide per call so that every step's statistics are visible: flat=True is what the harness gets from
    toad_amd.optim.get_optim (fused slide step + one-launch FlatAdam), flat=False the reference's literal sequence with torch.optim.Adam."""
    g, c, patches, steps, lr, reg, seed, b0 = c1
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.optim import get_optim
    from toad_amd.train import train_loop
    torch.manual_seed(seed)
    model = TOAD_fc_mtl_concat(dropout=False, n_classes=c)
    model.relocate()
    opt = get_optim(model, types.SimpleNamespace(opt="adam", lr=lr, reg=reg), flat=flat)
    for j in range(steps):
        batch = make_slide(j, c, patches, b0)
        st = train_loop(j, model, [batch], opt, c, fused=None if flat else False)
        y_hat = [i for i, (_, ok, n) in enumerate(st["cls_acc"]) if n]             # the one slide of this call was counted under its label
        assert y_hat == [int(batch[1])]
        hit, s_hit = st["cls_error"] == 0.0, st["site_error"] == 0.0
        # Y_hat itself: right if the slide was a hit, else anything but the label - the fixture's Y_hat must agree with that
        assert hit == (int(g["Y_hat"][j]) == int(batch[1])) and s_hit == (int(g["site_hat"][j]) == int(batch[2])), j
        params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        check_step(g, j, lr, params, st["cls_loss"], st["site_loss"], g["Y_hat"][j], g["site_hat"][j], loss_tol=1e-4, max_far=12)
