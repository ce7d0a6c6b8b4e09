"""Shared helpers for the test-suite (oracle side only; never imported by the product)."""
import numpy as np
import torch

from oracle import toad_oracle as orc

SLOT2KEY = dict(zip(
    ("w1", "b1", "w2", "b2", "wa", "ba", "wb", "bb", "wc", "bc", "wcls", "bcls", "wsite", "bsite"),
    orc.PARAM_KEYS))


def case_inputs(golden, name):
    n, c, sex, label, site, scale, equal = golden[name + "/meta"]
    n, c = int(n), int(c)
    params = orc.closed_form_params(c)
    x = orc.closed_form_bag(n, 1024, float(scale), "equal" if equal > 0.5 else "wave")
    return dict(n=n, c=c, params=params, x=x, sex=torch.tensor([float(sex)]),
                label=torch.tensor([int(label)]), site=torch.tensor([int(site)]))


def strided_sample(t, k=64):
    flat = t.detach().reshape(-1).cpu()
    n = flat.numel()
    if n == 0:
        return np.zeros((0,), dtype=np.float32)
    idx = (np.arange(k, dtype=np.int64) * max(n // k, 1)) % n
    return flat[torch.from_numpy(idx)].numpy().astype(np.float32)


def assert_close(a, b, atol, rtol=0.0, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), f"{what}: max err {err.max():.3e} (tol {tol.flat[err.argmax()]:.3e}) at {np.argwhere(bad)[:3].tolist()}"


def check_outputs_vs_golden(golden, name, out, loss, grads, atol=1e-4):
    """out: dict of CPU tensors (reference keys + 'features'); grads: {state-dict key: tensor}."""
    pre = name + "/"
    for k in ("logits", "Y_prob", "site_logits", "site_prob", "features"):
        assert_close(out[k].numpy(), golden[pre + k], atol, what=f"{name}:{k}")
    assert int(out["Y_hat"]) == int(golden[pre + "Y_hat"].item()), name
    assert int(out["site_hat"]) == int(golden[pre + "site_hat"].item()), name
    a = out["A"]
    assert tuple(a.shape) == (2, int(golden[pre + "meta"][0]))
    if pre + "A" in golden.files:
        assert_close(a.numpy(), golden[pre + "A"], atol, what=f"{name}:A")
    assert_close(strided_sample(a), golden[pre + "A_sample"], atol, what=f"{name}:A_sample")
    n = a.shape[1]
    assert_close(a.double().sum(1).numpy(), golden[pre + "A_sum"], atol * n, what=f"{name}:A_sum")
    if loss is not None:
        assert abs(float(loss) - float(golden[pre + "loss"])) <= atol, name
    if grads is not None:
        # Gradients are pinned to the REFERENCE run in fp64 (grad_sample64) with the reference's own
        # fp32-vs-fp64 deviation (grad_dev64, full-tensor max) as the yardstick: ReLU masks flip when a
        # pre-activation lies within fp32 roundoff of zero, so two correct fp32 implementations differ
        # by whole dZ rows (DESIGN.md "Parity").  Bound: the north star's 1e-4 absolute, or 4x the
        # reference's own fp32 noise where that is larger (only the x30-scaled adversarial bag).
        for k in orc.PARAM_KEYS:
            g = grads[k].detach().cpu()
            dev = float(golden[pre + "grad_dev64/" + k])
            tol = max(atol, 4.0 * dev)
            assert_close(strided_sample(g), golden[pre + "grad_sample64/" + k], tol, what=f"{name}:grad64:{k}")
            assert_close(strided_sample(g), golden[pre + "grad_sample/" + k], tol + dev, what=f"{name}:grad32:{k}")
            l2 = float(golden[pre + "grad_l2_64/" + k])
            assert abs(float(g.double().norm()) - l2) <= 5e-3 * l2 + 1e-6, (name, k, float(g.double().norm()), l2)
