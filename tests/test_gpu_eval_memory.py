"""GPU: forwards that no backward will follow (eval loops: utils/core_utils_mtl_concat.py:284,393, utils/eval_utils_mtl_concat.py:91) must
not keep a slide's saved activations alive through the small outputs the caller collects."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_no_grad_outputs_do_not_pin_the_activation_arena(cuda):
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(0)
    model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.eval()
    n = 20000
    x = torch.randn(n, 1024, device=cuda); sex = torch.tensor([1.0], device=cuda)
    ref = model(x, sex, return_features=True)                       # grad mode: autograd bridge, outputs are arena views
    assert ref["logits"].requires_grad
    with torch.no_grad():
        res = model(x, sex, return_features=True)
        att = model(x, sex, attention_only=True)
    for k in ("logits", "Y_prob", "Y_hat", "site_logits", "site_prob", "site_hat", "A", "features"):
        assert torch.equal(res[k], ref[k].detach()), k
        assert res[k].untyped_storage().nbytes() <= max(8 * n, 1 << 16), (k, res[k].untyped_storage().nbytes())   # own small storage
    assert torch.equal(att, ref["A"][0].detach()) and att.untyped_storage().nbytes() <= 8 * n
    # many slides collected the way validate()/summary() do: memory must not grow by an arena (~140 MB at 20k patches) per slide
    torch.cuda.synchronize(); base = torch.cuda.memory_allocated()
    keep = []
    with torch.no_grad():
        for _ in range(12):
            r = model(x, sex)
            keep.append((r["Y_prob"], r["Y_hat"], r["site_prob"]))
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() - base < 32 << 20, (torch.cuda.memory_allocated() - base) >> 20
    # a frozen model in grad mode takes the same route
    for p in model.parameters():
        p.requires_grad_(False)
    r2 = model(x, sex)
    assert not r2["logits"].requires_grad and torch.equal(r2["logits"], ref["logits"].detach())
