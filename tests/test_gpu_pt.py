"""GPU: prepared bags (ABI 9) - the plane-tiled two-piece form of a slide's bag (csrc/gemm_pt.inc), the NT kernel that takes it by
LDS-DMA and the TN kernel that reads it with transposing LDS reads, against the fp32-bag path and against fp64."""
import math

import pytest
import torch

from tests.helpers import assert_grad_close, assert_grad_close_or_few_flips

pytestmark = pytest.mark.gpu

# Since round 4 a RAW fp32 bag is measured inside the first GEMM (gemm_h2.inc AMODE 3: each tile's scale comes from its first 32 columns with
# three bits of room) while a prepared bag carries planes scaled with each row tile's true maximum. The two scales differ by a power of two, so
# the (h, m) pieces have identical significands wherever they are normal fp16 numbers; only second pieces more than 2^14 below a tile's scale
# (elements ~1e-4 of the maximum) round differently. The two routes therefore agree to a few ulp, not bit for bit: 5e-6 of each tensor's
# scale here (1e-4 is the parity bound against the reference; run-to-run determinism of EACH route stays bitwise).
ROUTE_TOL = 5e-6


def assert_routes_agree(a, b, what="", trunk=False):
    """trunk: a gradient a ReLU mask reaches (dW1, db1, dW2, db2). The two routes' activations differ by a few ulp, so a pre-activation at round-off
    of zero may fall on either side of the ReLU: such a LEGITIMATE flip moves the gradient by one patch's rank-one contribution
    (helpers.assert_grad_close_or_few_flips); everything else must meet ROUTE_TOL."""
    if not a.is_floating_point():
        assert torch.equal(a, b), what
        return
    sc = max(a.abs().max().item(), 1e-30)
    if trunk:
        assert_grad_close_or_few_flips(a, b, ROUTE_TOL, sc, what=what)
        return
    assert (a - b).abs().max().item() <= ROUTE_TOL * sc, (what, (a - b).abs().max().item(), sc)


def decode_planes(pb):
    """PreparedBag -> the fp64 values its (h, m) planes represent, [N, K]: undoes the tiling
    pt[row tile][column stage][plane][row][phys group][8], phys = k-group ^ ((row >> 2) & 3), and the per-row-tile power-of-two scale."""
    n, k = pb.shape
    nrt, ncs = (n + 255) // 256, (k + 31) // 32
    t = pb.planes.cpu().view(torch.float16).view(nrt, ncs, 2, 256, 4, 8).double()
    v = t[:, :, 0] + t[:, :, 1]                                         # [rt, cs, row, phys, 8]
    rows = torch.arange(256)
    out = torch.empty(nrt, 256, ncs, 4, 8, dtype=torch.float64)
    for g in range(4):
        phys = g ^ ((rows >> 2) & 3)                                     # where logical group g of each row lives
        out[:, :, :, g, :] = v[:, :, rows, phys, :].permute(0, 2, 1, 3)
    amax = pb.amax.cpu().double()
    kexp = torch.tensor([14 - math.frexp(a)[1] if a > 0 else 14 for a in amax.tolist()], dtype=torch.float64)
    out = out.reshape(nrt, 256, ncs * 32) * torch.pow(2.0, -kexp).view(-1, 1, 1)
    return out.reshape(nrt * 256, ncs * 32)[:n, :k], out.reshape(nrt * 256, ncs * 32)


@pytest.mark.parametrize("n", [64, 255, 256, 257, 1000, 5000])
def test_prepare_bag_represents_the_bag(cuda, n):
    """(h + m) / 2^k reproduces every element to 2^-21 of its row tile's abs-max (11 + 11 significand bits and a sign), rows beyond
    N are zeros, and the abs-max array is exact."""
    from toad_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 1024, generator=g) * 2.5
    x[n // 2, 17] = -31.0
    pb = ops.prepare_bag(x.to(cuda))
    assert pb.shape == (n, 1024) and pb.planes.numel() == ((n + 255) // 256) * 256 * 1024 * 4
    assert torch.equal(pb.amax.cpu(), ops.absmax_rows256(x.to(cuda)).cpu())
    got, full = decode_planes(pb)
    bmax = pb.amax.cpu().double().repeat_interleave(256)[:n].view(-1, 1)
    assert ((got - x.double()).abs() <= bmax * 2.0 ** -21).all()
    assert full[n:].abs().max().item() == 0.0 if full.shape[0] > n else True


def _setup(cuda, n, seed=0, scale_rows=None):
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(seed)
    model = TOAD_fc_mtl_concat(n_classes=18)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    model.relocate()
    x = torch.randn(n, 1024, generator=torch.Generator().manual_seed(seed + 1))
    if scale_rows is not None:
        x = x * scale_rows(n).view(-1, 1)
    return model, x.to(cuda)


def _step(model, bag, dev):
    from toad_amd import ops
    w = {k: v.detach() for k, v in model._weights().items()}
    g = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    sex = torch.tensor([1.0], device=dev); label = torch.tensor([3], device=dev); site = torch.tensor([1], device=dev)
    loss, logits, slog = ops.mil_step(w, g, 0.0, bag, sex, label, site, want_logits=True)
    return loss, logits, slog, g


@pytest.mark.parametrize("n", [64, 300, 777, 2049, 9000, 70000])
def test_step_on_a_prepared_bag_equals_the_fp32_bag(cuda, n):
    """toad_mil_step_xp_f32 vs toad_mil_step_f32: the planes hold the pieces the fp32 kernels derive every step up to the power of two of the
    scale (see ROUTE_TOL above), the products and their order are the same: every output and gradient agrees to a few ulp of its scale."""
    from toad_amd import ops
    model, x = _setup(cuda, n, seed=n)
    l0, lg0, sl0, g0 = _step(model, x, cuda)
    pb = ops.prepare_bag(x)
    l1, lg1, sl1, g1 = _step(model, pb, cuda)
    assert_routes_agree(lg0, lg1, "logits"); assert_routes_agree(sl0, sl1, "site logits"); assert_routes_agree(l0, l1, "loss")
    for k in ops.STEP_SLOTS:
        if k == "bc":                                           # exactly zero by the softmax's shift invariance: round-off of dWc-sized terms
            assert (g0[k] - g1[k]).abs().max().item() <= ROUTE_TOL * g0["wc"].abs().max().item(), k
            continue
        if k == "bab":                                          # column sums of dP that cancel almost completely: round-off of the cancelling terms
            assert (g0[k] - g1[k]).abs().max().item() <= 5e-5 * g0[k].abs().max().item(), k
            continue
        assert_routes_agree(g0[k], g1[k], k, trunk=k in ("w1", "b1", "w2", "b2"))
    # run-to-run determinism of the prepared path
    l2, lg2, sl2, g2 = _step(model, pb, cuda)
    assert all(torch.equal(g1[k], g2[k]) for k in ops.STEP_SLOTS) and torch.equal(l1, l2)


def test_first_layer_weight_gradient_with_row_tiles_of_different_magnitude(cuda):
    """The TN kernel rescales its accumulators when the bag's exponent changes between row tiles: tiles 1e-4 .. 1e+3 apart must
    give the fp64 weight gradient to round-off of its scale (the big tiles dominate the sum; the small ones must not corrupt it), and
    a bag that is tiny EVERYWHERE keeps full relative accuracy."""
    from toad_amd import ops

    def scales(n):
        s = torch.ones(n)
        s[:256] = 1e-4; s[256:512] = 1e3; s[768:1024] = 3e-2; s[1500:] = 7.0
        return s
    for sc_fn in (scales, lambda n: torch.full((n,), 1e-12)):
        model, x = _setup(cuda, 2100, seed=5, scale_rows=sc_fn)
        _, _, _, g0 = _step(model, x, cuda)
        _, _, _, g1 = _step(model, ops.prepare_bag(x), cuda)
        sc = g0["w1"].abs().max().item()
        assert torch.isfinite(g1["w1"]).all()
        assert (g0["w1"] - g1["w1"]).abs().max().item() <= 4e-6 * sc
    # and directly against fp64: dW = dY^T X
    g = torch.Generator().manual_seed(11)
    n = 1300
    xs = torch.randn(n, 1024, generator=g) * scales(n).view(-1, 1)
    dy = torch.randn(n, 512, generator=g) * 1e-3
    ref = dy.double().t() @ xs.double()
    dw, db = ops.linear_wgrad(dy.to(cuda), ops.prepare_bag(xs.to(cuda)))
    assert (dw.cpu().double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    assert (db.cpu().double() - dy.double().sum(0)).abs().max().item() <= 1e-5 * dy.double().abs().sum(0).max().item()


@pytest.mark.parametrize("n", [500, 9000])
def test_module_forward_backward_on_a_prepared_bag(cuda, n):
    """model(PreparedBag, sex) + loss.backward() (the drop-in path, toad_mil_fwd_xp_f32 / toad_mil_bwd_xp_f32) equals the fp32 bag."""
    from toad_amd import ops
    model, x = _setup(cuda, n, seed=3 + n)
    sex = torch.tensor([0.0], device=cuda); label = torch.tensor([5], device=cuda); site = torch.tensor([0], device=cuda)
    ce = torch.nn.CrossEntropyLoss()
    outs = []
    for bag in (x, ops.prepare_bag(x)):
        model.zero_grad(set_to_none=True)
        r = model(bag, sex, return_features=True)
        (0.75 * ce(r["logits"], label) + 0.25 * ce(r["site_logits"], site)).backward()
        outs.append((r, {k: p.grad.clone() for k, p in model.named_parameters()}))
    (r0, g0), (r1, g1) = outs
    for k in ("logits", "Y_prob", "site_logits", "site_prob", "A", "features", "Y_hat", "site_hat"):
        assert_routes_agree(r0[k], r1[k], k)
    for k in g0:
        if k.endswith("attention_c.bias"):
            assert (g0[k] - g1[k]).abs().max().item() <= ROUTE_TOL * g0[k.replace("bias", "weight")].abs().max().item(), k
            continue
        if k.endswith("attention_a.0.bias") or k.endswith("attention_b.0.bias"):     # cancellation-dominated column sums (tests/test_gpu_model.py)
            assert (g0[k] - g1[k]).abs().max().item() <= 5e-5 * g0[k].abs().max().item(), k
            continue
        assert_routes_agree(g0[k], g1[k], k, trunk=k.startswith("attention_net.0.") or k.startswith("attention_net.2."))
    with torch.no_grad():
        assert_routes_agree(model(x, sex, attention_only=True), model(ops.prepare_bag(x), sex, attention_only=True), "attention_only")
