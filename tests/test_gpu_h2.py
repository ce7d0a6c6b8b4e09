"""GPU: the fp16 two-piece GEMM path (csrc/gemm_h2.inc) - abs-max plumbing, power-of-two scales, the recomputed pooling addend -
and the whole-slide entry points (toad_mil_fwd_f32 / toad_mil_bwd_f32) against the per-op path and the oracle."""
import math

import pytest
import torch

from oracle import toad_oracle as orc
from tests.helpers import assert_grad_close, assert_step_grad_matches_per_op, grad_scale

pytestmark = pytest.mark.gpu


def _blockmax(t, rows=256):
    t = t.abs()
    return torch.stack([t[i:i + rows].max() for i in range(0, t.shape[0], rows)])


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-300)


@pytest.mark.parametrize("m,k", [(1, 512), (255, 1024), (256, 512), (257, 768), (5000, 1024), (70001, 512)])
def test_absmax_rows256_is_exact(cuda, m, k):
    from toad_amd import ops
    x = torch.randn(m, k, generator=torch.Generator().manual_seed(m + k)) * 3.0
    x[m // 2, k // 3] = -77.5                                    # a negative extreme: |.| must be taken
    got = ops.absmax_rows256(x.to(cuda)).cpu()
    assert got.shape == (ops.amax_floats(m),) and torch.equal(got, _blockmax(x))


@pytest.mark.parametrize("m", [1, 300, 777, 4096, 70000])
def test_epilogue_amax_equals_the_output_maximum(cuda, m):
    """y_amax written by the GEMM epilogue (whole tiles) and by the fix-up kernel (K-split remainder tiles) is EXACTLY the
    per-256-row abs-max of the stored output, with bias / ReLU / mask applied, and feeding it to the next GEMM gives bitwise
    the result of letting that GEMM measure its operand itself."""
    from toad_amd import ops
    g = torch.Generator().manual_seed(m)
    x = torch.randn(m, 1024, generator=g); w = torch.randn(512, 1024, generator=g) * 0.05; b = torch.randn(512, generator=g)
    y, ya = ops.linear_act_fwd(x.to(cuda), w.to(cuda), b.to(cuda), 1, want_amax=True)
    assert torch.equal(ya.cpu(), _blockmax(y.cpu()))
    w2 = (torch.randn(768, 512, generator=g) * 0.05).to(cuda)
    p1 = ops.linear_act_fwd(y, w2, None, 0, x_amax=ya)
    p2 = ops.linear_act_fwd(y, w2, None, 0)
    assert torch.equal(p1, p2)
    dy = torch.randn(m, 768, generator=g).to(cuda)
    dx, da = ops.linear_dgrad(dy, ops.transpose(w2), relu_src=y, want_amax=True)
    assert torch.equal(da.cpu(), _blockmax(dx.cpu()))
    dw1, db1 = ops.linear_wgrad(dx, x.to(cuda), dy_amax=da, x_amax=ops.absmax_rows256(x.to(cuda)))
    dw2, db2 = ops.linear_wgrad(dx, x.to(cuda))
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1.0, 1e12, 1e30])
def test_scales_cover_the_fp32_range(cuda, scale):
    """fp16 has 5 exponent bits: the power-of-two operand scales must bring any fp32 magnitude into range (and back)."""
    from toad_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(600, 512, generator=g) * scale
    w = torch.randn(512, 512, generator=g) * 0.05
    y = ops.linear_act_fwd(x.to(cuda), w.to(cuda), None, 0).cpu()
    ref = (x.double() @ w.double().t()).float()
    assert torch.isfinite(y).all() and _rel(y, ref) <= 1e-5
    dy = torch.randn(600, 512, generator=g) * scale
    dw, db = ops.linear_wgrad(dy.to(cuda), x.to(cuda) / scale)
    assert _rel(dw.cpu(), (dy.double().t() @ (x.double() / scale)).float()) <= 2e-5


def test_row_blocks_with_very_different_magnitudes(cuda):
    """NT products scale every 256-row tile by its own abs-max: a block of tiny rows next to a block of huge rows keeps
    fp32-level RELATIVE accuracy in both (a single tensor-wide scale would flush the tiny block)."""
    from toad_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1024, 512, generator=g)
    x[:256] *= 1e-9; x[256:512] *= 1e7; x[768:] *= 1e-3
    w = torch.randn(768, 512, generator=g) * 0.05
    y = ops.linear_act_fwd(x.to(cuda), w.to(cuda), None, 0).cpu()
    ref = (x.double() @ w.double().t()).float()
    for i in range(0, 1024, 256):
        assert _rel(y[i:i + 256], ref[i:i + 256]) <= 1e-5, i
    # weights: every output column (row of W) has its own scale as well
    w2 = w.clone(); w2[::3] *= 1e-8; w2[1::3] *= 1e6
    y2 = ops.linear_act_fwd(x[512:768].to(cuda), w2.to(cuda), None, 0).cpu()
    ref2 = (x[512:768].double() @ w2.double().t()).float()
    for j in range(3):
        assert _rel(y2[:, j::3], ref2[:, j::3]) <= 1e-5, j


def test_zero_and_constant_operands(cuda):
    from toad_amd import ops
    x = torch.zeros(300, 512, device=cuda); w = torch.randn(512, 512, device=cuda)
    b = torch.randn(512, device=cuda)
    y, ya = ops.linear_act_fwd(x, w, b, 0, want_amax=True)
    assert torch.equal(y, b.expand(300, 512)) and torch.equal(ya.cpu(), _blockmax(y.cpu()))
    y0 = ops.linear_act_fwd(x, torch.zeros_like(w), None, 1)
    assert y0.abs().max().item() == 0.0
    ones = torch.ones(513, 512, device=cuda)
    assert torch.equal(ops.linear_act_fwd(ones, torch.ones(512, 512, device=cuda), None, 0), torch.full((513, 512), 512.0, device=cuda))


@pytest.mark.parametrize("n,t", [(1, 2), (300, 2), (777, 1), (5000, 2), (70000, 2)])
def test_dgrad_with_recomputed_pool_addend(cuda, n, t):
    """dX = (dY W + softmax(A_raw) dM) * (H > 0): the pooling gradient recomputed in the epilogue (and in the fix-up kernel)
    equals the one the pooling backward materialises, to fp32 round-off."""
    from toad_amd import ops
    g = torch.Generator().manual_seed(n + t)
    l, d2 = 512, 768
    dy = torch.randn(n, d2, generator=g) * 1e-3; w = torch.randn(d2, l, generator=g) * 0.05
    h = torch.randn(n, l, generator=g).relu()
    a_raw = torch.randn(n, t, generator=g) * 2.0; dm = torch.randn(t, l, generator=g) * 0.01
    mx = a_raw.max(0).values; ex = (a_raw - mx).exp(); ssum = ex.sum(0)
    stats = torch.stack([mx, ssum], 1).contiguous()
    p = (a_raw.double() - mx.double()).exp() / ssum.double()
    ref = ((dy.double() @ w.double() + p @ dm.double()) * (h > 0)).float()
    dx, da = ops.linear_dgrad(dy.to(cuda), ops.transpose(w.to(cuda)), relu_src=h.to(cuda),
                              pool=(a_raw.to(cuda), stats.to(cuda), dm.to(cuda)), want_amax=True)
    assert _rel(dx.cpu(), ref) <= 1e-5
    assert torch.equal(da.cpu(), _blockmax(dx.cpu()))
    # against the materialised form through the same kernel
    dh = (p @ dm.double()).float()
    dx2 = ops.linear_dgrad(dy.to(cuda), ops.transpose(w.to(cuda)), addend=dh.to(cuda), relu_src=h.to(cuda))
    assert _rel(dx, dx2) <= 2e-6


def _model_and_bag(cuda, n, c=18, seed=0, dropout=False):
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(seed)
    model = TOAD_fc_mtl_concat(dropout=dropout, n_classes=c)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.relocate()
    x = torch.randn(n, 1024, generator=torch.Generator().manual_seed(seed + 1))
    return model, params, x


@pytest.mark.parametrize("n,drop", [(1, 0.0), (300, 0.0), (777, 0.25), (9000, 0.0)])
def test_whole_slide_calls_are_bitwise_the_per_op_path(cuda, n, drop):
    """toad_mil_fwd_f32 / toad_mil_bwd_f32 (one C call each, what model(data, sex) / loss.backward() run) == the per-op
    sequence of functional.mil_forward / mil_backward: same kernels, same order -> bitwise-equal outputs and gradients.
    Short bags: the three trunk / attention weight gradients of the whole-slide call share ONE launch with its own row splits, so those six
    tensors agree with the per-op calls to summation round-off instead (helpers.assert_step_grad_matches_per_op)."""
    from toad_amd import functional as F_, ops
    model, _, x = _model_and_bag(cuda, n, seed=n)
    w = {k: v.detach() for k, v in model._weights().items()}
    xg = x.to(cuda); sex = torch.ones(1, device=cuda)
    seed = 4242
    outs, sv = F_.mil_forward(w, xg, sex, drop, seed)
    arena = ops.mil_fwd(w, xg, sex, drop, seed)
    c = 18
    assert torch.equal(arena.view("logits", (1, c)), outs["logits"]) and torch.equal(arena.view("site_logits", (1, 2)), outs["site_logits"])
    assert torch.equal(arena.view("a_raw", (n, 2)), outs["A_nt"]) and torch.equal(arena.view("mcat", (2, 513)), outs["features"])
    assert torch.equal(arena.view("y_prob", (1, c)), outs["Y_prob"]) and torch.equal(arena.view("y_hat", (1, 1), torch.int64), outs["Y_hat"])
    assert torch.equal(arena.view("h", (n, 512)), sv.h) and torch.equal(arena.view("p", (n, 768)), sv.p)
    assert torch.equal(arena.view("h_amax", (ops.amax_floats(n),)), sv.h_amax)
    g0 = torch.Generator(device=cuda).manual_seed(5)
    dl = torch.randn(1, c, device=cuda, generator=g0); dsite = torch.randn(1, 2, device=cuda, generator=g0)
    da = torch.randn(n, 2, device=cuda, generator=g0) * 0.01; dfe = torch.randn(2, 513, device=cuda, generator=g0) * 0.01
    g_ref, dx_ref, dsex_ref = F_.mil_backward(w, sv, dl, dsite, da, dfe, need_dx=True, need_dsex=True)
    grads = {k: torch.full_like(w[k], 3.0) for k in ops.STEP_SLOTS}
    dx, dsex = ops.mil_bwd(w, grads, 0.0, xg, arena, dl, dsite, da, dfe, drop, seed, need_dx=True, need_dsex=True)
    d = w["wa"].shape[0]
    ref = dict(g_ref); ref["wab"] = torch.cat([g_ref["wa"], g_ref["wb"]], 0); ref["bab"] = torch.cat([g_ref["ba"], g_ref["bb"]], 0)
    for k in ops.STEP_SLOTS:
        assert_step_grad_matches_per_op(grads[k], ref[k], k, n)
    assert torch.equal(dx, dx_ref) and torch.equal(dsex, dsex_ref)
    # attention_only stops after the scores
    a_only = ops.mil_fwd(w, xg, None, drop, seed, attention_only=True).view("a_raw", (n, 2))
    assert torch.equal(a_only, outs["A_nt"])


def test_sex_and_bag_gradients_match_autograd_on_the_oracle(cuda):
    """d loss / d sex flows through BOTH heads (models/model_toad.py:99 appends sex to both pooled rows) and through
    results['features']; d loss / d data through the whole trunk. Reference: torch autograd over the oracle in fp64."""
    model, params, x = _model_and_bag(cuda, 1500, seed=7)
    sex = torch.tensor([1.0]); label = torch.tensor([4]); site = torch.tensor([1])
    xg = x.to(cuda).requires_grad_(True); sg = sex.to(cuda).requires_grad_(True)
    res = model(xg, sg, return_features=True)
    ce = torch.nn.CrossEntropyLoss()
    loss = ce(res["logits"], label.to(cuda)) * 0.75 + ce(res["site_logits"], site.to(cuda)) * 0.25 + 0.1 * res["features"].sum() \
        + 0.01 * res["A"].pow(2).sum()
    loss.backward()
    p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    x64 = x.double().requires_grad_(True); s64 = sex.double().requires_grad_(True)
    o, _ = orc.forward(p64, x64, s64, return_features=True)
    l64 = ce(o["logits"], label) * 0.75 + ce(o["site_logits"], site) * 0.25 + 0.1 * o["features"].sum() + 0.01 * o["A"].pow(2).sum()
    l64.backward()
    assert abs(loss.item() - l64.item()) <= 1e-4 * max(abs(l64.item()), 1.0)
    assert_grad_close(sg.grad, s64.grad, 2e-5, float(s64.grad.abs().max()))
    assert_grad_close(xg.grad, x64.grad, 1e-4, float(x64.grad.abs().max()))     # ReLU flips allowed for: fp32 forward vs fp64
    g64 = {k: v.grad for k, v in p64.items()}
    for k, p in model.named_parameters():
        assert_grad_close(p.grad, g64[k], 2e-4, grad_scale(g64, k), what=k)


def test_backward_twice_with_retain_graph(cuda):
    model, _, x = _model_and_bag(cuda, 400, seed=3)
    res = model(x.to(cuda), torch.zeros(1, device=cuda))
    loss = res["logits"].sum() + res["site_logits"].sum()
    loss.backward(retain_graph=True)
    g1 = [p.grad.clone() for p in model.parameters()]
    model.zero_grad(set_to_none=True)
    loss.backward()
    for a, p in zip(g1, model.parameters()):
        assert torch.equal(a, p.grad)


def test_out_of_range_label_poisons_the_loss(cuda):
    """torch's CrossEntropyLoss raises for a label outside [0, C); a kernel cannot - it must not read out of bounds and must
    make the error impossible to miss: NaN loss and NaN gradients."""
    from toad_amd import ops
    lg = torch.randn(1, 18, device=cuda); sl = torch.randn(1, 2, device=cuda)
    for lab, st in ((18, 0), (-1, 1), (3, 2)):
        loss, dl, ds = ops.mtl_ce_fwd_bwd(lg, sl, torch.tensor([lab], device=cuda), torch.tensor([st], device=cuda))
        assert torch.isnan(loss[0]).item()
        assert torch.isnan(dl).all().item() if not 0 <= lab < 18 else torch.isnan(ds).all().item()
    loss, dl, ds = ops.mtl_ce_fwd_bwd(lg, sl, torch.tensor([17], device=cuda), torch.tensor([1], device=cuda))
    assert torch.isfinite(loss).all() and torch.isfinite(dl).all() and torch.isfinite(ds).all()


def test_flat_sgd_matches_torch_sgd(cuda):
    """toad_sgd_step_f32 == torch.optim.SGD(lr, momentum=0.9, weight_decay) (get_optim's SGD branch, utils/utils.py:66-67)."""
    from toad_amd.optim import FlatSGD
    torch.manual_seed(8)
    n = 1192768
    p0 = torch.randn(n); grads = [torch.randn(n) * 0.01 for _ in range(4)]
    ref_p = torch.nn.Parameter(p0.clone())
    ref = torch.optim.SGD([ref_p], lr=0.05, momentum=0.9, weight_decay=1e-5)
    mine_p = p0.clone().to(cuda)
    mine = FlatSGD(mine_p, lr=0.05, momentum=0.9, weight_decay=1e-5)
    for g in grads:
        ref_p.grad = g.clone(); ref.step()
        mine.step(g.to(cuda))
    assert (mine_p.cpu() - ref_p.detach()).abs().max().item() <= 2e-6
    plain = FlatSGD(p0.clone().to(cuda), lr=0.1, momentum=0.0, weight_decay=0.0)
    plain.step(grads[0].to(cuda))
    assert (plain.p.cpu() - (p0 - 0.1 * grads[0])).abs().max().item() <= 1e-6


def test_flat_adam_state_round_trip_and_stale_buffer_guard(cuda):
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.dp import SlideShardedDP
    from toad_amd.optim import FlatAdam
    torch.manual_seed(1)
    p = torch.randn(4096, device=cuda); g = torch.randn(4096, device=cuda)
    a = FlatAdam(p.clone()); a.step(g); a.step(g)
    b = FlatAdam(a.p.clone()); b.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.state_dict().items()})
    a.step(g); b.step(g)
    assert torch.equal(a.p, b.p) and b.t == 3
    model = TOAD_fc_mtl_concat(n_classes=4); model.relocate()
    dp = SlideShardedDP(model, "adam")
    model.flatten_parameters()                                  # re-homes every parameter in a NEW flat buffer
    with pytest.raises(RuntimeError, match="flat parameter buffer was replaced"):
        dp.accumulate([], 1)


@pytest.mark.parametrize("m", [1, 300, 4096, 70000])
def test_one_bit_relu_image_gives_the_fp32_mask_result(cuda, m):
    """The forward epilogue's one-bit image of its ReLU output, read back by the dgrad of the same layer (same M, same width),
    gives bitwise the result of masking with the fp32 activations - whole tiles read the bits, remainder tiles the floats."""
    from toad_amd import ops
    g = torch.Generator().manual_seed(m + 5)
    x = torch.randn(m, 1024, generator=g).to(cuda); w = (torch.randn(512, 1024, generator=g) * 0.05).to(cuda); b = torch.randn(512, generator=g).to(cuda)
    y, ya, bits = ops.linear_act_fwd(x, w, b, 1, drop_p=0.25, drop_seed=77, want_bits=True)      # ReLU + dropout zeros
    y_plain = ops.linear_act_fwd(x, w, b, 1, drop_p=0.25, drop_seed=77)
    assert torch.equal(y, y_plain) and bits.numel() == ops.relu_bits_bytes(m, 512)
    dy = torch.randn(m, 768, generator=g).to(cuda); w2t = ops.transpose((torch.randn(768, 512, generator=g) * 0.05).to(cuda))
    a = ops.linear_dgrad(dy, w2t, relu_src=y, mask_scale=4.0 / 3.0, relu_bits=bits)
    r = ops.linear_dgrad(dy, w2t, relu_src=y, mask_scale=4.0 / 3.0)
    assert torch.equal(a, r)
    assert (a == 0).float().mean().item() > 0.5                                                   # the mask really masks


# ---- ONE arithmetic: shapes outside the two-piece kernels' envelope are padded, not rerouted (round 6) ---------------------------------

@pytest.mark.parametrize("m,k,n", [(999, 1000, 512), (300, 36, 200), (64, 100, 30), (1, 8, 4), (63, 520, 766), (2049, 1024, 512)])
def test_every_linear_shape_runs_on_the_two_piece_kernels(cuda, m, k, n):
    """toad_linear_h2_ok is false for reductions that are not a multiple of 32 and output widths that are not a multiple of 4 (the reference's
    nn.Linear / Attn_Net_Gated take any sizes, models/model_toad.py:19): the per-op wrappers zero-pad such operands (exact: padded products are
    zero) instead of handing them to the library's exact-fp32 fallback kernels. Forward, dgrad and wgrad against fp64, and the library's count
    of fallback launches (toad_fallback_launches) must not move - also for weight gradients over fewer than 64 rows, which took the fp32 TN
    kernel until round 6."""
    from toad_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(m * 31 + k + n)
    x = torch.randn(m, k, generator=g); w = torch.randn(n, k, generator=g) * 0.1; b = torch.randn(n, generator=g)
    dy = torch.randn(m, n, generator=g)
    before = lib.toad_fallback_launches()
    y = ops.linear_act_fwd(x.to(cuda), w.to(cuda), b.to(cuda), 1)
    dx = ops.linear_dgrad(dy.to(cuda), ops.transpose(w.to(cuda)))
    dw, db = ops.linear_wgrad(dy.to(cuda), x.to(cuda))
    torch.cuda.synchronize()
    assert lib.toad_fallback_launches() == before, "a public entry point fell back to the exact-fp32 kernels"
    x64, w64, dy64 = x.double(), w.double(), dy.double()
    assert _rel(y.cpu(), torch.relu(x64 @ w64.t() + b.double()).float()) <= 1e-5
    assert _rel(dx.cpu(), (dy64 @ w64).float()) <= 1e-5
    assert _rel(dw.cpu(), (dy64.t() @ x64).float()) <= 2e-5 and _rel(db.cpu(), dy64.sum(0).float()) <= 2e-5
    assert y.shape == (m, n) and dx.shape == (m, k) and dw.shape == (n, k)


def test_tiny_bags_and_odd_attention_shapes_never_reach_the_fallback_kernels(cuda):
    """The whole drop-in path on bags of 1, 2 and 63 patches (their weight gradients reduce over fewer than 64 rows) and a standalone
    Attn_Net_Gated whose L / D are no multiples of 32 / 4: zero launches of the exact-fp32 fallback kernels."""
    from toad_amd import Attn_Net_Gated, TOAD_fc_mtl_concat, _lib
    lib = _lib.load()
    torch.manual_seed(2)
    model = TOAD_fc_mtl_concat(n_classes=18); model.relocate(); model.train()
    sex, label, site = torch.ones(1, device=cuda), torch.tensor([3], device=cuda), torch.tensor([1], device=cuda)
    ce = torch.nn.CrossEntropyLoss()
    before = lib.toad_fallback_launches()
    for n in (1, 2, 63):
        out = model(torch.randn(n, 1024, device=cuda), sex)
        (ce(out["logits"], label) * 0.75 + ce(out["site_logits"], site) * 0.25).backward()
    net = Attn_Net_Gated(L=100, D=30, n_tasks=3).to(cuda)
    a, _ = net(torch.randn(50, 100, device=cuda).requires_grad_(True))
    a.sum().backward()
    torch.cuda.synchronize()
    assert lib.toad_fallback_launches() == before


# ---- fp16 feature bags (toad_mil_*_x16_f32) ------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [1, 255, 777, 2049, 20000])
def test_fp16_bag_equals_the_upcast_bag(cuda, n):
    """A bag stored as fp16 goes through the two-term kernels (first Linear: A16, its weight gradient: B16); an fp16 element is
    exactly a first piece with m = 0, so every product and the accumulation order are those of the fp32 call on the up-cast bag:
    outputs and all 14 gradients agree to round-off of the operand SCALING only (power-of-two scales: expected bitwise; the bound
    below is 1e-6 of each tensor's magnitude to stay independent of that argument). Also: drop-in forward/backward, attention_only,
    the fused step, and the errors (gradient w.r.t. an fp16 bag)."""
    from toad_amd import TOAD_fc_mtl_concat, ops
    from toad_amd.dp import hip_slide_grad
    torch.manual_seed(5)
    c = 18
    model = TOAD_fc_mtl_concat(n_classes=c); model.relocate(); model.train()
    g = torch.Generator().manual_seed(40 + n)
    x16 = (torch.randn(n, 1024, generator=g) * 0.7).half().cuda()
    x32 = x16.float()
    sex = torch.tensor([1.0]).cuda(); label = torch.tensor([3]).cuda(); site = torch.tensor([1]).cuda()
    loss_fn = torch.nn.CrossEntropyLoss()

    def run(x):
        model.zero_grad(set_to_none=True)
        r = model(x, sex, return_features=True)
        loss = loss_fn(r["logits"], label) * 0.75 + loss_fn(r["site_logits"], site) * 0.25
        loss.backward()
        return r, {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    r32, g32 = run(x32)
    r16, g16 = run(x16)
    for k in ("logits", "site_logits", "Y_prob", "site_prob", "features", "A"):
        tol = 1e-6 * max(r32[k].abs().max().item(), 1e-6)
        assert (r16[k] - r32[k]).abs().max().item() <= tol, k
    assert torch.equal(r16["Y_hat"], r32["Y_hat"])
    for k in g32:
        tol = 1e-6 * max(g32[k].abs().max().item(), 1e-12)
        assert (g16[k] - g32[k]).abs().max().item() <= tol, (k, (g16[k] - g32[k]).abs().max().item(), g32[k].abs().max().item())
    with torch.no_grad():
        a16, a32 = model(x16, sex, attention_only=True), model(x32, sex, attention_only=True)
    assert (a16 - a32).abs().max().item() <= 1e-6 * max(a32.abs().max().item(), 1e-6)
    # fused step
    w = {k: v.detach() for k, v in model._weights().items()}
    gr16 = {k: torch.zeros_like(v) for k, v in w.items()}
    gr32 = {k: torch.zeros_like(v) for k, v in w.items()}
    l16 = hip_slide_grad(model, gr16, (x16, sex, label, site), beta=0.0)
    l32 = hip_slide_grad(model, gr32, (x32, sex, label, site), beta=0.0)
    assert (l16 - l32).abs().max().item() <= 1e-6 * max(l32.abs().max().item(), 1.0)
    for k in ops.STEP_SLOTS:
        assert (gr16[k] - gr32[k]).abs().max().item() <= 1e-6 * max(gr32[k].abs().max().item(), 1e-12), k
    # a bag that requires grad is up-cast (the fp16 kernels have no dX); bf16 is up-cast as well
    xr = x16.clone().requires_grad_(True)
    r = model(xr, sex)
    (r["logits"].sum()).backward()
    assert xr.grad is not None and xr.grad.dtype == torch.float16
    rb = model(x16.bfloat16(), sex)
    assert rb["logits"].dtype == torch.float32
    if ops.x16_ok(n):
        with pytest.raises(ValueError):
            arena = ops.mil_fwd(w, x16, sex)
            ops.mil_bwd(w, gr16, 0.0, x16, arena, torch.zeros(1, c).cuda(), torch.zeros(1, 2).cuda(), need_dx=True)
    assert ops.x16_ok(n) == (n >= 64)                 # below 64 patches the bag is up-cast (the small wgrad kernel is fp32-only)


# ---- dynamic range INSIDE a scale group (review item: the tests above only vary magnitudes BETWEEN 256-row blocks) -----------------
# The two-piece operand x*s = h + m carries 22 bits relative to x as long as m is a normal fp16 number, i.e. for |x| >= 2^-17 of the
# abs-max its scale s was derived from; below that m goes subnormal and the representation error becomes ABSOLUTE: <= 2^-39 of that
# abs-max (gemm_h2.inc header). Sums over such operands therefore obey
#     |err| <= c * eps32 * sum_k |a_k b_k|   +   2^-38 * ( amax_A * sum_k |b_k|  +  amax_B * sum_k |a_k| )
# (c covers the representation of both operands, the dropped m.m term and the fp32 ACCUMULATION of K terms, whose round-off walks like
# sqrt(K) * eps32 * |partial sum| in any fp32 GEMM - the reference's CPU sgemm included: c = max(6, 0.75 sqrt(K)))
# with amax_A / amax_B the abs-max of the scale group (NT: the A row's 256-row block and the weight ROW; TN: the whole tensor, or the
# row tile for a prepared operand). These tests assert exactly that bound against fp64, on data built to sit in the absolute regime.
_EPS = 2.0 ** -24


def _nt_bound(x, w, blk_amax_rows, w_row_amax, c=None):
    c = max(6.0, 0.75 * x.shape[1] ** 0.5) if c is None else c
    ax, aw = x.double().abs(), w.double().abs()
    return c * _EPS * (ax @ aw.t()) + 2.0 ** -38 * (blk_amax_rows.double().view(-1, 1) * aw.sum(1).view(1, -1) + ax.sum(1).view(-1, 1) * w_row_amax.double().view(1, -1))


@pytest.mark.parametrize("kind", ["decades_in_block", "heavy_tail", "one_giant_row"])
def test_nt_rows_far_below_their_blocks_abs_max(cuda, kind):
    from toad_amd import ops
    g = torch.Generator().manual_seed(21)
    m, k, n = 1024, 1024, 512
    x = torch.randn(m, k, generator=g)
    if kind == "decades_in_block":          # rows at 2^0, 2^-10, 2^-20, 2^-30 of the block maximum, interleaved inside every 256-row block
        x *= torch.pow(2.0, -10.0 * (torch.arange(m) % 4).float()).view(-1, 1)
    elif kind == "heavy_tail":              # log-normal magnitudes (sigma = 4: ~8 decades) per ELEMENT, as ReLU features with rare huge activations
        x = x.sign() * torch.exp(4.0 * torch.randn(m, k, generator=g)).clamp(max=1e12)
    else:                                   # one row 2^24 above everything else in its block
        x[300] *= 2.0 ** 24
    w = torch.randn(n, k, generator=g) * 0.05
    w[::7] *= 1e-5                          # weight rows carry their own scale: small rows must stay relatively accurate
    y = ops.linear_act_fwd(x.to(cuda), w.to(cuda), None, 0).cpu().double()
    ref = x.double() @ w.double().t()
    blk = torch.stack([x[i:i + 256].abs().max() for i in range(0, m, 256)]).repeat_interleave(256)
    bound = _nt_bound(x, w, blk, w.abs().max(1).values)
    assert torch.isfinite(y).all()
    viol = ((y - ref).abs() - bound).max().item()
    assert viol <= 0.0, (kind, viol)
    # the small rows are NOT flushed: wherever a row sits above 2^-17 of its block maximum its result is relatively accurate
    big_enough = x.abs().max(1).values >= blk * 2.0 ** -17
    rel = ((y - ref).abs().sum(1) / ref.abs().sum(1).clamp_min(1e-300))[big_enough]
    assert rel.max().item() <= 2e-5, (kind, rel.max().item())


@pytest.mark.parametrize("prepared", [False, True])
def test_tn_attention_weighted_rows_spanning_ten_decades(cuda, prepared):
    """Weight gradient dW = dY^T X with dY rows scaled by softmax weights from 1 down to 1e-10 (what a peaked attention does to dZ)
    and heavy-tailed features: the reduction runs over ALL rows with one scale per operand (per row tile for a prepared X), so small
    rows live in the absolute regime; the sum must still be within the bound above of the fp64 value."""
    from toad_amd import ops
    g = torch.Generator().manual_seed(22)
    m, i, j = 3000, 512, 1024
    p = torch.pow(10.0, -10.0 * torch.rand(m, generator=g))
    p[17] = 1.0
    dy = torch.randn(m, i, generator=g) * p.view(-1, 1)
    x = torch.randn(m, j, generator=g) * torch.exp(1.5 * torch.randn(m, 1, generator=g))     # per-row magnitudes over ~3 decades
    xd = x.to(cuda)
    dw, db = ops.linear_wgrad(dy.to(cuda), ops.prepare_bag(xd) if prepared else xd)
    ref = dy.double().t() @ x.double()
    ady, ax = dy.double().abs(), x.double().abs()
    # scale groups: dY one abs-max for the tensor; X one for the tensor, or one per 256-row tile when prepared (then sum tile-wise)
    if prepared:
        xa = torch.stack([x[r:r + 256].abs().max() for r in range(0, m, 256)]).repeat_interleave(256)[:m].double()
        floor = 2.0 ** -38 * (dy.abs().max().double() * ax.sum(0).view(1, -1) + (ady * xa.view(-1, 1)).sum(0).view(-1, 1))
    else:
        floor = 2.0 ** -38 * (dy.abs().max().double() * ax.sum(0).view(1, -1) + x.abs().max().double() * ady.sum(0).view(-1, 1))
    bound = max(6.0, 0.75 * m ** 0.5) * _EPS * (ady.t() @ ax) + floor
    viol = ((dw.cpu().double() - ref).abs() - bound).max().item()
    assert viol <= 0.0, viol
    assert ((dw.cpu().double() - ref).abs().max() / ref.abs().max()).item() <= 1e-6        # and it is fp32-accurate at the tensor's scale
    assert (db.cpu().double() - dy.double().sum(0)).abs().max().item() <= 8 * _EPS * ady.sum(0).max().item()


@pytest.mark.parametrize("m", [300, 9000, 70000])
@pytest.mark.parametrize("kind", ["normal", "late_outliers", "decades"])
def test_first_gemm_measures_a_raw_bag_itself(cuda, m, kind):
    """x_amax = None: the GEMM measures its fp32 A operand while it stages it (gemm_h2.inc AMODE 3) - an item's scale from its first 32 columns
    with three bits of room; an item whose later columns outgrow that room is POISONED and repeated in a second pass with its true exponent.
    Checked against fp64 with the bound of the two-piece arithmetic, against the route that is handed the abs-max array (a few ulp), and the
    abs-max array the whole-slide forward leaves in its arena against absmax_rows256 (bitwise: the weight gradient scales the bag with it).
    Shapes: 300 rows = K-split slices only; 9,000 = slices of remainder tiles; 70,000 = whole tiles + slices.
    late_outliers: leading columns ~1e-3, values up to 4e3 further right in a third of the row tiles (every such item must be repeated);
    decades: log-normal features over eight decades."""
    from toad_amd import ops
    g = torch.Generator().manual_seed(m + len(kind))
    k, n = 1024, 512
    x = torch.randn(m, k, generator=g)
    if kind == "late_outliers":
        x[:, :64] *= 1e-3
        blocks = torch.arange((m + 255) // 256)
        for b in blocks[blocks % 3 == 1].tolist():
            r = torch.randint(b * 256, min(m, b * 256 + 256), (5,), generator=g)
            c = torch.randint(100, k, (5,), generator=g)
            x[r, c] = torch.tensor([4e3, -2.5e3, 900.0, 1.7e3, -3e3])
    elif kind == "decades":
        x = x.sign() * torch.exp(torch.randn(m, k, generator=g) * 4.0)
    w = torch.randn(n, k, generator=g) * 0.03
    b = torch.randn(n, generator=g) * 0.05
    xg, wg, bg = x.to(cuda), w.to(cuda), b.to(cuda)
    y_run = ops.linear_act_fwd(xg, wg, bg, ops.ACT_NONE)                               # measured inside
    y_arr = ops.linear_act_fwd(xg, wg, bg, ops.ACT_NONE, x_amax=ops.absmax_rows256(xg))   # handed the array
    ref = torch.addmm(b.double(), x.double(), w.double().t())
    assert torch.isfinite(y_run).all()
    # two-piece bound (test_gpu_h2.py above): c eps sum|ab| + 2^-35 (amax_A sum|b| + ...): rows are independent, take the row-wise form
    sab = x.double().abs() @ w.double().abs().t()
    blk_max = torch.stack([x[i:i + 256].abs().max() for i in range(0, m, 256)]).double().repeat_interleave(256)[:m, None]
    # (_nt_bound's form; the absolute term of A is 2^-35 here: three bits of room above an item's first stage. c = 36 instead of 24: the
    #  statistical accumulation term is asked to hold for up to 36 M outputs, 70 x the sample of the tests above)
    w_row_amax = w.double().abs().max(1).values[None, :]
    bound = 1.5 * max(6.0, 0.75 * math.sqrt(k)) * 2.0 ** -24 * sab + 2.0 ** -35 * blk_max * w.double().abs().sum(1)[None, :] \
        + 2.0 ** -38 * x.double().abs().sum(1)[:, None] * w_row_amax + 1e-30
    assert ((y_run.cpu().double() - ref).abs() <= bound).all(), ((y_run.cpu().double() - ref).abs() / bound).max().item()
    sc = ref.abs().max().item()
    assert (y_run - y_arr).abs().max().item() <= 5e-6 * sc
    # run-to-run determinism (the redo pass and the slab alignment are fixed-order)
    assert torch.equal(y_run, ops.linear_act_fwd(xg, wg, bg, ops.ACT_NONE))


@pytest.mark.parametrize("n", [777, 9000, 70000])
def test_whole_slide_forward_leaves_the_bags_abs_max_array(cuda, n):
    """toad_mil_fwd_f32 on a raw fp32 bag runs no abs-max pass: the first GEMM's by-product must be exactly the array absmax_rows256 gives."""
    from toad_amd import ops
    model, _, x = _model_and_bag(cuda, n, seed=n + 1)
    x = x.clone(); x[:, :40] *= 1e-4; x[n // 2, 900] = 777.0                            # one poisoned item among ordinary ones
    w = {k: v.detach() for k, v in model._weights().items()}
    xg = x.to(cuda)
    arena = ops.mil_fwd(w, xg, torch.ones(1, device=cuda))
    assert torch.equal(arena.view("x_amax", (ops.amax_floats(n),)), ops.absmax_rows256(xg))

