"""GPU: each C-ABI kernel against its CPU twin in the oracle on the same seeded inputs.
Tolerances: fp32-equivalent GEMMs 1e-5 of the output scale (fp16 two-piece operands, three MFMA terms, fp32 accumulation: as close to
the exact value as an fp32 fma chain, differing from MKL by summation order and round-off); fused pool 1e-5 abs (v_exp/v_rcp based tanh/sigmoid, abs err ~1e-7/elem)."""
import pytest
import torch

from oracle import toad_oracle as orc

pytestmark = pytest.mark.gpu


def _rel(a, b, floor=1e-3):
    # error relative to the reference's own scale; the floor covers exactly-zero references
    # (N == 1: softmax of one element has zero gradient, both sides are pure roundoff)
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), floor)


@pytest.mark.parametrize("m,k,n", [(1, 1024, 512), (63, 512, 512), (129, 512, 768), (777, 1024, 512), (4096, 512, 768),
                                   (300, 36, 132)])
@pytest.mark.parametrize("act", [0, 1])
def test_linear_fwd(cuda, m, k, n, act):
    from toad_amd import ops
    g = torch.Generator().manual_seed(m * 7 + n)
    x = torch.randn(m, k, generator=g); w = torch.randn(n, k, generator=g) * 0.05; b = torch.randn(n, generator=g)
    y = ops.linear_act_fwd(x.to(cuda), w.to(cuda), b.to(cuda), act).cpu()
    ref = orc.linear_act(x, w, b, "relu" if act else "none")
    assert _rel(y, ref) <= 1e-5
    y2 = ops.linear_act_fwd(x.to(cuda), w.to(cuda), None, act).cpu()
    assert _rel(y2, orc.linear_act(x, w, torch.zeros(n), "relu" if act else "none")) <= 1e-5


def test_linear_fwd_is_transpose_sensitive(cuda):
    """A = I with an asymmetric B catches swapped C-write indices (guide rule 16)."""
    from toad_amd import ops
    n = 256
    x = torch.eye(n)
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) / 251.0
    y = ops.linear_act_fwd(x.to(cuda), w.to(cuda), None, 0).cpu()
    # (two fp16 pieces carry 22-23 significand bits: 1.0 * w is w to 2^-22, not bit for bit - a swapped index is off by O(1))
    assert (y - w.t()).abs().max().item() <= 2.0 ** -21


@pytest.mark.parametrize("m,n,k", [(1, 512, 512), (65, 768, 512), (1000, 512, 512), (5000, 768, 512)])
def test_linear_dgrad(cuda, m, n, k):
    from toad_amd import ops
    g = torch.Generator().manual_seed(m + n)
    dy = torch.randn(m, n, generator=g); w = torch.randn(n, k, generator=g) * 0.05
    add = torch.randn(m, k, generator=g); src = torch.randn(m, k, generator=g)
    wt = ops.transpose(w.to(cuda))
    assert torch.equal(wt.cpu(), w.t().contiguous())
    dx = ops.linear_dgrad(dy.to(cuda), wt).cpu()
    assert _rel(dx, dy @ w) <= 1e-5
    dx = ops.linear_dgrad(dy.to(cuda), wt, add.to(cuda), src.to(cuda)).cpu()
    ref = (dy @ w + add) * (src > 0).float()
    assert _rel(dx, ref) <= 1e-5
    buf = add.to(cuda)                                         # in place over the addend
    ops.linear_dgrad(dy.to(cuda), wt, buf, src.to(cuda), out=buf)
    assert _rel(buf.cpu(), ref) <= 1e-5


@pytest.mark.parametrize("m,n,k", [(1, 512, 512), (31, 768, 512), (257, 512, 1024), (3000, 512, 512), (20000, 768, 512),
                                   (500, 132, 36)])
def test_linear_wgrad(cuda, m, n, k):
    from toad_amd import ops
    g = torch.Generator().manual_seed(m + k)
    dy = torch.randn(m, n, generator=g); x = torch.randn(m, k, generator=g)
    dw, db = ops.linear_wgrad(dy.to(cuda), x.to(cuda))
    ref_w = (dy.double().t() @ x.double()).float(); ref_b = dy.double().sum(0).float()
    assert _rel(dw.cpu(), ref_w) <= 2e-5 and _rel(db.cpu(), ref_b) <= 2e-5
    # accumulate: beta = 1 on top of an existing gradient; run twice -> bitwise reproducible
    base_w = torch.randn(n, k, generator=g); base_b = torch.randn(n, generator=g)
    outs = []
    for _ in range(2):
        aw, ab = base_w.to(cuda), base_b.to(cuda)
        ops.linear_wgrad(dy.to(cuda), x.to(cuda), aw, ab, beta=1.0)
        outs.append((aw.cpu(), ab.cpu()))
    assert _rel(outs[0][0], ref_w + base_w) <= 2e-5 and _rel(outs[0][1], ref_b + base_b) <= 2e-5
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _pool_inputs(n, d, l, t, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(n, 2 * d, generator=g) * scale
    h = torch.randn(n, l, generator=g).relu()
    wc = torch.randn(t, d, generator=g) * 0.1
    bc = torch.randn(t, generator=g) * 0.1
    return p, h, wc, bc


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 63, 64, 65, 1000, 12289, 50000])
@pytest.mark.parametrize("d,l,t", [(384, 512, 2)])
def test_gated_pool_fwd(cuda, n, d, l, t):
    from toad_amd import ops
    p, h, wc, bc = _pool_inputs(n, d, l, t, n)
    a_raw, m, stats = ops.gated_pool_fwd(p.to(cuda), d, h.to(cuda), wc.to(cuda), bc.to(cuda))
    ra, rm = orc.gated_pool_fwd(p[:, :d], p[:, d:], h, wc, bc)
    assert (a_raw.cpu() - ra).abs().max().item() <= 1e-5
    assert (m.cpu() - rm).abs().max().item() <= 1e-5
    assert (stats[:, 0].cpu() - ra.max(0).values).abs().max().item() <= 1e-5
    lse = torch.logsumexp(ra.double(), 0).float()
    assert ((stats[:, 0] + stats[:, 1].log()).cpu() - lse).abs().max().item() <= 1e-5
    only, _, _ = ops.gated_pool_fwd(p.to(cuda), d, None, wc.to(cuda), bc.to(cuda))
    assert torch.equal(only, a_raw)


@pytest.mark.parametrize("d,l,t", [(256, 512, 2), (256, 1024, 1), (384, 1024, 2), (384, 512, 1),
                                   # shapes only the covering instantiation serves (any L, D, n_tasks Attn_Net_Gated's constructor takes
                                   # up to L 1024, D 512, 4 tasks: models/model_toad.py:19): multiples of 128, odd sizes, more tasks
                                   (128, 768, 3), (512, 640, 4), (100, 200, 1), (512, 1024, 4), (384, 512, 3), (4, 8, 1), (256, 1024, 2), (260, 520, 2)])
def test_gated_pool_other_shapes(cuda, d, l, t):
    from toad_amd import ops
    n = 777
    p, h, wc, bc = _pool_inputs(n, d, l, t, d + l + t)
    a_raw, m, _ = ops.gated_pool_fwd(p.to(cuda), d, h.to(cuda), wc.to(cuda), bc.to(cuda))
    ra, rm = orc.gated_pool_fwd(p[:, :d], p[:, d:], h, wc, bc)
    assert (a_raw.cpu() - ra).abs().max().item() <= 1e-5 and (m.cpu() - rm).abs().max().item() <= 1e-5
    dm = torch.randn(t, l)
    dp, dh, dwc, dbc = ops.gated_pool_bwd(p.to(cuda), d, h.to(cuda), wc.to(cuda), a_raw, _, m, dm.to(cuda))
    rpa, rpb, rdh, rdwc, rdbc = orc.gated_pool_bwd(p[:, :d], p[:, d:], h, wc, ra, rm, dm)
    assert _rel(dp.cpu(), torch.cat([rpa, rpb], 1)) <= 5e-5 and _rel(dh.cpu(), rdh) <= 5e-5
    assert _rel(dwc.cpu(), rdwc) <= 5e-5
    assert (dbc.cpu() - rdbc).abs().max().item() <= 5e-5          # sum_i dS[i,t] == 0 without an external dA


def test_gated_pool_softmax_saturation_and_rescale(cuda):
    """Forces the online-softmax rescale branch: scores rise steadily along the bag and one late
    row dominates (guide rule 26: a rare data-dependent branch needs an input that takes it)."""
    from toad_amd import ops
    d, l, t, n = 384, 512, 2, 4099
    p, h, wc, bc = _pool_inputs(n, d, l, t, 11)
    wc = wc.abs() * 3.0
    ramp = torch.linspace(-6, 6, n)[:, None]
    p = torch.cat([ramp.expand(n, d) + 0.1 * p[:, :d], 4.0 + 0.0 * p[:, d:]], 1).contiguous()
    p[n - 7, :d] = 15.0                                             # spike
    a_raw, m, _ = ops.gated_pool_fwd(p.to(cuda), d, h.to(cuda), wc.to(cuda), bc.to(cuda))
    ra, rm = orc.gated_pool_fwd(p[:, :d], p[:, d:], h, wc, bc)
    assert (ra.max(0).values - ra.min(0).values).min().item() > 50           # softmax really saturates
    assert (a_raw.cpu() - ra).abs().max().item() <= 1e-4 and (m.cpu() - rm).abs().max().item() <= 1e-5
    # extreme pre-activations: tanh/sigmoid saturate, nothing overflows
    p2 = torch.cat([torch.full((64, d), 90.0), torch.full((64, d), -95.0)], 1)
    p2[::2] *= -1
    a2, m2, _ = ops.gated_pool_fwd(p2.to(cuda), d, h[:64].to(cuda), wc.to(cuda), bc.to(cuda))
    r2, rm2 = orc.gated_pool_fwd(p2[:, :d], p2[:, d:], h[:64], wc, bc)
    assert torch.isfinite(a2).all() and (a2.cpu() - r2).abs().max().item() <= 1e-4
    assert (m2.cpu() - rm2).abs().max().item() <= 1e-5


def test_gated_pool_uniform_attention(cuda):
    """All rows equal -> uniform attention -> M equals the common row; A_raw constant."""
    from toad_amd import ops
    d, l, t, n = 384, 512, 2, 3001
    p, h, wc, bc = _pool_inputs(1, d, l, t, 5)
    p = p.expand(n, 2 * d).contiguous(); h = h.expand(n, l).contiguous()
    a_raw, m, stats = ops.gated_pool_fwd(p.to(cuda), d, h.to(cuda), wc.to(cuda), bc.to(cuda))
    assert (a_raw - a_raw[0]).abs().max().item() == 0.0
    assert (m.cpu() - h[0]).abs().max().item() <= 1e-5
    assert (stats[:, 1].cpu() - n).abs().max().item() <= 1e-2


@pytest.mark.parametrize("n", [1, 5, 64, 1000, 20001])
def test_gated_pool_bwd(cuda, n):
    from toad_amd import ops
    d, l, t = 384, 512, 2
    p, h, wc, bc = _pool_inputs(n, d, l, t, 100 + n)
    g = torch.Generator().manual_seed(n)
    dm = torch.randn(t, l, generator=g); da = torch.randn(n, t, generator=g) * 0.1
    a_raw, m, stats = ops.gated_pool_fwd(p.to(cuda), d, h.to(cuda), wc.to(cuda), bc.to(cuda))
    ra, rm = orc.gated_pool_fwd(p[:, :d], p[:, d:], h, wc, bc)
    for ext in (None, da):
        dp, dh, dwc, dbc = ops.gated_pool_bwd(p.to(cuda), d, h.to(cuda), wc.to(cuda), a_raw, stats, m, dm.to(cuda),
                                              None if ext is None else ext.to(cuda))
        rpa, rpb, rdh, rdwc, rdbc = orc.gated_pool_bwd(p[:, :d], p[:, d:], h, wc, ra, rm, dm, ext)
        # n == 1 without an external dA: softmax of one element -> dS == 0 exactly; both sides are
        # cancellation noise ~1e-6, so the floor of the scale is raised there
        fl = 0.1 if n == 1 else 1e-3
        assert _rel(dp.cpu(), torch.cat([rpa, rpb], 1), floor=fl) <= 5e-5
        assert _rel(dh.cpu(), rdh) <= 2e-5
        assert _rel(dwc.cpu(), rdwc, floor=fl) <= 5e-5
        assert (dbc.cpu() - rdbc).abs().max().item() <= 5e-5 * max(rdbc.abs().max().item(), 1.0)
    # beta accumulate
    bw = torch.randn(t, d, generator=g); bb = torch.randn(t, generator=g)
    aw, ab = bw.to(cuda), bb.to(cuda)
    ops.gated_pool_bwd(p.to(cuda), d, h.to(cuda), wc.to(cuda), a_raw, stats, m, dm.to(cuda), da.to(cuda), aw, ab, 1.0)
    assert _rel(aw.cpu(), rdwc + bw) <= 5e-5 and _rel(ab.cpu(), rdbc + bb) <= 5e-5


@pytest.mark.parametrize("c", [2, 18, 33])
def test_heads_and_loss(cuda, c):
    from toad_amd import ops
    l = 512
    g = torch.Generator().manual_seed(c)
    m = torch.randn(2, l, generator=g); sex = torch.tensor([1.0])
    wcls = torch.randn(c, l + 1, generator=g) * 0.1; bcls = torch.randn(c, generator=g) * 0.1
    wsite = torch.randn(2, l + 1, generator=g) * 0.1; bsite = torch.randn(2, generator=g) * 0.1
    outs = ops.heads_fwd(*(t.to(cuda) for t in (m, sex, wcls, bcls, wsite, bsite)))
    ref = orc.heads_fwd(m, sex, wcls, bcls, wsite, bsite)
    for o, r in zip(outs, ref):
        if o.dtype == torch.int64:
            assert torch.equal(o.cpu(), r)
        else:
            assert (o.cpu() - r).abs().max().item() <= 1e-5
    label = torch.tensor([c - 1]); site = torch.tensor([1])
    loss, dl, ds = ops.mtl_ce_fwd_bwd(outs[1], outs[4], label.to(cuda), site.to(cuda))
    rl = orc.loss_fn(ref[1], label, ref[4], site)
    rdl, rds = orc.loss_grad(ref[1], label, ref[4], site)
    assert abs(loss[0].item() - rl.item()) <= 1e-5
    assert (dl.cpu() - rdl).abs().max().item() <= 1e-6 and (ds.cpu() - rds).abs().max().item() <= 1e-6
    ext = torch.randn(2, l + 1, generator=g)
    dwcls, dbcls, dwsite, dbsite, dm = ops.heads_bwd(outs[0], dl, ds, wcls.to(cuda), wsite.to(cuda), ext.to(cuda))
    rw, rb, rws, rbs, rdm = orc.heads_bwd(ref[0], rdl, rds, wcls, wsite)
    assert _rel(dwcls.cpu(), rw) <= 1e-5 and _rel(dwsite.cpu(), rws) <= 1e-5
    assert _rel(dbcls.cpu(), rb) <= 1e-5 and _rel(dbsite.cpu(), rbs) <= 1e-5
    assert _rel(dm.cpu(), rdm + ext[:, :l]) <= 1e-5
