"""GPU: the ragged multi-slide training step (toad_mil_multi_step_f32): gradients == the sum of the per-slide gradients - against
B calls of the one-slide step (same kernels) and against the CPU oracle (the reference's op sequence, oracle/toad_oracle.py)."""
import pytest
import torch

from oracle import toad_oracle as orc
from tests.helpers import SLOT2KEY, assert_grad_close, assert_grad_close_or_few_flips, check_batch_against_oracle, grad_scale

pytestmark = pytest.mark.gpu

C = 18


def _model(cuda, seed=0, dropout=False):
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(seed)
    m = TOAD_fc_mtl_concat(n_classes=C, dropout=dropout)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.relocate()
    return m, params


def _slides(lens, seed=0):
    out = []
    for i, n in enumerate(lens):
        g = torch.Generator().manual_seed(seed * 1000 + i)
        out.append((torch.randn(n, 1024, generator=g), torch.tensor([float(i % 2)]), torch.tensor([(7 * i) % C]), torch.tensor([(i // 2) % 2])))
    return out


@pytest.mark.parametrize("lens", [[256] * 8, [1, 2, 63, 300, 1000, 257, 64], [3000, 5000, 777], [10000] * 4, [40]])
def test_batch_gradient_is_the_sum_of_the_slide_gradients(cuda, lens):
    from toad_amd import ops
    model, params = _model(cuda, seed=len(lens))
    w = {k: v.detach() for k, v in model._weights().items()}
    slides = _slides(lens, seed=len(lens))
    dev_slides = [tuple(t.to(cuda) for t in s) for s in slides]
    B = len(lens)
    # reference 1: B one-slide steps accumulating (beta = 1) into one buffer, each loss scaled by 1/B
    g1 = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    l1, lg1 = [], []
    for i, (bag, sex, label, site) in enumerate(dev_slides):
        if bag.shape[0] >= 1:
            loss, lg, sl = ops.mil_step(w, g1, 0.0 if i == 0 else 1.0, bag, sex, label, site, 0.75 / B, 0.25 / B, want_logits=True)
        l1.append(loss); lg1.append(lg)
    # the batch call
    g2 = {k: torch.full_like(w[k], 3.0) for k in ops.STEP_SLOTS}              # beta = 0 must overwrite
    sex = torch.cat([s[1] for s in dev_slides]); label = torch.cat([s[2] for s in dev_slides]); site = torch.cat([s[3] for s in dev_slides])
    loss, logits, slog = ops.mil_multi_step(w, g2, 0.0, [s[0] for s in dev_slides], sex, label, site, 0.75 / B, 0.25 / B, want_logits=True)
    assert loss.shape == (B, 3) and logits.shape == (B, C)
    for i in range(B):
        assert (logits[i] - lg1[i][0]).abs().max().item() <= 2e-5 * max(lg1[i].abs().max().item(), 1.0), i
        assert (loss[i] - l1[i]).abs().max().item() <= 1e-5, i
    # The oracle's per-slide gradients, summed - in fp32 (the reference's arithmetic) and in fp64. The two device routes split the GEMMs'
    # reductions differently (K-slices of remainder tiles depend on the row count), so H1 / H differ by fp32 round-off and a pre-activation
    # within round-off of zero can land on the other side of the ReLU in one of them: a LEGITIMATE flip moves one dZ element, which moves
    # a whole row / a rank-one update of the trunk gradients by that patch's contribution. The yardstick for that is, as in the golden
    # tests, the oracle's OWN fp32-vs-fp64 deviation `dev` (its fp32 run flips too); the ten gradients no ReLU mask reaches have none.
    p64 = {k: v.double() for k, v in params.items()}
    tot32 = {k: torch.zeros_like(v) for k, v in params.items()}
    tot64 = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in params.items()}
    for (bag, sx, lb, st) in slides:
        _, _, g = orc.fwd_bwd(params, bag, sx, lb, st)
        _, _, gd = orc.fwd_bwd(p64, bag.double(), sx.double(), lb, st)
        for k in tot32:
            tot32[k] += g[k] / B; tot64[k] += gd[k] / B
    d = w["wc"].shape[1]

    def full(g):
        o = dict(g); o["wa"], o["wb"], o["ba"], o["bb"] = g["wab"][:d], g["wab"][d:], g["bab"][:d], g["bab"][d:]
        return o
    got1, got2 = full(g1), full(g2)
    for slot, key in SLOT2KEY.items():
        dev = (tot32[key].double() - tot64[key]).abs().max().item()
        sc = grad_scale(tot64, key)
        # (`dev` only knows the oracle's own flips; the device can flip where the CPU did not: trunk gradients go through the rank-one test)
        chk = assert_grad_close_or_few_flips if slot in ("w1", "b1", "w2", "b2") else assert_grad_close
        chk(got2[slot], tot64[key], 2e-5, sc, what=f"batch vs oracle: {key}", floor=10.0 * dev)
        # and the batch call against B one-slide calls of the same kernels
        chk(got2[slot], got1[slot], 5e-5, sc, what=f"batch vs per-slide: {key}", floor=20.0 * dev)
    # beta = 1 accumulates on top
    g3 = {k: g2[k].clone() for k in ops.STEP_SLOTS}
    ops.mil_multi_step(w, g3, 1.0, [s[0] for s in dev_slides], sex, label, site, 0.75 / B, 0.25 / B)
    for k in ops.STEP_SLOTS:
        assert (g3[k] - 2 * g2[k]).abs().max().item() <= 1e-6 * max(g2[k].abs().max().item(), 1e-30) + 1e-12, k
    # deterministic run to run, and the pre-concatenated form is the same call
    g4 = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    xcat = torch.cat([s[0] for s in dev_slides], 0)
    offs = [0]
    for n in lens:
        offs.append(offs[-1] + n)
    ops.mil_multi_step(w, g4, 0.0, xcat, sex, label, site, 0.75 / B, 0.25 / B, offsets=offs)
    assert all(torch.equal(g2[k], g4[k]) for k in ops.STEP_SLOTS)


def _step_workspace_activations(n, nb, c, d, dev):
    """Test-only view of what toad_mil_multi_step_f32 left in the cached step workspace: the forward arena sits at the aligned base of the
    workspace (csrc/step.hip: align_base + arena_layout), so H1, H, P and A_raw of the CONCATENATION can be read back after the call."""
    import ctypes
    from toad_amd import _lib, ops
    lib = _lib.load()
    ws = ops._ws(int(lib.toad_mil_multi_ws_bytes(n, nb, c, d)), dev, "step")
    base = (-ws.data_ptr()) % int(lib.toad_mil_buffer_align(n))
    offs = (ctypes.c_int64 * len(ops.ARENA_SLOTS))()
    _lib.check(lib.toad_mil_arena_layout(n, c, d, offs), "toad_mil_arena_layout")
    off = dict(zip(ops.ARENA_SLOTS, (base + int(o) for o in offs)))

    def view(name, cols):
        return ws[off[name]:off[name] + n * cols * 4].view(torch.float32).view(n, cols)
    return view("h1", 512), view("h", 512), view("p", 2 * d), view("a_raw", 2)


# The shapes bench.py --config 4 actually calls (two 50,000-patch slides per toad_mil_multi_step_f32 = 100,000 concatenated rows) and their
# neighbours. Everything above 40,000 rows switches on three code paths no smaller batch reaches:
#   (a) staggered workgroup starts of the persistent NT GEMMs (csrc/gemm_f32.hip launch_nt_h2: es.stagger for launches of >= 512 tiles);
#   (b) dP's abs-max array taken from the per-row |dP| bound the batched pool backward leaves in dZ1's buffer, folded 256 rows per slot
#       (csrc/step.hip: launch_pool_bwd_batch(..., w.amax_dP, w.dZ1, N)) - slide boundaries off the 256-row grid matter here;
#   (c) the attention dgrad on the one-bit ReLU image with the BATCHED pooled addend recomputed per row from records {w0, w1, slide} and the slides'
#       dM (gemm_nt_h2_big_kernel<true, false, 2, 4>, gemm_h2_epilogue.inc PBATCH; round 4 read a materialised addend through <true, false, 2, 0>).
# Reference semantics: utils/core_utils_mtl_concat.py:200-234, one forward / loss / backward per slide; the batch gradient is their sum.
BIG_BATCHES = [("config4_pair", [50000, 50000], 0.0), ("batch_rows_limit", [65536, 65536], 0.0), ("off_grid_boundaries", [50000, 30001, 20000], 0.0),
               ("dropout_70k_rows", [40000, 30001], 0.25),
               # round 5 raised the default call size to 524,288 rows (toad_amd/dp.py BATCH_ROWS: ten 50k-patch slides per call): one batch of exactly that size
               ("default_call_size_524288_rows", [100000, 50000, 50000, 100000, 100000, 100000, 24288], 0.0),
               # ... and the per-slide limit to 262,144 patches (BATCH_MAX_PATCHES): both limits at once; the extremes `bench.py --config 4 --ragged` sends
               # through the call (197k- and 9k-patch slides, boundaries off the 256-row grid); a slide above 100k patches under train-mode dropout
               ("both_limits_2x262144", [262144, 262144], 0.0),
               ("ragged_extremes_197k_9k_131k", [197000, 9000, 131072], 0.0),
               ("dropout_slide_above_100k", [120001, 30000], 0.25)]


@pytest.mark.parametrize("name,lens,drop_p", BIG_BATCHES, ids=[b[0] for b in BIG_BATCHES])
def test_config4_shape_batches_match_the_oracle(cuda, name, lens, drop_p):
    """Value check of the ragged multi-slide step at config 4's own call shape (see BIG_BATCHES). Per slide: logits, site logits and loss against
    the oracle's fp32 forward (1e-4, the north star's bound); H1 / H of the concatenation against the exact fp64 forward (1e-4 of their abs-max),
    every ReLU-mask difference from the exact forward shown to be legitimate (pre-activation within round-off of zero) and rare; all 14 gradients
    against the SUM over slides of the oracle's fp64 backward evaluated on the device's own activations (identical masks, so no flip allowance is
    needed): 2e-5 of each gradient's own scale, plus the measured fp32 noise of the same backward for the cancellation-dominated ones. With
    drop_p the device's masks are exported (toad_dropout_mask_f32) and fed to the oracle, as in the small-batch dropout test."""
    from toad_amd import functional as F_, ops
    model, params = _model(cuda, seed=len(lens) + 40, dropout=drop_p > 0)
    if drop_p > 0:                                            # dropout=True shifts the state-dict indices; the oracle uses the dropout=False names
        rename = {"attention_net.3.": "attention_net.2.", "attention_net.6.": "attention_net.4."}
        params = {next((b + k[len(a):] for a, b in rename.items() if k.startswith(a)), k): v for k, v in params.items()}
    w = {k: v.detach() for k, v in model._weights().items()}
    slides = _slides(lens, seed=77)
    B, ntot, d = len(lens), sum(lens), w["wc"].shape[1]
    xcat = torch.cat([s[0] for s in slides], 0).to(cuda)
    sex = torch.cat([s[1] for s in slides]).to(cuda); label = torch.cat([s[2] for s in slides]).to(cuda); site = torch.cat([s[3] for s in slides]).to(cuda)
    offs = [0]
    for n in lens:
        offs.append(offs[-1] + n)
    seed = 20260927
    g = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    loss, logits, slog = ops.mil_multi_step(w, g, 0.0, xcat, sex, label, site, 0.75 / B, 0.25 / B, drop_p=drop_p, seed=seed, want_logits=True, offsets=offs)
    torch.cuda.synchronize()
    h1_d, h_d, p_d, a_d = (t.cpu() for t in _step_workspace_activations(ntot, B, C, d, cuda))
    loss, logits, slog = loss.cpu(), logits.cpu(), slog.cpu()
    G, mask64 = 0x9E3779B97F4A7C15, 0xFFFFFFFFFFFFFFFF
    masks = None
    if drop_p > 0:
        s1, s2, sa, sb = F_.drop_seeds(seed)
        mk_h1 = ops.dropout_mask(ntot * 512, drop_p, s1, cuda).reshape(ntot, 512).cpu()
        mk_h = ops.dropout_mask(ntot * 512, drop_p, s2, cuda).reshape(ntot, 512).cpu()
        masks = [{"h1": mk_h1[offs[b]:offs[b + 1]], "h": mk_h[offs[b]:offs[b + 1]],
                  "a": ops.dropout_mask(lens[b] * d, drop_p, (sa + 2 * b * G) & mask64, cuda).reshape(lens[b], d).cpu(),
                  "b": ops.dropout_mask(lens[b] * d, drop_p, (sb + 2 * b * G) & mask64, cuda).reshape(lens[b], d).cpu()} for b in range(B)]

    def full(gd):
        o = dict(gd); o["wa"], o["wb"], o["ba"], o["bb"] = gd["wab"][:d], gd["wab"][d:], gd["bab"][:d], gd["bab"][d:]
        return o
    got = full(g)
    tot32, tot64, flips = check_batch_against_oracle(name, params, slides, offs, dict(h1=h1_d, h=h_d, p=p_d, a_raw=a_d, logits=logits, site_logits=slog, loss=loss),
                                                     {s_: got[s_].cpu() for s_ in SLOT2KEY}, masks)
    if drop_p == 0.0:
        # and against B one-slide steps of the same kernels, on the ten gradients no ReLU mask reaches (the two routes scale their GEMM operands per
        # 256-row block of different row ranges, so a pre-activation at round-off of zero may fall either way between them)
        g1 = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
        for i in range(B):
            ops.mil_step(w, g1, 0.0 if i == 0 else 1.0, xcat[offs[i]:offs[i + 1]], sex[i:i + 1], label[i:i + 1], site[i:i + 1], 0.75 / B, 0.25 / B)
        got1 = full(g1)
        for slot, key in SLOT2KEY.items():
            if slot not in ("w1", "b1", "w2", "b2"):
                noise = (tot32[key] - tot64[key]).abs().max().item()
                assert_grad_close(got[slot], got1[slot], 5e-5, grad_scale(tot64, key), what=f"{name}: batch vs per-slide: {key}", floor=32.0 * noise)


def test_dp_step_batches_small_slides(cuda):
    """SlideShardedDP.step routes a shard of small slides through the multi-slide call by default: same parameters after the step
    (to round-off) as with the per-slide path, and far fewer library calls."""
    from toad_amd.dp import SlideShardedDP
    lens = [300, 512, 64, 1000, 2049, 128]
    slides = [tuple(t.to(cuda) for t in s) for s in _slides(lens, seed=9)]
    outs = []
    for batched in (False, True):
        model, _ = _model(cuda, seed=4)
        model.train()
        dp = SlideShardedDP(model, {"lr": 1e-3, "weight_decay": 1e-5})
        dp.accumulate(slides, len(slides), batched=batched)
        outs.append(dp.flat_grad.clone())
        if batched:
            losses = dp.step(slides, len(slides))           # default = batched for such a shard
            assert len(losses) == len(slides) and torch.isfinite(torch.stack([l[0] for l in losses])).all()
    sc = outs[0].abs().max().item()
    assert (outs[0] - outs[1]).abs().max().item() <= 5e-5 * sc


def test_adjacent_bags_run_without_a_copy_and_give_the_same_gradients(cuda):
    """Bags cut from one resident buffer (bench.py config 4, an ingest buffer) are taken as their own concatenation: bitwise the results of the
    same bags passed as separate tensors (which torch.cat copies), and SlideShardedDP batches two 50,000-patch slides into one call."""
    from toad_amd import ops
    from toad_amd.dp import SlideShardedDP
    model, _ = _model(cuda, seed=6)
    w = {k: v.detach() for k, v in model._weights().items()}
    lens = [700, 64, 1300]
    pool = torch.randn(sum(lens), 1024, device=cuda, generator=torch.Generator(device=cuda).manual_seed(3))
    views, off = [], 0
    for n in lens:
        views.append(pool[off:off + n]); off += n
    sex = torch.tensor([0.0, 1.0, 1.0], device=cuda); label = torch.tensor([1, 5, 9], device=cuda); site = torch.tensor([0, 1, 0], device=cuda)
    res = []
    for bags in (views, [v.clone() for v in views]):
        g = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
        loss, _, _ = ops.mil_multi_step(w, g, 0.0, bags, sex, label, site, 0.25, 0.08)
        res.append((loss.clone(), {k: v.clone() for k, v in g.items()}))
    assert torch.equal(res[0][0], res[1][0]) and all(torch.equal(res[0][1][k], res[1][1][k]) for k in ops.STEP_SLOTS)
    assert ops._adjacent_rows(views).data_ptr() == pool.data_ptr()
    assert SlideShardedDP.BATCH_MAX_PATCHES >= 50000 and 2 * 50000 <= SlideShardedDP.BATCH_ROWS


def test_train_mode_dropout_runs_and_is_seeded(cuda):
    from toad_amd import ops
    model, _ = _model(cuda, seed=2, dropout=True)
    w = {k: v.detach() for k, v in model._weights().items()}
    slides = [tuple(t.to(cuda) for t in s) for s in _slides([500, 700, 300], seed=3)]
    sex = torch.cat([s[1] for s in slides]); label = torch.cat([s[2] for s in slides]); site = torch.cat([s[3] for s in slides])
    res = []
    for seed in (11, 11, 12):
        g = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
        loss, _, _ = ops.mil_multi_step(w, g, 0.0, [s[0] for s in slides], sex, label, site, drop_p=0.25, seed=seed)
        res.append((loss.clone(), g["w1"].clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert not torch.equal(res[0][1], res[2][1]) and torch.isfinite(res[2][1]).all()


def test_train_mode_dropout_matches_the_oracle_with_reproduced_masks(cuda):
    """Train-mode dropout in the ragged step: the trunk masks hash the element index in the CONCATENATION (seeds s1, s2), the pooling masks
    the index inside each slide with the slide's own pair of seeds (sa + 2bG, sb + 2bG: no slide's tanh mask equals another slide's sigmoid
    mask). The masks are reproduced with toad_dropout_mask_f32 and fed to the oracle slide by slide: per-slide logits and losses agree."""
    from toad_amd import functional as F_, ops
    model, _ = _model(cuda, seed=5, dropout=True)
    sd = model.state_dict()
    rename = {"attention_net.3.": "attention_net.2.", "attention_net.6.": "attention_net.4."}      # dropout=True key names -> the oracle's
    params = {}
    for k, v in sd.items():
        for a, b in rename.items():
            if k.startswith(a):
                k = b + k[len(a):]
        params[k] = v.detach().cpu().clone()
    w = {k: v.detach() for k, v in model._weights().items()}
    lens = [300, 64, 777, 300]
    slides = _slides(lens, seed=8)
    dev = [tuple(t.to(cuda) for t in s) for s in slides]
    sex = torch.cat([s[1] for s in dev]); label = torch.cat([s[2] for s in dev]); site = torch.cat([s[3] for s in dev])
    seed, B, ntot = 123456789, len(lens), sum(lens)
    g = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    loss, logits, slog = ops.mil_multi_step(w, g, 0.0, [s[0] for s in dev], sex, label, site, 0.75 / B, 0.25 / B, drop_p=F_.DROP_P, seed=seed, want_logits=True)
    s1, s2, sa, sb = F_.drop_seeds(seed)
    m_h1 = ops.dropout_mask(ntot * 512, F_.DROP_P, s1, cuda).reshape(ntot, 512).cpu()
    m_h = ops.dropout_mask(ntot * 512, F_.DROP_P, s2, cuda).reshape(ntot, 512).cpu()
    G, mask64 = 0x9E3779B97F4A7C15, 0xFFFFFFFFFFFFFFFF
    off, pool_masks = 0, []
    for b, (x, sx, lb, st) in enumerate(slides):
        n = lens[b]
        mk = {"h1": m_h1[off:off + n], "h": m_h[off:off + n],
              "a": ops.dropout_mask(n * 384, F_.DROP_P, (sa + 2 * b * G) & mask64, cuda).reshape(n, 384).cpu(),
              "b": ops.dropout_mask(n * 384, F_.DROP_P, (sb + 2 * b * G) & mask64, cuda).reshape(n, 384).cpu()}
        pool_masks.append(mk)
        o_out, o_loss, _ = orc.fwd_bwd(params, x, sx, lb, st, masks=mk)
        assert (logits[b].cpu() - o_out["logits"][0]).abs().max().item() <= 1e-4, b
        assert (slog[b].cpu() - o_out["site_logits"][0]).abs().max().item() <= 1e-4, b
        assert abs(loss[b][0].item() * B - float(o_loss)) <= 1e-4, b
        off += n
    # equal-length slides 0 and 3: four different pooling masks, none shared across slides or branches
    ms = [pool_masks[0]["a"], pool_masks[0]["b"], pool_masks[3]["a"], pool_masks[3]["b"]]
    assert all(not torch.equal(ms[i], ms[j]) for i in range(4) for j in range(i + 1, 4))


def test_train_loop_dp_learns_on_small_bags(cuda):
    """train_loop_dp: one optimiser step per batch of slides through the ragged multi-slide call; the class loss falls on a learnable signal
    and the epoch statistics equal the mean of the per-slide losses the step reports."""
    from toad_amd.dp import SlideShardedDP
    from toad_amd.train import train_loop_dp
    model, _ = _model(cuda, seed=21)
    model.train()
    dp = SlideShardedDP(model, {"lr": 5e-4, "weight_decay": 1e-5})
    slides = []
    for i in range(48):
        g = torch.Generator().manual_seed(400 + i)
        x = torch.randn(96 + (i * 31) % 200, 1024, generator=g)
        x[:, :16] += (i % 3) * 1.0
        slides.append((x, torch.tensor([i % 3]), torch.tensor([i % 2]), torch.tensor([float(i % 2)])))
    first = train_loop_dp(0, dp, slides, batch_slides=16)
    for e in range(1, 6):
        last = train_loop_dp(e, dp, slides, batch_slides=16)
    assert first["slides"] == 48 and last["cls_loss"] < 0.8 * first["cls_loss"], (first, last)
