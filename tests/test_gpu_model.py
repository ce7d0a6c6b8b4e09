"""GPU: the drop-in nn.Module end to end (through the C ABI) against the committed golden
vectors (the reference's outputs) and against the oracle on random bags; full-size properties."""
import pytest
import torch

from oracle import toad_oracle as orc
from tests.helpers import (LAYER2_KEYS, MASK_FREE_KEYS, SLOT2KEY, assert_grad_close, assert_step_grad_matches_per_op, case_inputs, check_activations_vs_golden,
                           check_outputs_vs_golden, check_trunk_grads_vs_golden_blocks, grad_scale, relu_flip_positions)

pytestmark = pytest.mark.gpu

CASES = ["n0", "n1", "n2", "n63", "n64", "n65", "n256", "n777", "n777_c2", "n1024_sat", "n300_equal", "n10000", "n100000", "r256", "r10000", "r100000"]
FLIPS = {}              # case -> (legitimate ReLU-mask flips in layer 1, in layer 2), filled by test_module_matches_reference_golden
# Of the 13 golden cases, at most this many may have ANY gradient compared with something other than the reference's golden values.
# Measured on MI355X (round 3, gpurun_out/golden_flip_cases.json): 6 - n256 (0 flips in layer 1, 1 in layer 2), n777 (1, 1), n777_c2 (1, 1),
# n1024_sat (0, 3), n10000 (4, 1), n100000 (21, 22 of 2 x 51.2 M mask elements): the closed-form golden bags (cos / sin of the indices) put
# pre-activations within fp32 round-off of zero. In those cases the TEN mask-free gradients are still held to the golden values; only the four
# trunk gradients (dW1, db1, dW2, db2) go to the fp64 backward on the device's own activations. The other 7 cases compare all 14.
MAX_FLIP_CASES = 7
# Round 5: the second golden family, N(0,1) bags with Xavier-normal weights (r256 / r10000 / r100000, oracle/pin_against_reference.py): no
# pre-activation is parked at round-off of zero and both layers' pre-activations have std ~1, so a flip needs |z| < ~1e-7: expected 0.02 / 0.7 / 7
# flipped mask elements of 2.6e5 / 1e7 / 1e8. r256 compares all 14 gradients element-wise with the reference's values on every run, r10000 on most;
# r100000 keeps the flip machinery (1e8 mask elements cannot all be further than round-off from zero). The closed-form weights of the first
# family make layer 2's pre-activations 15x smaller (std 0.065), which is why it flips so much more (measured with N(0,1) bags on those
# weights: r10000 (0, 12), r100000 (5, 82); the CPU oracle's own fp32 run: r10000 (0, 6)).
RANDN_CASES = ("r256", "r10000", "r100000")
MAX_FLIP_CASES_RANDN = 2


def _model(c, params, cuda):
    from toad_amd import TOAD_fc_mtl_concat
    m = TOAD_fc_mtl_concat(dropout=False, n_classes=c)
    m.load_state_dict(params, strict=True)
    m.relocate()
    return m


@pytest.mark.parametrize("name", CASES)
def test_module_matches_reference_golden(cuda, golden, name):
    """Same call sequence as the reference train loop (utils/core_utils_mtl_concat.py:201-231)."""
    ci = case_inputs(golden, name)
    model = _model(ci["c"], ci["params"], cuda)
    model.train()
    data, sex = ci["x"].to(cuda), ci["sex"].to(cuda)
    label, site = ci["label"].to(cuda), ci["site"].to(cuda)
    res = model(data, sex, return_features=True)
    loss_fn = torch.nn.CrossEntropyLoss()
    loss = loss_fn(res["logits"], label) * 0.75 + loss_fn(res["site_logits"], site) * 0.25
    loss.backward()
    out = {k: v.detach().cpu() for k, v in res.items()}
    grads = {k: p.grad for k, p in model.named_parameters()}
    assert set(grads) == set(orc.PARAM_KEYS)
    # Gradients vs the reference are only comparable element-wise when both sides took the same ReLU branches. The closed-form
    # golden bags put some pre-activations within round-off of zero; where this implementation lands on the other side of zero
    # than the exact forward (a LEGITIMATE flip: relu_flips asserts |pre-activation| is round-off there), a whole dZ row differs
    # and the gradient is compared with the fp64 backward on the device's own activations instead (same bound, identical masks).
    from toad_amd import functional as F_
    w = {s_: ci["params"][k].to(cuda) for s_, k in SLOT2KEY.items()}
    f1 = f2 = 0
    if ci["n"] > 0:
        outs, sv = F_.mil_forward(w, data, sex)
        assert torch.equal(outs["logits"], res["logits"].detach())          # the per-op route is bitwise the module's
        pos1, pos2 = relu_flip_positions(ci["params"], ci["x"], sv.h1, sv.h)
        f1, f2 = int(pos1.shape[0]), int(pos2.shape[0])
        # H1 / H themselves against the reference's activations (forward hooks on its ReLU modules) at the BASELINE sizes
        assert check_activations_vs_golden(golden, name, sv.h1, sv.h) == (name in ("n10000", "n100000", "r10000", "r100000"))
        # the four trunk gradients against the REFERENCE's fp64 matrices, flips or not: what the known flipped mask elements cannot
        # explain must match the reference (tests/helpers.py); the device-activation comparison below stays as the all-rows check
        rows = check_trunk_grads_vs_golden_blocks(golden, name, grads, ci["x"], pos1, pos2)
        assert (rows > 0) == (name in ("n256", "n777", "n777_c2", "n1024_sat", "n10000", "n100000") + RANDN_CASES), (name, rows)
    FLIPS[name] = (f1, f2)
    n_flips = f1 + f2
    # against the REFERENCE's golden gradients: all 14 when no mask flipped; otherwise every gradient a flip cannot reach (the ten
    # attention / head gradients always, layer 2's as well when only layer-1 masks flipped) - see tests/helpers.py MASK_FREE_KEYS
    golden_keys = None if n_flips == 0 else (MASK_FREE_KEYS + (LAYER2_KEYS if f2 == 0 else ()))
    check_outputs_vs_golden(golden, name, out, loss.item(), grads, atol=1e-4, grad_keys=golden_keys)
    if n_flips:
        # the gradients downstream of a flipped mask: fp64 backward on the device's own activations (identical masks), same bound
        dl, ds = orc.loss_grad(outs["logits"].cpu(), ci["label"], outs["site_logits"].cpu(), ci["site"])
        sv_cpu = orc.Saved(x=ci["x"], h1=sv.h1.cpu(), h=sv.h.cpu(), p=sv.p.cpu(), a_raw=sv.a_raw.cpu(), m=sv.m.cpu(), mcat=sv.mcat.cpu(), sex=ci["sex"])
        s64 = orc.Saved(**{k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in sv_cpu.__dict__.items()})
        og = orc.backward({k: v.double() for k, v in ci["params"].items()}, s64, dl.double(), ds.double())
        o32 = orc.backward(ci["params"], sv_cpu, dl, ds)
        for k in orc.PARAM_KEYS:
            if k in golden_keys:
                continue
            noise = (o32[k].double() - og[k]).abs().max().item()
            assert_grad_close(grads[k], og[k], 2e-5, grad_scale(og, k), what=f"{name}:{k} ({f1}+{f2} legit ReLU flips)", floor=10.0 * noise)
    a_only = model(data, sex, attention_only=True)
    assert a_only.shape == (ci["n"],)
    assert torch.equal(a_only, res["A"][0].detach())


def test_golden_gradient_comparison_rarely_leaves_the_golden_values():
    """How many of the golden cases above had a legitimate ReLU-mask flip, i.e. compared SOME gradients (never the ten mask-free
    ones) with the fp64 backward on the device's activations instead of the reference's captured values. Visible and bounded."""
    if len(FLIPS) < len(CASES):
        pytest.skip("runs after test_module_matches_reference_golden in the same process")
    flipped = {k: v for k, v in FLIPS.items() if sum(v)}
    print(f"golden cases with legitimate ReLU flips (layer 1, layer 2): {flipped} of {len(FLIPS)}")
    import json, os
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/golden_flip_cases.json", "w") as f:
            json.dump({"cases": len(FLIPS), "flipped": flipped, "max_allowed_closed_form": MAX_FLIP_CASES, "max_allowed_randn": MAX_FLIP_CASES_RANDN}, f)
    except OSError:
        pass
    assert sum(1 for k in flipped if k not in RANDN_CASES) <= MAX_FLIP_CASES, flipped
    assert sum(1 for k in flipped if k in RANDN_CASES) <= MAX_FLIP_CASES_RANDN and "r256" not in flipped and sum(flipped.get("r10000", (0, 0))) <= 4, flipped
    assert all(sum(v) <= 64 for v in flipped.values()), flipped     # a handful of boundary elements, not a systematic difference


@pytest.mark.parametrize("name", ["n777", "n1024_sat", "n10000", "n100000", "r100000"])
def test_backward_chain_tight_with_identical_relu_masks(cuda, golden, name):
    """Flip-free check of the whole backward chain: the oracle's hand-written backward is run on
    the activations the GPU forward saved (so ReLU masks are identical on both sides); what is left
    is rounding only, so the bound is 2e-5 of each gradient's OWN scale (|grad|max of the fp64 result; no absolute
    floor - |dWa|max is 5e-6) for all 14 gradients, including the headline size N = 100,000."""
    from toad_amd import functional as F_
    ci = case_inputs(golden, name)
    w = {s: ci["params"][k].to(cuda) for s, k in SLOT2KEY.items()}
    outs, sv = F_.mil_forward(w, ci["x"].to(cuda), ci["sex"].to(cuda))
    dl, ds = orc.loss_grad(outs["logits"].cpu(), ci["label"], outs["site_logits"].cpu(), ci["site"])
    g, _ = F_.mil_backward(w, sv, dl.to(cuda), ds.to(cuda))
    saved_cpu = orc.Saved(x=ci["x"], h1=sv.h1.cpu(), h=sv.h.cpu(), p=sv.p.cpu(), a_raw=sv.a_raw.cpu(),
                          m=sv.m.cpu(), mcat=sv.mcat.cpu(), sex=ci["sex"])
    s64 = orc.Saved(**{k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in saved_cpu.__dict__.items()})
    og = orc.backward({k: v.double() for k, v in ci["params"].items()}, s64, dl.double(), ds.double())
    # Three of the 14 gradients are sums that cancel almost completely (d attention_a/b biases = column sums of dP, d attention_c
    # bias = column sums of dS = 0 by the softmax's shift invariance): their fp32 round-off is set by the size of the cancelling
    # terms, not of the result. The allowance for that is measured, not assumed: 10 x the deviation of the SAME backward evaluated
    # in plain fp32 on the CPU from its fp64 value. For the well-conditioned gradients that deviation is ~1e-6 of the scale and the
    # 2e-5 bound is the one that binds.
    # The three cancellation-dominated gradients get 32 x since round 3: their device error is eps32 x the size of the cancelling terms
    # (e.g. d attention_c.bias = sum_r p_r (dM.H_r - dM.M): an error of 1e-7 relative in the softmax normaliser shifts it by 1e-7 |dM.M|),
    # and which way the last bits of that normaliser fall depends on the summation order of the merge kernel, which round 3 shortened.
    # Measured there: 18 x (attention_c.bias, n777) and 12 x (attention_a.bias, n777) the CPU's ONE noise sample; every other gradient
    # of every case stays at or below 3 x, and the eleven well-conditioned gradients keep 10 x (where the 2e-5 bound is the one that binds).
    cancelling = ("ba", "bb", "bc")
    o32 = orc.backward(ci["params"], saved_cpu, dl, ds)
    for sl, k in SLOT2KEY.items():
        noise = (o32[k].double() - og[k]).abs().max().item()
        assert_grad_close(g[sl], og[k], 2e-5, grad_scale(og, k), what=f"{name}:{sl}", floor=(32.0 if sl in cancelling else 10.0) * noise)


def test_module_random_bag_vs_oracle(cuda):
    torch.manual_seed(0)
    from toad_amd import TOAD_fc_mtl_concat
    model = TOAD_fc_mtl_concat(n_classes=18)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.relocate()
    x = torch.randn(5003, 1024); sex = torch.tensor([1.0]); label = torch.tensor([3]); site = torch.tensor([0])
    res = model(x.to(cuda), sex.to(cuda))
    loss = orc.loss_fn(res["logits"], label.to(cuda), res["site_logits"], site.to(cuda))
    loss.backward()
    o_out, o_loss, o_grads = orc.fwd_bwd(params, x, sex, label, site)
    for k in ("logits", "Y_prob", "site_logits", "site_prob", "A"):
        assert (res[k].detach().cpu() - o_out[k]).abs().max().item() <= 1e-4, k
    assert abs(loss.item() - o_loss.item()) <= 1e-4
    # yardstick for ReLU-boundary flips: the oracle's own fp32-vs-fp64 deviation on this bag (as in the goldens)
    p64 = {k: v.double() for k, v in params.items()}
    _, _, g64 = orc.fwd_bwd(p64, x.double(), sex.double(), label, site)
    for k, p in model.named_parameters():
        dev = (o_grads[k].double() - g64[k]).abs().max().item()
        assert_grad_close(p.grad, g64[k], 2e-5, grad_scale(g64, k), what=k, floor=4.0 * dev)


def test_fused_loss_path_equals_autograd_path(cuda):
    """functional.mil_forward/mil_backward + the fused CE kernel give the same gradients as
    module + torch CE + autograd, and accumulate (beta=1) into preset destinations."""
    from toad_amd import TOAD_fc_mtl_concat, functional as F_, ops
    torch.manual_seed(1)
    model = TOAD_fc_mtl_concat(n_classes=18); model.relocate()
    x = torch.randn(3001, 1024, device=cuda); sex = torch.ones(1, device=cuda)
    label = torch.tensor([5], device=cuda); site = torch.tensor([1], device=cuda)
    res = model(x, sex)
    ce = torch.nn.CrossEntropyLoss()
    (ce(res["logits"], label) * 0.75 + ce(res["site_logits"], site) * 0.25).backward()
    w = {k: v.detach() for k, v in model._weights().items()}
    outs, saved = F_.mil_forward(w, x, sex)
    lossv, dl, ds = ops.mtl_ce_fwd_bwd(outs["logits"], outs["site_logits"], label, site)
    g, _ = F_.mil_backward(w, saved, dl, ds)
    sp = model._slot_params()
    ref = {k: sp[k].grad for k in F_.SLOTS}
    for k in F_.SLOTS:      # same kernels; only d(loss)/d(logits) comes from a different CE implementation
        assert_grad_close(g[k], ref[k], 1e-5, grad_scale(ref, k), what=k)
    dest = {k: torch.ones_like(sp[k]) for k in F_.SLOTS}
    g2, _ = F_.mil_backward(w, saved, dl, ds, grads=dest, beta=1.0)
    for k in F_.SLOTS:
        assert (dest[k] - (g[k] + 1.0)).abs().max().item() <= 1e-5 * max(g[k].abs().max().item(), 1.0), k


def test_full_size_properties_100k(cuda):
    """BASELINE size (100k x 1024): properties that do not need a CPU oracle run.
    (1) permutation invariance of the bag; (2) pooled features are a convex combination of rows;
    (3) run-to-run bitwise determinism of forward and gradients."""
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(2)
    model = TOAD_fc_mtl_concat(n_classes=18); model.relocate()
    n = 100000
    g = torch.Generator(device=cuda).manual_seed(1000)
    x = torch.randn(n, 1024, device=cuda, generator=g)
    sex = torch.zeros(1, device=cuda); label = torch.tensor([1], device=cuda); site = torch.tensor([0], device=cuda)

    def run(xx):
        model.zero_grad(set_to_none=True)
        r = model(xx, sex, return_features=True)
        loss = orc.loss_fn(r["logits"], label, r["site_logits"], site)
        loss.backward()
        return r, torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()

    r1, g1 = run(x)
    r2, g2 = run(x)
    assert torch.equal(r1["logits"], r2["logits"]) and torch.equal(g1, g2)
    perm = torch.randperm(n, device=cuda, generator=g)
    r3, g3 = run(x[perm].contiguous())
    assert (r3["logits"] - r1["logits"]).abs().max().item() <= 1e-4
    assert (r3["A"][:, torch.argsort(perm)] - r1["A"]).abs().max().item() <= 1e-5
    off = 0
    names = [k for k, _ in model.named_parameters()]
    gd1, gd3 = {}, {}
    for k, p in model.named_parameters():
        gd1[k], gd3[k] = g1[off:off + p.numel()], g3[off:off + p.numel()]
        off += p.numel()
    for k in names:         # summation order over 100,000 rows changes: 5e-4 of each gradient's own scale
        assert_grad_close(gd3[k], gd1[k], 5e-4, grad_scale(gd1, k), what=f"perm:{k}")
    feats = r1["features"][:, :512]
    assert feats.min().item() >= 0.0                                 # H = relu(...) >= 0, weights >= 0
    a = torch.softmax(r1["A"].double(), dim=1)
    assert (a.sum(1) - 1).abs().max().item() < 1e-9


def test_million_patch_bag_duplication_property(cuda):
    """Maximum size: a 1.1 M-patch bag (4.5 GB, M*K*4 >= 2^32: beyond the 32-bit row offsets of the persistent NT GEMMs, which therefore run over
    ROW CHUNKS of 1,047,552 rows - csrc/step.hip nt_rows; until round 5 the 64-bit generic kernels served it). Property: repeating every row r
    times leaves the softmax-pooled features, the logits and - because each copy carries 1/r of the attention - all gradients unchanged.
    Then the same bag in TRAIN mode with dropout: chunk j > 0 draws its trunk masks from the stream seed + j * kChunkSeedStep at the chunk-local
    element index (nothing else reproduces that, so it is pinned here): the second chunk's H1 / H must be zero exactly where the exported masks of
    those streams are zero, and the fused step (toad_mil_step_f32) must give the gradients of forward + loss + backward as separate calls."""
    from toad_amd import TOAD_fc_mtl_concat
    params = orc.xavier_params(18, seed=2)
    m = TOAD_fc_mtl_concat(n_classes=18); m.load_state_dict(params); m.relocate(); m.train()
    small = torch.randn(1100, 1024, generator=torch.Generator().manual_seed(3)).to(cuda)
    sex, label, site = torch.ones(1, device=cuda), torch.tensor([3], device=cuda), torch.tensor([1], device=cuda)
    ce = torch.nn.CrossEntropyLoss()

    def run(bag):
        m.zero_grad()
        out = m(bag, sex)
        (ce(out["logits"], label) * 0.75 + ce(out["site_logits"], site) * 0.25).backward()
        return out, {k: p.grad.clone() for k, p in m.named_parameters()}

    out_s, g_s = run(small)
    out_b, g_b = run(small.repeat(1000, 1))
    assert out_b["A"].shape == (2, 1_100_000)
    assert (out_b["logits"] - out_s["logits"]).abs().max().item() <= 1e-5
    assert (out_b["A"][:, :1100] - out_s["A"]).abs().max().item() <= 1e-4
    for k in g_s:       # 1e-4 absolute: the two sizes run on different kernels, so ReLU-boundary flips differ
        assert (g_b[k] - g_s[k]).abs().max().item() <= 1e-4, k
    del out_b, g_b
    # ---- train-mode dropout above the chunk size
    from toad_amd import functional as F_, ops
    big = small.repeat(1000, 1)
    n, chunk, step_c, M64 = big.shape[0], 4092 * 256, 0xD1B54A32D192ED03, 0xFFFFFFFFFFFFFFFF          # csrc/step.hip kChunkRows, kChunkSeedStep
    w = {k: v.detach() for k, v in m._weights().items()}
    drop, seed = 0.25, 987654321
    arena = ops.mil_fwd(w, big, sex, drop, seed)
    s1, s2, _, _ = F_.drop_seeds(seed)
    rows1 = n - chunk
    for name, sd in (("h1", s1), ("h", s2)):
        act = arena.view(name, (n, 512))
        mk1 = ops.dropout_mask(rows1 * 512, drop, (sd + step_c) & M64, cuda).reshape(rows1, 512)      # chunk 1: its own stream, chunk-local index
        c1 = act[chunk:]
        assert (c1[mk1 == 0] == 0).all(), name
        kept = (c1 != 0) & (mk1 > 0)
        assert 0.2 < kept.float().mean().item() < 0.6, name                                          # ~ 0.75 x P(relu > 0)
        mk0 = ops.dropout_mask(4096 * 512, drop, sd, cuda).reshape(4096, 512)                         # chunk 0: the bag-wide stream
        assert (act[:4096][mk0 == 0] == 0).all(), name
        # chunk 1 does NOT continue chunk 0's stream (that would be index (chunk + r) * 512 + c of the stream `sd`): its zero pattern differs from it
        assert not torch.equal(mk1[:4096] == 0, mk0 == 0), name
        del act, mk1, c1, kept, mk0
    outs_logits = arena.view("logits", (1, 18)).clone(); outs_slog = arena.view("site_logits", (1, 2)).clone()
    lossv, dl, ds = ops.mtl_ce_fwd_bwd(outs_logits, outs_slog, label, site, 0.75, 0.25)
    g_sep = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    ops.mil_bwd(w, g_sep, 0.0, big, arena, dl, ds, None, None, drop, seed)
    del arena
    g_fused = {k: torch.zeros_like(w[k]) for k in ops.STEP_SLOTS}
    loss2, _, _ = ops.mil_step(w, g_fused, 0.0, big, sex, label, site, 0.75, 0.25, drop, seed)
    assert torch.equal(loss2, lossv)
    for k in ops.STEP_SLOTS:
        assert torch.equal(g_fused[k], g_sep[k]), k


def test_non_default_stream_and_autograd_thread(cuda):
    """The C ABI takes the stream explicitly: the whole path on a side stream (forward on the caller's thread, backward on
    PyTorch's autograd worker thread) gives bitwise the results of the default stream, and two models on two streams do not
    disturb each other (no hidden global state in the library)."""
    from toad_amd import TOAD_fc_mtl_concat
    params = orc.xavier_params(18, seed=4)
    x = torch.randn(3000, 1024, generator=torch.Generator().manual_seed(9)).to(cuda)
    sex, label, site = torch.zeros(1, device=cuda), torch.tensor([7], device=cuda), torch.tensor([0], device=cuda)
    ce = torch.nn.CrossEntropyLoss()

    def run(model):
        model.zero_grad()
        out = model(x, sex)
        (ce(out["logits"], label) * 0.75 + ce(out["site_logits"], site) * 0.25).backward()
        return out["logits"].detach().clone(), [p.grad.clone() for p in model.parameters()]

    m1 = TOAD_fc_mtl_concat(n_classes=18); m1.load_state_dict(params); m1.relocate(); m1.train()
    m2 = TOAD_fc_mtl_concat(n_classes=18); m2.load_state_dict(params); m2.relocate(); m2.train()
    ref_logits, ref_grads = run(m1)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):                                  # interleave the two streams
        with torch.cuda.stream(s1):
            l1, g1 = run(m1)
        with torch.cuda.stream(s2):
            l2, g2 = run(m2)
    torch.cuda.synchronize()
    for lg, gs in ((l1, g1), (l2, g2)):
        assert torch.equal(lg, ref_logits)
        assert all(torch.equal(a, b) for a, b in zip(gs, ref_grads))


@pytest.mark.parametrize("shape", [None, (512, 384, 2), (200, 100, 3), (768, 128, 4), (640, 512, 1), (1024, 256, 2),
                                   (2048, 640, 5), (100, 30, 7), (1536, 1100, 1), (36, 516, 9)])
def test_attn_net_gated_standalone(cuda, shape):
    """Attn_Net_Gated with its constructor defaults (L=1024, D=256, n_tasks=1; model_toad.py:19) and with other (L, D, n_tasks) the
    constructor accepts: forward scores and all gradients (parameters and input) against autograd on the oracle's formula. The last four
    shapes lie OUTSIDE the pool kernels' covering instantiation (L > 1024 or not a multiple of 8, D > 512 or not a multiple of 4, n_tasks > 4:
    round 4 raised NotImplementedError there; the reference takes any) and run over column / task blocks with zero padding."""
    from toad_amd import Attn_Net_Gated
    torch.manual_seed(4)
    l, d, t = shape or (1024, 256, 1)
    net = (Attn_Net_Gated() if shape is None else Attn_Net_Gated(L=l, D=d, n_tasks=t)).to(cuda)
    x = torch.randn(999, l)
    xg = x.to(cuda).requires_grad_(True)
    a, xo = net(xg)
    assert xo is xg and a.shape == (999, t)
    wa, ba = net.attention_a[0].weight.detach().cpu(), net.attention_a[0].bias.detach().cpu()
    wb, bb = net.attention_b[0].weight.detach().cpu(), net.attention_b[0].bias.detach().cpu()
    wc, bc = net.attention_c.weight.detach().cpu(), net.attention_c.bias.detach().cpu()
    xr = x.clone().requires_grad_(True)
    prm = [t.clone().requires_grad_(True) for t in (wa, ba, wb, bb, wc, bc)]
    ref = orc.gated_scores(torch.addmm(prm[1], xr, prm[0].t()), torch.addmm(prm[3], xr, prm[2].t()), prm[4], prm[5])
    assert (a.detach().cpu() - ref.detach()).abs().max().item() <= 1e-5
    a.sin().sum().backward(); ref.sin().sum().backward()
    mine = [net.attention_a[0].weight, net.attention_a[0].bias, net.attention_b[0].weight, net.attention_b[0].bias,
            net.attention_c.weight, net.attention_c.bias]
    for p, r in zip(mine, prm):        # smooth (no ReLU): rounding only, 1e-4 of each gradient's own scale
        assert_grad_close(p.grad, r.grad, 1e-4, float(r.grad.abs().max()))
    assert_grad_close(xg.grad, xr.grad, 1e-4, float(xr.grad.abs().max()))


def test_attn_net_gated_dropout_column_blocks_draw_independent_masks(cuda):
    """Standalone Attn_Net_Gated(dropout=True) with D > 512 (models/model_toad.py:19-41 take any D): the gate columns run as blocks of <= 512, each
    block and each BRANCH (tanh / sigmoid) on a mask stream of its own - block j uses (sa + 2 j G, sb + 2 j G), sb = sa + G. (Round 5 stepped the
    blocks by G: block j + 1's tanh masks were block j's sigmoid masks.) The device masks are exported per stream (toad_dropout_mask_f32, element
    index row * block_width + column), shown to be pairwise different, and fed to autograd on the reference formula: forward scores and every
    gradient must agree."""
    from toad_amd import Attn_Net_Gated, functional as F_, ops
    l, d, t, n = 512, 1100, 2, 700
    torch.manual_seed(11)
    net = Attn_Net_Gated(L=l, D=d, dropout=True, n_tasks=t).to(cuda)
    net.train()
    x = torch.randn(n, l)
    xg = x.to(cuda).requires_grad_(True)
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # what _draw_dropout will draw next
    torch.manual_seed(77)
    a, _ = net(xg)
    _, _, sa, sb = F_.drop_seeds(seed)
    G, M64 = 0x9E3779B97F4A7C15, 0xFFFFFFFFFFFFFFFF
    blocks = [(0, 512), (512, 1024), (1024, 1100)]
    ma, mb, streams = torch.empty(n, d), torch.empty(n, d), []
    for j, (d0, d1) in enumerate(blocks):
        wd = d1 - d0
        ka = ops.dropout_mask(n * wd, F_.DROP_P, (sa + 2 * j * G) & M64, cuda).reshape(n, wd).cpu()
        kb = ops.dropout_mask(n * wd, F_.DROP_P, (sb + 2 * j * G) & M64, cuda).reshape(n, wd).cpu()
        ma[:, d0:d1], mb[:, d0:d1] = ka, kb
        streams += [ka, kb]
    for i in range(4):                                           # the four 512-wide streams (two blocks x two branches) are pairwise different
        for k in range(i + 1, 4):
            assert (streams[i] != streams[k]).float().mean().item() > 0.2, (i, k)
    prm = [p_.detach().cpu().clone().requires_grad_(True) for p_ in (net.attention_a[0].weight, net.attention_a[0].bias, net.attention_b[0].weight,
                                                                     net.attention_b[0].bias, net.attention_c.weight, net.attention_c.bias)]
    xr = x.clone().requires_grad_(True)
    ga = torch.tanh(torch.addmm(prm[1], xr, prm[0].t())) * ma      # nn.Dropout after Tanh / Sigmoid (model_toad.py:27-29): mask = 0 or 1 / (1 - p)
    gb = torch.sigmoid(torch.addmm(prm[3], xr, prm[2].t())) * mb
    ref = torch.addmm(prm[5], ga * gb, prm[4].t())
    assert (a.detach().cpu() - ref.detach()).abs().max().item() <= 2e-5
    a.sin().sum().backward(); ref.sin().sum().backward()
    mine = [net.attention_a[0].weight, net.attention_a[0].bias, net.attention_b[0].weight, net.attention_b[0].bias, net.attention_c.weight, net.attention_c.bias]
    for p_, r in zip(mine, prm):
        assert_grad_close(p_.grad, r.grad, 1e-4, float(r.grad.abs().max()))
    assert_grad_close(xg.grad, xr.grad, 1e-4, float(xr.grad.abs().max()))


def test_dp_step_single_gpu_matches_mean_of_oracle_gradients(cuda):
    """SlideShardedDP on one GPU: two slides, gradient bucket == mean of per-slide oracle gradients
    (kernels accumulate with beta, the 1/global scale is folded into the CE weights), and the flat
    single-tensor SGD step moves every parameter by -lr * that mean."""
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.dp import SlideShardedDP
    torch.manual_seed(5)
    model = TOAD_fc_mtl_concat(n_classes=18)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.relocate()
    dp = SlideShardedDP(model, lambda ps: torch.optim.SGD(ps, lr=0.5))
    slides_cpu = []
    for i, n in enumerate((1500, 0, 777)):          # the middle slide is an EMPTY bag (heads-only gradient)
        g = torch.Generator().manual_seed(50 + i)
        slides_cpu.append((torch.randn(n, 1024, generator=g), torch.tensor([float(i % 2)]), torch.tensor([3 + i]),
                           torch.tensor([i % 2])))
    slides = [tuple(t.to(cuda) for t in s) for s in slides_cpu]
    dp.flat_grad.fill_(123.0)                       # stale contents must be overwritten, not accumulated
    losses = dp.step(slides, global_slides=3)
    mean = {k: torch.zeros_like(v) for k, v in params.items()}
    mean64 = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in params.items()}
    p64 = {k: v.double() for k, v in params.items()}
    for s in slides_cpu:
        _, _, g = orc.fwd_bwd(params, *s)
        _, _, g64 = orc.fwd_bwd(p64, s[0].double(), s[1].double(), s[2], s[3])
        for k in mean:
            mean[k] += g[k] / 3
            mean64[k] += g64[k] / 3
    new = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for k, p in model.named_parameters():
        ref = mean64[k]
        # relative to each gradient's own scale; the allowance for ReLU-boundary flips is 4x the oracle's own fp32-vs-fp64 deviation
        dev = (mean[k].double() - ref).abs().max().item()
        assert_grad_close(p.grad, ref, 2e-5, grad_scale(mean64, k), what=k, floor=4.0 * dev)
        assert (new[k].double() - (params[k].double() - 0.5 * p.grad.cpu().double())).abs().max().item() <= 1e-6, k   # SGD moved by -lr * bucket
    assert len(losses) == 3 and losses[0].shape == (3,)


def _masks_from_seed(cuda, seed, n):
    from toad_amd import functional as F_, ops
    s1, s2, sa, sb = F_.drop_seeds(seed)
    mk = {}
    for name, sd, w in (("h1", s1, 512), ("h", s2, 512), ("a", sa, 384), ("b", sb, 384)):
        mk[name] = ops.dropout_mask(n * w, F_.DROP_P, sd, cuda).reshape(n, w).cpu()
    return mk


@pytest.mark.parametrize("n", [300, 5000])
def test_train_mode_dropout_matches_oracle_with_the_same_masks(cuda, n):
    """dropout=True + train(): the in-kernel masks (hash of seed and element index, never stored) are
    reproduced with toad_dropout_mask_f32 and fed to the oracle; forward and every gradient must agree.
    Also: masks are ~25 % zeros scaled 1/0.75, differ between the four sites, and eval() is mask-free."""
    from toad_amd import TOAD_fc_mtl_concat, functional as F_
    torch.manual_seed(11)
    model = TOAD_fc_mtl_concat(dropout=True, n_classes=18)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    sd = model.state_dict()
    # reference key names with dropout=True: attention_net.{0,3,6.*}; map onto the oracle's (dropout=False) names
    rename = {"attention_net.3.": "attention_net.2.", "attention_net.6.": "attention_net.4."}
    params = {}
    for k, v in sd.items():
        for a, b in rename.items():
            if k.startswith(a):
                k = b + k[len(a):]
        params[k] = v.detach().clone()
    assert set(params) == set(orc.PARAM_KEYS)
    model.relocate(); model.train()
    x = torch.randn(n, 1024); sex = torch.tensor([1.0]); label = torch.tensor([6]); site = torch.tensor([1])
    torch.manual_seed(99)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # what the module will draw next
    torch.manual_seed(99)
    res = model(x.to(cuda), sex.to(cuda))
    loss = orc.loss_fn(res["logits"], label.to(cuda), res["site_logits"], site.to(cuda))
    loss.backward()
    mk = _masks_from_seed(cuda, seed, n)
    for name, m in mk.items():
        vals = set(m.unique().tolist())
        assert vals <= {0.0, float(torch.tensor(1.0) / 0.75)} and len(vals) == 2, name
        assert abs((m == 0).float().mean().item() - 0.25) < 0.01, name
    assert not torch.equal(mk["h1"], mk["h"]) and not torch.equal(mk["a"], mk["b"])
    # flip-free comparison: oracle backward on the GPU-saved activations with the same masks
    w = {k: v.detach() for k, v in model._weights().items()}
    outs, sv = F_.mil_forward(w, x.to(cuda), sex.to(cuda), F_.DROP_P, seed)
    assert torch.equal(outs["logits"], res["logits"].detach())  # same seed -> same masks -> bitwise same forward
    o_out, _ = orc.forward(params, x, sex, masks=mk)
    for k in ("logits", "site_logits", "A"):
        assert (res[k].detach().cpu() - o_out[k]).abs().max().item() <= 1e-4, k
    dl, ds = orc.loss_grad(outs["logits"].cpu(), label, outs["site_logits"].cpu(), site)
    og = orc.backward(params, orc.Saved(x=x, h1=sv.h1.cpu(), h=sv.h.cpu(), p=sv.p.cpu(), a_raw=sv.a_raw.cpu(),
                                        m=sv.m.cpu(), mcat=sv.mcat.cpu(), sex=sex, masks=mk), dl, ds)
    inv = {}
    for k in sd:
        kk = k
        for a, b in rename.items():
            if k.startswith(a):
                kk = b + k[len(a):]
        inv[kk] = k
    grads = dict(model.named_parameters())
    for k in orc.PARAM_KEYS:
        ref = og[k]
        got = grads[inv[k]].grad.cpu()
        assert_grad_close(got, ref, 1e-4, grad_scale(og, k), what=k)     # fp32 oracle backward on identical masks: rounding only
    # a second forward draws a new seed -> different masks; eval() -> deterministic, equals the no-dropout oracle
    res2 = model(x.to(cuda), sex.to(cuda))
    assert not torch.equal(res2["logits"], res["logits"])
    model.eval()
    with torch.no_grad():
        e1 = model(x.to(cuda), sex.to(cuda)); e2 = model(x.to(cuda), sex.to(cuda))
    assert torch.equal(e1["logits"], e2["logits"])
    o_eval, _ = orc.forward(params, x, sex)
    assert (e1["logits"].cpu() - o_eval["logits"]).abs().max().item() <= 1e-4


@pytest.mark.parametrize("n,drop", [(1, 0.0), (777, 0.0), (4000, 0.25)])
def test_fused_step_entry_is_bitwise_the_per_op_path(cuda, n, drop):
    """toad_mil_step_f32 (one C call) == mil_forward + mtl_ce + mil_backward (per-op calls): same kernels, same
    order -> bitwise-equal loss and gradients, with and without dropout, and beta-accumulation works."""
    from toad_amd import TOAD_fc_mtl_concat, functional as F_, ops
    torch.manual_seed(21)
    model = TOAD_fc_mtl_concat(n_classes=18); model.relocate()
    w = {k: v.detach() for k, v in model._weights().items()}
    x = torch.randn(n, 1024, device=cuda); sex = torch.ones(1, device=cuda)
    label = torch.tensor([9], device=cuda); site = torch.tensor([0], device=cuda)
    seed = 123456789
    outs, saved = F_.mil_forward(w, x, sex, drop, seed)
    lossv, dl, ds = ops.mtl_ce_fwd_bwd(outs["logits"], outs["site_logits"], label, site, 0.75, 0.25)
    g, _ = F_.mil_backward(w, saved, dl, ds)
    dest = {k: torch.full_like(w[k], 2.0) for k in ops.STEP_SLOTS}
    loss2, lg, slg = ops.mil_step(w, dest, 0.0, x, sex, label, site, 0.75, 0.25, drop, seed, want_logits=True)
    assert torch.equal(loss2, lossv) and torch.equal(lg, outs["logits"]) and torch.equal(slg, outs["site_logits"])
    d = w["wa"].shape[0]
    ref = dict(g); ref["wab"] = torch.cat([g["wa"], g["wb"]], 0); ref["bab"] = torch.cat([g["ba"], g["bb"]], 0)
    for k in ops.STEP_SLOTS:
        assert_step_grad_matches_per_op(dest[k], ref[k], k, n)
    ops.mil_step(w, dest, 1.0, x, sex, label, site, 0.75, 0.25, drop, seed)       # accumulate on top
    for k in ops.STEP_SLOTS:
        assert (dest[k] - 2 * ref[k]).abs().max().item() <= 1e-6 * max(ref[k].abs().max().item(), 1.0), k


def test_flat_adam_matches_torch_adam(cuda):
    """toad_adam_step_f32 == torch.optim.Adam(lr, weight_decay) (the reference's get_optim, utils/utils.py:63-70)."""
    from toad_amd.optim import FlatAdam
    torch.manual_seed(3)
    n = 1192768
    p0 = torch.randn(n); grads = [torch.randn(n) * 0.01 for _ in range(5)]
    ref_p = torch.nn.Parameter(p0.clone())
    ref = torch.optim.Adam([ref_p], lr=1e-3, weight_decay=1e-5)
    mine_p = p0.clone().to(cuda)
    mine = FlatAdam(mine_p, lr=1e-3, weight_decay=1e-5)
    for g in grads:
        ref_p.grad = g.clone(); ref.step()
        mine.step(g.to(cuda))
    assert (mine_p.cpu() - ref_p.detach()).abs().max().item() <= 2e-6


def test_size_arg_small_end_to_end(cuda):
    """size_arg="small" ([1024, 512, 256], models/model_toad.py:56): forward, loss, backward vs the oracle."""
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(12)
    model = TOAD_fc_mtl_concat(size_arg="small", n_classes=5)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert params["attention_net.4.attention_a.0.weight"].shape == (256, 512)
    model.relocate()
    x = torch.randn(2111, 1024); sex = torch.tensor([0.0]); label = torch.tensor([4]); site = torch.tensor([1])
    res = model(x.to(cuda), sex.to(cuda))
    loss = orc.loss_fn(res["logits"], label.to(cuda), res["site_logits"], site.to(cuda))
    loss.backward()
    o_out, o_loss, o_grads = orc.fwd_bwd(params, x, sex, label, site)
    for k in ("logits", "site_logits", "A"):
        assert (res[k].detach().cpu() - o_out[k]).abs().max().item() <= 1e-4, k
    p64 = {k: v.double() for k, v in params.items()}
    _, _, g64 = orc.fwd_bwd(p64, x.double(), sex.double(), label, site)
    for k, p in model.named_parameters():
        dev = (o_grads[k].double() - g64[k]).abs().max().item()
        assert_grad_close(p.grad, g64[k], 2e-5, grad_scale(g64, k), what=k, floor=4.0 * dev)


# ---- ragged multi-slide forward (validate / summary) ------------------------------------------------------------------------

def _ragged_slides(lens, c, seed=4321):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, n in enumerate(lens):
        out.append((torch.randn(n, 1024, generator=g).cuda(), torch.tensor([(5 * i) % c]).cuda(), torch.tensor([i % 2]).cuda(),
                    torch.tensor([float((i // 2) % 2)]).cuda()))
    return out


def test_forward_many_equals_per_slide_forward(cuda):
    """One pass of the trunk GEMMs over the concatenated ragged batch == model(data, sex) slide by slide (the reference's eval
    loop, eval_utils_mtl_concat.py:88-91). Operand scales differ (256-row blocks of the concatenation), so fp32 round-off:
    2e-5 of each tensor's own magnitude; the discrete outputs must agree exactly."""
    from toad_amd import TOAD_fc_mtl_concat
    torch.manual_seed(7)
    c = 18
    model = TOAD_fc_mtl_concat(n_classes=c); model.relocate(); model.eval()
    lens = [1, 255, 256, 257, 700, 3, 2049, 64]
    slides = _ragged_slides(lens, c)
    many = model.forward_many([s[0] for s in slides], [s[3] for s in slides], return_features=True)
    assert len(many) == len(lens)
    for s, r in zip(slides, many):
        with torch.no_grad():
            one = model(s[0], s[3], return_features=True)
        assert set(r) == set(one)
        for k in ("logits", "Y_prob", "site_logits", "site_prob", "features", "A"):
            assert r[k].shape == one[k].shape, k
            tol = 2e-5 * max(one[k].abs().max().item(), 1e-6)
            assert (r[k] - one[k]).abs().max().item() <= tol, (k, s[0].shape[0])
        assert torch.equal(r["Y_hat"], one["Y_hat"]) and torch.equal(r["site_hat"], one["site_hat"])
    with pytest.raises(ValueError):
        model.forward_many([slides[0][0]], [])
    with pytest.raises(ValueError):
        model.forward_many([torch.empty(0, 1024).cuda()], [slides[0][3]])
    assert model.forward_many([], []) == []


def test_summary_and_validate_grouped_equal_ungrouped(cuda):
    """summary / validate with the ragged grouping on (default) and off (the reference's one call per slide) tabulate the same
    numbers, in loader order, including a slide larger than the group size (forwarded alone) and groups cut at the row budget."""
    from types import SimpleNamespace
    from toad_amd import TOAD_fc_mtl_concat
    from toad_amd.eval import summary, forward_grouped
    from toad_amd.train import validate
    torch.manual_seed(8)
    c = 5
    model = TOAD_fc_mtl_concat(n_classes=c); model.relocate(); model.eval()
    lens = [300, 5000, 40, 40, 1200, 7, 900, 2600]
    slides = _ragged_slides(lens, c, seed=99)
    loader = [(s[0], s[1], s[2], s[3]) for s in slides]
    ids = ["s%d" % i for i in range(len(lens))]
    args = SimpleNamespace(n_classes=c, micro_average=False)
    a = summary(model, loader, args, slide_ids=ids, group_rows=0)
    b = summary(model, loader, args, slide_ids=ids, group_rows=4096)
    assert list(a["df"]["slide_id"]) == list(b["df"]["slide_id"]) == ids
    for col in a["df"].columns:
        if col == "slide_id":
            continue
        x, y = a["df"][col].to_numpy(dtype=float), b["df"][col].to_numpy(dtype=float)
        assert abs(x - y).max() <= 2e-5, col
    assert a["cls_test_error"] == b["cls_test_error"] and a["site_test_error"] == b["site_test_error"]
    va, vb = validate(model, loader, c, group_rows=0), validate(model, loader, c, group_rows=4096)
    assert va["slides"] == vb["slides"] == len(lens)
    for k in ("cls_loss", "site_loss", "cls_error", "site_error"):
        assert abs(va[k] - vb[k]) <= 2e-5 * max(abs(va[k]), 1.0), k
    assert abs(va["prob"] - vb["prob"]).max() <= 2e-5
    # grouping plan: [300] | [5000 alone] | [40, 40, 1200, 7, 900] | [2600]
    calls = []

    class Spy:
        def __call__(self, data, sex):
            calls.append([int(data.shape[0])]); return {}

        def forward_many(self, bags, sexes):
            calls.append([int(b.shape[0]) for b in bags]); return [{} for _ in bags]

    got = [int(bt[0].shape[0]) for bt, _ in forward_grouped(Spy(), loader, 4096)]
    assert got == lens
    assert calls == [[300], [5000], [40, 40, 1200, 7, 900], [2600]]
