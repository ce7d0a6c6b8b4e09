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
    kind = {0: "wave", 1: "equal", 2: "randn"}[int(round(float(equal)))]        # meta[6]: bag family (oracle/pin_against_reference.py KIND_CODE)
    x = orc.random_bag(n, 1000 + n) if kind == "randn" else orc.closed_form_bag(n, 1024, float(scale), kind)
    params = orc.random_params(c, 2000 + n) if kind == "randn" else orc.closed_form_params(c)
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


_ORACLE_DEV = {}


def oracle_fp32_noise(golden, name):
    """{key: max |oracle gradient in fp32 - oracle gradient in fp64|} for a golden case (cached).
    Several gradients on this path are sums that cancel almost completely (column sums of dS are zero by the softmax's shift
    invariance, so d attention biases / d attention_c.bias are round-off of terms ~1e4 x larger than the result; with all rows
    equal every attention gradient is exactly zero). For those the honest yardstick is the fp32 round-off of the computation
    itself, measured here on the CPU restatement, not the size of the (vanishing) result."""
    if name not in _ORACLE_DEV:
        ci = case_inputs(golden, name)
        _, _, g32 = orc.fwd_bwd(ci["params"], ci["x"], ci["sex"], ci["label"], ci["site"])
        p64 = {k: v.double() for k, v in ci["params"].items()}
        _, _, g64 = orc.fwd_bwd(p64, ci["x"].double(), ci["sex"].double(), ci["label"], ci["site"])
        _ORACLE_DEV[name] = {k: float((g32[k].double() - g64[k]).abs().max()) for k in orc.PARAM_KEYS}
    return _ORACLE_DEV[name]


def check_outputs_vs_golden(golden, name, out, loss, grads, atol=1e-4, grad_keys=None):
    """out: dict of CPU tensors (reference keys + 'features'); grads: {state-dict key: tensor}; grad_keys: the subset of
    gradients to hold against the golden values (default: all 14)."""
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
        # Gradients are pinned to the REFERENCE run in fp64 (grad_sample64). Tolerance is RELATIVE to each gradient's own
        # scale (golden grad_absmax): 2e-5 of it, or 10x the fp32 noise of this very computation where that is larger - the
        # reference's own fp32-vs-fp64 deviation (grad_dev64, full-tensor max; ReLU masks flip when a pre-activation lies within
        # fp32 roundoff of zero, so two correct fp32 implementations differ by whole dZ rows, DESIGN.md "Parity") or the CPU
        # restatement's (oracle_fp32_noise: the cancellation-dominated gradients). Nothing here is an absolute 1e-4 any more:
        # |grad|max is 5e-6 for Wa/Wb, so an absolute bound would be vacuous. The flip-free test in test_gpu_model.py holds
        # every gradient to 2e-5 of its scale without any noise allowance (also at N = 100,000).
        gmax_all = max(float(golden[pre + "grad_absmax/" + k]) for k in orc.PARAM_KEYS)
        onoise = oracle_fp32_noise(golden, name)
        for k in (orc.PARAM_KEYS if grad_keys is None else grad_keys):
            g = grads[k].detach().cpu()
            dev = max(float(golden[pre + "grad_dev64/" + k]), onoise[k])
            scale = float(golden[pre + "grad_absmax/" + k])
            tol = max(2e-5 * scale, 10.0 * dev)     # other fp32 implementations (other summation orders) differ by small multiples of the noise
            if scale == 0.0:        # exactly zero in the reference (one-patch bag: softmax of a single score has no gradient):
                tol = 1e-6 * gmax_all       # what an implementation returns is round-off of the cancelling terms
            assert_close(strided_sample(g), golden[pre + "grad_sample64/" + k], tol, what=f"{name}:grad64:{k}")
            assert_close(strided_sample(g), golden[pre + "grad_sample/" + k], tol + dev, what=f"{name}:grad32:{k}")
            l2 = float(golden[pre + "grad_l2_64/" + k])
            assert abs(float(g.double().norm()) - l2) <= 1e-3 * l2 + tol * g.numel() ** 0.5 + 1e-30, (name, k, float(g.double().norm()), l2)


# gradients whose exact value is zero by symmetry: the softmax is shift invariant, so d loss / d attention_c.bias == 0 and
# what any fp32 implementation returns is round-off of a sum whose terms have the scale of d attention_c.weight
_SCALE_PARTNER = {"attention_net.4.attention_c.bias": "attention_net.4.attention_c.weight", "bc": "wc"}


def grad_scale(ref: dict, key: str) -> float:
    """|gradient|max of `key` in the reference dict `ref` (its own scale), never an absolute floor."""
    s = float(ref[key].abs().max())
    if key in _SCALE_PARTNER and _SCALE_PARTNER[key] in ref:
        s = max(s, float(ref[_SCALE_PARTNER[key]].abs().max()))
    return s


def assert_grad_close(got, ref, rel, scale, what="", floor=0.0):
    err = float((got.detach().cpu().double() - ref.detach().cpu().double()).abs().max())
    tol = rel * scale + floor
    assert err <= tol, f"{what}: max err {err:.3e} > {tol:.3e} (scale {scale:.3e}, rel {err / max(scale, 1e-300):.2e})"


def assert_grad_close_or_few_flips(got, ref, rel, scale, what="", floor=0.0, max_flips=3, flip_size=1e-2):
    """assert_grad_close for a gradient a ReLU mask reaches (dW1, db1, dW2, db2), tolerant of LEGITIMATE mask flips. Two correct fp32
    implementations (or two summation orders of one) can land on different sides of zero at a pre-activation within round-off of it;
    the one dZ element that flips then moves the weight gradient by that patch's contribution: a RANK-ONE matrix (dz x^T; a layer-2
    flip reaches dW1 through one row of dZ1, again rank one). So: either the error meets the tight bound, or it does after at most
    `max_flips` rank-one terms are removed from it, each no larger than `flip_size` of the gradient's scale (one patch among many).
    Bias gradients (vectors) get the flip allowance directly."""
    g, r = got.detach().cpu().double(), ref.detach().cpu().double()
    e = g - r
    tol = rel * scale + floor
    if float(e.abs().max()) <= tol:
        return 0
    if e.dim() == 2:
        u, sv, vh = torch.linalg.svd(e, full_matrices=False)
        for k in range(1, max_flips + 1):
            res = e - (u[:, :k] * sv[:k]) @ vh[:k]
            if float(res.abs().max()) <= tol:
                big = float((u[:, :1] * sv[:1] @ vh[:1]).abs().max())
                assert big <= flip_size * scale, f"{what}: rank-{k} part of the error is {big / scale:.2e} of the scale - not a single-patch flip"
                return k
        raise AssertionError(f"{what}: max err {float(e.abs().max()):.3e} > {tol:.3e} (scale {scale:.3e}) and not explained by <= {max_flips} rank-one (ReLU-flip) terms; "
                             f"singular values {sv[:5].tolist()}")
    assert float(e.abs().max()) <= tol + flip_size * scale, f"{what}: max err {float(e.abs().max()):.3e} (scale {scale:.3e})"
    return 1


def relu_flip_positions(params, x, h1_dev, h_dev):
    """Positions [k, 2] = (patch, unit) where the device's ReLU masks (h1 > 0, h > 0) differ from the masks of the exact (fp64) forward,
    and a check that every such flip is LEGITIMATE: the exact pre-activation there is within fp32 round-off of zero (two correct fp32
    implementations can land on either side). models/model_toad.py:59-64."""
    p = {k: v.double() for k, v in params.items()}
    z1 = torch.addmm(p["attention_net.0.bias"], x.double(), p["attention_net.0.weight"].t())
    h1d = h1_dev.detach().cpu()
    f1 = (h1d > 0) != (z1 > 0)
    # layer 2 of the EXACT chain (the reference's fp64 run is the comparison target: its h comes from its own exact h1)
    z2 = torch.addmm(p["attention_net.2.bias"], torch.relu(z1), p["attention_net.2.weight"].t())
    f2 = (h_dev.detach().cpu() > 0) != (z2 > 0)
    for z, f, nm in ((z1, f1, "h1"), (z2, f2, "h")):
        if f.any():
            worst = z[f].abs().max().item()
            assert worst <= 2e-5 * max(z.abs().max().item(), 1e-30), f"{nm}: mask flip at |pre-activation| = {worst:.3e} is not round-off"
    return f1.nonzero(), f2.nonzero()


def relu_flips(params, x, h1_dev, h_dev):
    """Number of legitimate ReLU-mask flips in layer 1 and layer 2 (relu_flip_positions)."""
    i1, i2 = relu_flip_positions(params, x, h1_dev, h_dev)
    return int(i1.shape[0]), int(i2.shape[0])


TRUNK_KEYS = ("attention_net.0.weight", "attention_net.0.bias", "attention_net.2.weight", "attention_net.2.bias")


def check_trunk_grads_vs_golden_blocks(golden, name, grads, x, flips1, flips2):
    """The four trunk gradients against the REFERENCE's fp64 values (fixture matrices grad_block64/*), tolerant of exactly what the known
    legitimate ReLU-mask flips can do and of nothing else (models/model_toad.py:59-64 under autograd):
      * a layer-2 flip at (patch n, unit j) changes dZ2[n, j] only: dW2 row j and db2[j] move, every other row / entry must match;
        through dZ1[n, :] = (dZ2[n, :] W2) * mask1 it adds ONE rank-one term c x_n^T to dW1 (x_n = the patch's feature row, known)
        and the column c to db1;
      * a layer-1 flip at (n, j) changes dZ1[n, j] only: dW1 row j and db1[j].
    So: rows of dW2 / entries of db2 outside the flipped units are held to the reference; rows of the dW1 block outside the layer-1
    flipped units are held to it after removing their component in span{x_n : layer-2 flips} (at most a few dozen of 1024 directions),
    and the coefficients that projection finds must explain db1's deviation on the same rows. Returns the number of block rows checked."""
    pre = name + "/"
    w1k, b1k, w2k, b2k = TRUNK_KEYS
    if pre + "grad_block64/" + w1k not in golden.files:
        return 0
    onoise = oracle_fp32_noise(golden, name)

    def tol_of(k):
        dev = max(float(golden[pre + "grad_dev64/" + k]), onoise[k])
        # the same bound check_outputs_vs_golden uses - except that grad_dev64 of a flip-affected gradient already CONTAINS the
        # reference's own fp32 flips; cap the noise term so that it cannot swallow a real error on an unaffected row
        return max(2e-5, min(10.0 * dev / max(float(golden[pre + "grad_absmax/" + k]), 1e-300), 2e-4)) * float(golden[pre + "grad_absmax/" + k])

    u1 = set(int(j) for j in flips1[:, 1].tolist()) if len(flips1) else set()
    u2 = set(int(j) for j in flips2[:, 1].tolist()) if len(flips2) else set()
    checked = 0
    # ---- layer 2: rows / entries outside the flipped units
    blk = torch.from_numpy(golden[pre + "grad_block_rows"]).long()             # output units (rows of dW1 / dW2) the fixture carries
    g2 = grads[w2k].detach().cpu().double()
    ref2 = torch.from_numpy(golden[pre + "grad_block64/" + w2k]).double()
    rows2 = [i for i, r in enumerate(blk.tolist()) if r not in u2]
    e2 = (g2[blk] - ref2)[rows2]
    assert float(e2.abs().max()) <= tol_of(w2k), f"{name}: dW2 rows without a flipped unit deviate from the reference by {float(e2.abs().max()):.3e} (tol {tol_of(w2k):.3e})"
    checked += len(rows2)
    gb2 = grads[b2k].detach().cpu().double()
    refb2 = torch.from_numpy(golden[pre + "grad_block64/" + b2k]).double()
    keep = torch.tensor([j not in u2 for j in range(512)])
    assert float((gb2 - refb2)[keep].abs().max()) <= tol_of(b2k), f"{name}: db2 entries without a flipped unit deviate from the reference"
    # ---- layer 1: block rows outside the layer-1 flipped units, minus their component along the layer-2 flipped patches' feature rows
    g1 = grads[w1k].detach().cpu().double()
    ref1 = torch.from_numpy(golden[pre + "grad_block64/" + w1k]).double()
    rows1 = [i for i, r in enumerate(blk.tolist()) if r not in u1]
    e1 = (g1[blk] - ref1)[rows1]
    gb1 = grads[b1k].detach().cpu().double()
    eb1 = (gb1 - torch.from_numpy(golden[pre + "grad_block64/" + b1k]).double())[blk][rows1]
    patches = sorted(set(int(n) for n in flips2[:, 0].tolist())) if len(flips2) else []
    if patches:
        xs = x[patches].double()                                  # [k, 1024]
        coef = torch.linalg.lstsq(xs.t(), e1.t()).solution.t()    # [rows, k]: e1 ~ coef @ xs
        explained = coef @ xs
        scale1 = float(golden[pre + "grad_absmax/" + w1k])
        assert float(explained.abs().max()) <= 5e-2 * scale1, f"{name}: the part of dW1's deviation along the flipped patches is {float(explained.abs().max()) / scale1:.2e} of its scale"
        e1 = e1 - explained
        eb1 = eb1 - coef.sum(1)                                   # the same dZ1 rows summed give db1's deviation
    assert float(e1.abs().max()) <= tol_of(w1k), f"{name}: dW1 block deviates from the reference by {float(e1.abs().max()):.3e} beyond what {len(patches)} flipped patches explain (tol {tol_of(w1k):.3e})"
    assert float(eb1.abs().max()) <= tol_of(b1k) + 1e-6 * float(golden[pre + "grad_absmax/" + w1k]), f"{name}: db1 deviates from the reference by {float(eb1.abs().max()):.3e} beyond the flips' contribution"
    return checked + len(rows1)


def check_activations_vs_golden(golden, name, h1_dev, h_dev, atol=1e-4):
    """H1 / H of the device forward against strided samples (and sums) of the reference's own activations, captured with forward hooks
    on its two ReLU modules (models/model_toad.py:59-64) - for the cases the fixture carries them (the BASELINE sizes)."""
    pre = name + "/"
    if pre + "act_sample/h1" not in golden.files:
        return False
    for k, t in (("h1", h1_dev), ("h", h_dev)):
        t = t.detach().cpu()
        ref = golden[pre + "act_sample/" + k]
        scale = max(float(golden[pre + "act_absmax/" + k]), 1.0)
        assert_close(strided_sample(t, ref.shape[0]), ref, atol * scale, what=f"{name}:{k} sample")
        assert abs(float(t.abs().max()) - float(golden[pre + "act_absmax/" + k])) <= atol * scale, (name, k)
        s, ss = float(t.double().sum()), float((t.double() ** 2).sum())
        assert abs(s - float(golden[pre + "act_sum/" + k])) <= 1e-6 * abs(float(golden[pre + "act_sum/" + k])) + 1e-3, (name, k, "sum")
        assert abs(ss - float(golden[pre + "act_sumsq/" + k])) <= 2e-6 * float(golden[pre + "act_sumsq/" + k]) + 1e-3, (name, k, "sumsq")
    return True


# Which gradients a ReLU-mask flip can reach. The masks enter the backward only where dZ = dH * (H > 0) is formed
# (models/model_toad.py:59-64 under autograd): a flip in layer 2's mask changes dZ2 -> dW2, db2 and, through dZ1 = (dZ2 W2) * mask1,
# dW1, db1; a flip in layer 1's mask changes dZ1 -> dW1, db1 only. The ten attention / head gradients are functions of the forward
# VALUES (which a flip moves by round-off) and never of the masks, so they are compared with the reference's golden values always.
MASK_FREE_KEYS = tuple(k for k in orc.PARAM_KEYS if not (k.startswith("attention_net.0.") or k.startswith("attention_net.2.")))
LAYER2_KEYS = tuple(k for k in orc.PARAM_KEYS if k.startswith("attention_net.2."))


def check_batch_against_oracle(name, params, slides, offs, dev, grads, masks=None):
    """The checker behind tests/test_gpu_multi_step.py::test_config4_shape_batches_match_the_oracle (and its CPU self-test in
    tests/test_oracle_golden.py). ``slides``: [(x, sex, label, site)] CPU tensors; ``offs``: row offsets of the concatenation; ``dev``: what the
    device produced, as CPU tensors - h1, h [rows, 512], p [rows, 2D], a_raw [rows, 2] of the concatenation, logits [B, C], site_logits [B, 2],
    loss [B, 3] (scaled by 1/B like the call's loss weights); ``grads``: {slot: gradient of the batch}; ``masks``: per slide the exported dropout
    multipliers {"h1", "h", "a", "b"} or None. Reference semantics: utils/core_utils_mtl_concat.py:200-234 per slide, models/model_toad.py:90-116.
      * per slide: logits / site logits / loss / A_raw against the oracle's fp32 forward, 1e-4 absolute;
      * H1 / H against the exact fp64 forward, 1e-4 of their abs-max; where the device's ReLU mask differs from the exact one the exact
        pre-activation must be round-off of zero;
      * all 14 gradients against the sum over slides of the oracle's fp64 backward on the DEVICE's activations (identical masks): 2e-5 of each
        gradient's own scale + 10 x (32 x for the three cancellation-dominated ones) the fp32 noise of the same backward.
    Returns (sum of fp32 backward, sum of fp64 backward, [mask differences layer 1, layer 2])."""
    B = len(slides)
    p64 = {k: v.double() for k, v in params.items()}
    tot64 = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in params.items()}
    tot32 = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in params.items()}
    flips = [0, 0]
    for b, (x, sx, lb, st) in enumerate(slides):
        r0, r1 = offs[b], offs[b + 1]
        mk = None if masks is None else masks[b]
        o_out, _ = orc.forward(params, x, sx, masks=mk)
        o_loss = orc.loss_fn(o_out["logits"], lb, o_out["site_logits"], st)
        assert (dev["logits"][b] - o_out["logits"][0]).abs().max().item() <= 1e-4, (name, b, "logits")
        assert (dev["site_logits"][b] - o_out["site_logits"][0]).abs().max().item() <= 1e-4, (name, b, "site_logits")
        assert abs(dev["loss"][b][0].item() * B - float(o_loss)) <= 1e-4, (name, b, "loss")
        assert (dev["a_raw"][r0:r1] - o_out["A"].t()).abs().max().item() <= 1e-4, (name, b, "A_raw")
        z1 = torch.addmm(p64["attention_net.0.bias"], x.double(), p64["attention_net.0.weight"].t())
        e1 = torch.relu(z1) if mk is None else torch.relu(z1) * mk["h1"].double()
        z2 = torch.addmm(p64["attention_net.2.bias"], e1, p64["attention_net.2.weight"].t())
        e2 = torch.relu(z2) if mk is None else torch.relu(z2) * mk["h"].double()
        for li, (z, e, hd, key) in enumerate(((z1, e1, dev["h1"][r0:r1], "h1"), (z2, e2, dev["h"][r0:r1], "h"))):
            assert (hd.double() - e).abs().max().item() <= 1e-4 * max(e.abs().max().item(), 1.0), (name, b, key)
            f = (hd > 0) != (z > 0)
            if mk is not None:
                f &= mk[key] > 0
            if f.any():
                assert z[f].abs().max().item() <= 2e-5 * z.abs().max().item(), f"{name}: slide {b} {key}: a mask differs at a pre-activation that is not round-off"
            flips[li] += int(f.sum())
        del z1, z2, e1, e2
        for dt, tot, pp in ((torch.float64, tot64, p64), (torch.float32, tot32, params)):
            s_ = orc.Saved(x=x.to(dt), h1=dev["h1"][r0:r1].to(dt), h=dev["h"][r0:r1].to(dt), p=dev["p"][r0:r1].to(dt), a_raw=dev["a_raw"][r0:r1].to(dt),
                           m=None, mcat=None, sex=sx.to(dt), masks=None if mk is None else {k: v.to(dt) for k, v in mk.items()})
            s_.m = orc.softmax_pool(s_.a_raw, s_.h)
            s_.mcat, lg, _, _, sl, _, _ = orc.heads_fwd(s_.m, s_.sex, pp["classifier.weight"], pp["classifier.bias"], pp["site_classifier.weight"],
                                                         pp["site_classifier.bias"])
            dl, ds = orc.loss_grad(lg, lb, sl, st)
            gb = orc.backward(pp, s_, dl, ds)
            for k in tot:
                tot[k] += gb[k].double() / B
    rows = offs[-1]
    assert flips[0] + flips[1] <= max(8, rows // 5000), (name, flips)      # a handful per 10^8 elements on random bags, never a pattern
    for slot, key in SLOT2KEY.items():
        noise = (tot32[key] - tot64[key]).abs().max().item()
        assert_grad_close(grads[slot], tot64[key], 2e-5, grad_scale(tot64, key), what=f"{name}: {key} ({flips[0]}+{flips[1]} legit mask differences)",
                          floor=(32.0 if slot in ("ba", "bb", "bc") else 10.0) * noise)
    return tot32, tot64, flips


# ---- whole-slide calls against the per-op sequence (round 6) --------------------------------------------------------------------------------
WGRAD_BATCH_MIN_ROWS, WGRAD_BATCH_MAX_ROWS = 64, 262144         # csrc/common.h kTnBatchMaxRows, gemm_f32.hip wgrad_batch_ok
BATCHED_WGRAD_SLOTS = ("w1", "b1", "w2", "b2", "wab", "bab")


def assert_step_grad_matches_per_op(got, ref, slot, n_rows):
    """A gradient of a whole-slide call (toad_mil_bwd_f32 / toad_mil_step_f32) against the same gradient from the per-op calls. Same kernels on the
    same operands in the same order: bitwise - except, for bags of 64 ... 262,144 rows, the three trunk / attention weight gradients (and their bias
    gradients, which are column sums formed by the same kernel): the whole-slide call runs them as ONE launch (gemm_tn_h2_batch_kernel) whose row
    splits differ from those of three separate launches, i.e. the same products summed in a different fixed order. 2e-6 of the tensor's abs-max bounds
    fp32 summation round-off over <= 262k rows with room (measured 1e-7 ... 4e-7); each route stays bitwise reproducible run to run."""
    import torch
    if slot in BATCHED_WGRAD_SLOTS and WGRAD_BATCH_MIN_ROWS <= n_rows <= WGRAD_BATCH_MAX_ROWS:
        scale = max(ref.abs().max().item(), 1e-30)
        err = (got - ref).abs().max().item()
        assert err <= 2e-6 * scale, (slot, err, scale)
    else:
        assert torch.equal(got, ref), slot
