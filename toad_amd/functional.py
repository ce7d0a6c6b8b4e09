"""Forward / backward of the TOAD MIL path on the C ABI.

Two routes to the same kernels, in the same order, with bitwise-equal results:

  * ``ToadMIL`` (the autograd bridge behind ``model(data, sex)`` / ``loss.backward()``, utils/core_utils_mtl_concat.py:206,231)
    makes ONE library call per direction (``toad_mil_fwd_f32`` / ``toad_mil_bwd_f32``): forward arena owned by the autograd
    context, scratch cached per stream, nothing else allocated.
  * ``mil_forward`` / ``mil_backward``: the same sequence as per-op calls over plain tensors (kernel-level tests, empty bags,
    bags beyond the whole-slide entry points' shapes).  Kernel order:

  fwd:  [abs-max of X] -> linear+relu -> linear+relu -> linear([Wa;Wb]) -> fused gated pool -> heads
  bwd:  heads_bwd -> gated_pool_bwd -> wgrad(ab) -> dgrad(ab)+pool+mask -> wgrad(2) -> dgrad(2)+mask -> wgrad(1)

The abs-max arrays that scale the fp16 two-piece GEMM operands travel with the tensors: each GEMM / the pooling backward emits
the array of its output, the consumer takes it as an argument, so no tensor is re-read just to be measured (only the caller's
bag is, once).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import ops

# parameter slots, in the order the reference's state dict lists them
SLOTS = ("w1", "b1", "w2", "b2", "wa", "ba", "wb", "bb", "wc", "bc", "wcls", "bcls", "wsite", "bsite")


DROP_P = 0.25          # nn.Dropout(0.25): models/model_toad.py:27-29,61,64
_GOLDEN = 0x9E3779B97F4A7C15


def drop_seeds(seed: int):
    """Four independent streams from one per-forward seed: trunk layer 1, trunk layer 2, tanh branch, sigmoid branch."""
    return tuple((seed + (i + 1) * _GOLDEN) & 0xFFFFFFFFFFFFFFFF for i in range(4))


@dataclass
class Saved:
    x: torch.Tensor
    h1: torch.Tensor
    h: torch.Tensor
    p: torch.Tensor        # [N, 2D] pre-activations (Pa | Pb)
    a_raw: torch.Tensor    # [N, T]
    stats: torch.Tensor    # [T, 2] (max, sum-exp) per task
    m: torch.Tensor        # [T, L]
    mcat: torch.Tensor     # [T, L+1]
    drop_p: float = 0.0    # 0 = no dropout (eval, or dropout=False)
    seed: int = 0
    x_amax: Optional[torch.Tensor] = None    # abs-max arrays of x, h1, h (operand scales of the backward GEMMs)
    h1_amax: Optional[torch.Tensor] = None
    h_amax: Optional[torch.Tensor] = None
    h1_bits: Optional[torch.Tensor] = None   # one-bit ReLU images of h1, h (the dgrad epilogues read these instead of 32x the bytes)
    h_bits: Optional[torch.Tensor] = None


def _stack_ab(w: Dict[str, torch.Tensor]):
    """[Wa;Wb] and [ba;bb] as single tensors. Zero-copy when the module keeps them adjacent in its
    flat parameter buffer (``wab``/``bab`` views supplied by the module), else one small cat."""
    if "wab" in w:
        return w["wab"], w["bab"]
    return torch.cat([w["wa"], w["wb"]], 0), torch.cat([w["ba"], w["bb"]], 0)


def trunk_scores(w: Dict[str, torch.Tensor], x: torch.Tensor, drop_p: float = 0.0, seed: int = 0, x_amax=None):
    """models/model_toad.py:59-64 trunk (+Dropout when training with dropout=True) + :21,:25 stacked
    attention pre-activations. Returns (h1, h, p, (x_amax, h1_amax, h_amax, h1_bits, h_bits))."""
    s1, s2, _, _ = drop_seeds(seed)
    # a bag without an abs-max array is measured inside the first GEMM (as in the whole-slide calls: the two routes stay bitwise equal); the
    # array itself - the weight gradient of this layer scales the bag with its maximum - is taken by a pass of its own on this per-op route
    h1, h1_amax, h1_bits = ops.linear_act_fwd(x, w["w1"], w["b1"], ops.ACT_RELU, drop_p=drop_p, drop_seed=s1, x_amax=x_amax, want_bits=True)
    if x_amax is None:
        x_amax = ops.absmax_rows256(x)
    h, h_amax, h_bits = ops.linear_act_fwd(h1, w["w2"], w["b2"], ops.ACT_RELU, drop_p=drop_p, drop_seed=s2, x_amax=h1_amax, want_bits=True)
    wab, bab = _stack_ab(w)
    p = ops.linear_act_fwd(h, wab, bab, ops.ACT_NONE, x_amax=h_amax)
    return h1, h, p, (x_amax, h1_amax, h_amax, h1_bits, h_bits)


def mil_forward(w: Dict[str, torch.Tensor], x: torch.Tensor, sex: torch.Tensor, drop_p: float = 0.0, seed: int = 0):
    """TOAD_fc_mtl_concat.forward (models/model_toad.py:90-116) without the python dict.
    ``drop_p`` > 0 = training with dropout=True; ``seed`` selects the masks (recomputed in backward)."""
    d = w["wa"].shape[0]
    if x.shape[0] == 0:
        # empty bag: the reference's softmax over zero patches followed by mm([T,0],[0,L]) gives M = 0 and the heads still
        # run on (0, sex) (model_toad.py:96-107); there is no trunk / pooling work to launch
        l, t = w["w2"].shape[0], w["wc"].shape[0]
        e = lambda c: torch.empty((0, c), dtype=torch.float32, device=x.device)
        h1, h, p, a_raw = e(w["w1"].shape[0]), e(l), e(2 * d), e(t)
        m = torch.zeros((t, l), dtype=torch.float32, device=x.device)
        stats = torch.zeros((t, 2), dtype=torch.float32, device=x.device)
        amax = (None, None, None, None, None)
    else:
        h1, h, p, amax = trunk_scores(w, x, drop_p, seed)
        _, _, sa, sb = drop_seeds(seed)
        a_raw, m, stats = ops.gated_pool_fwd(p, d, h, w["wc"], w["bc"], drop_p, sa, sb)
    mcat, logits, y_prob, y_hat, site_logits, site_prob, site_hat = ops.heads_fwd(
        m, sex, w["wcls"], w["bcls"], w["wsite"], w["bsite"])
    saved = Saved(x=x, h1=h1, h=h, p=p, a_raw=a_raw, stats=stats, m=m, mcat=mcat, drop_p=drop_p, seed=seed,
                  x_amax=amax[0], h1_amax=amax[1], h_amax=amax[2], h1_bits=amax[3], h_bits=amax[4])
    outs = dict(logits=logits, Y_prob=y_prob, Y_hat=y_hat, site_logits=site_logits, site_prob=site_prob,
                site_hat=site_hat, A_nt=a_raw, features=mcat)
    return outs, saved


def attention_scores(w: Dict[str, torch.Tensor], x: torch.Tensor, drop_p: float = 0.0, seed: int = 0) -> torch.Tensor:
    """attention_only path (models/model_toad.py:93-94): A_raw [N,T] without pooling."""
    if x.shape[0] == 0:
        return torch.empty((0, w["wc"].shape[0]), dtype=torch.float32, device=x.device)
    _, _, p, _ = trunk_scores(w, x, drop_p, seed)
    _, _, sa, sb = drop_seeds(seed)
    a_raw, _, _ = ops.gated_pool_fwd(p, w["wa"].shape[0], None, w["wc"], w["bc"], drop_p, sa, sb)
    return a_raw


def mil_backward(w: Dict[str, torch.Tensor], s: Saved, dlogits: torch.Tensor, dsite: torch.Tensor,
                 da_ext: Optional[torch.Tensor] = None, dmcat_ext: Optional[torch.Tensor] = None,
                 grads: Optional[Dict[str, torch.Tensor]] = None, beta: float = 0.0,
                 need_dx: bool = False, need_dsex: bool = False):
    """Backward of mil_forward. ``grads`` (slot -> destination, plus optional 'wab'/'bab' stacked
    views) receives ``beta*old + new``; when None fresh tensors are returned (beta ignored).
    Returns (gradient dict, dX | None) - and dsex [1] as a third value with ``need_dsex``."""
    d = w["wa"].shape[0]
    g: Dict[str, torch.Tensor] = {}
    if grads is None:
        beta = 0.0
    hg = None if grads is None else (grads["wcls"], grads["bcls"], grads["wsite"], grads["bsite"])
    hb = ops.heads_bwd(s.mcat, dlogits, dsite, w["wcls"], w["wsite"], dmcat_ext, hg, beta, want_dsex=need_dsex)
    g["wcls"], g["bcls"], g["wsite"], g["bsite"], dm = hb[:5]
    dsex = hb[5] if need_dsex else None

    def ret(dx):
        return (g, dx, dsex) if need_dsex else (g, dx)

    if s.x.shape[0] == 0:
        # empty bag: nothing upstream of the pooled features received data, so those gradients are exactly zero
        for k in ("w1", "b1", "w2", "b2", "wa", "ba", "wb", "bb", "wc", "bc"):
            if grads is None:
                g[k] = torch.zeros_like(w[k])
            else:
                g[k] = grads[k].mul_(beta)
        return ret(torch.empty_like(s.x) if need_dx else None)
    n = s.x.shape[0]
    l = s.h.shape[1]
    _, _, sa, sb = drop_seeds(s.seed)
    mscale = 1.0 / (1.0 - s.drop_p) if s.drop_p > 0 else 1.0      # ReLU+Dropout outputs: zeros already carry the mask
    # the pooling gradient dH_pool = softmax(A) dM is recomputed inside the dgrad epilogue when the fp16 two-piece kernel serves
    # the shape (always, up to ~1 M patches); otherwise the pooling backward materialises it
    fused_pool = ops.h2_ok(n, l, 2 * d)
    dp, dh, g["wc"], g["bc"], dp_amax = ops.gated_pool_bwd(
        s.p, d, s.h, w["wc"], s.a_raw, s.stats, s.m, dm, da_ext,
        None if grads is None else grads["wc"], None if grads is None else grads["bc"], beta,
        drop_p=s.drop_p, seed_a=sa, seed_b=sb, want_dh=not fused_pool, want_amax=True)
    # attention_a / attention_b Linear (stacked)
    wab, _ = _stack_ab(w)
    if grads is not None and "wab" in grads:
        dwab, dbab = ops.linear_wgrad(dp, s.h, grads["wab"], grads["bab"], beta, dy_amax=dp_amax, x_amax=s.h_amax)
    elif grads is not None:
        # destinations are not adjacent: reduce into a temporary and accumulate the halves
        dwab, dbab = ops.linear_wgrad(dp, s.h, dy_amax=dp_amax, x_amax=s.h_amax)
        for k, v in (("wa", dwab[:d]), ("wb", dwab[d:]), ("ba", dbab[:d]), ("bb", dbab[d:])):
            grads[k].mul_(beta).add_(v)
    else:
        dwab, dbab = ops.linear_wgrad(dp, s.h, dy_amax=dp_amax, x_amax=s.h_amax)
    g["wa"], g["wb"], g["ba"], g["bb"] = dwab[:d], dwab[d:], dbab[:d], dbab[d:]
    # dZ2 = (dP Wab + dH_pool) * (H > 0)
    if fused_pool:
        dz2, dz2_amax = ops.linear_dgrad(dp, ops.transpose(wab), relu_src=s.h, mask_scale=mscale, pool=(s.a_raw, s.stats, dm),
                                         dy_amax=dp_amax, want_amax=True, relu_bits=s.h_bits)
    else:
        dz2, dz2_amax = ops.linear_dgrad(dp, ops.transpose(wab), addend=dh, relu_src=s.h, out=dh, mask_scale=mscale,
                                         dy_amax=dp_amax, want_amax=True)
    del dp
    g["w2"], g["b2"] = ops.linear_wgrad(dz2, s.h1, None if grads is None else grads["w2"],
                                        None if grads is None else grads["b2"], beta, dy_amax=dz2_amax, x_amax=s.h1_amax)
    dz1, dz1_amax = ops.linear_dgrad(dz2, ops.transpose(w["w2"]), relu_src=s.h1, mask_scale=mscale, dy_amax=dz2_amax, want_amax=True,
                                     relu_bits=s.h1_bits)
    del dz2
    g["w1"], g["b1"] = ops.linear_wgrad(dz1, s.x, None if grads is None else grads["w1"],
                                        None if grads is None else grads["b1"], beta, dy_amax=dz1_amax, x_amax=s.x_amax)
    dx = ops.linear_dgrad(dz1, ops.transpose(w["w1"]), dy_amax=dz1_amax) if need_dx else None
    return ret(dx)


class ToadMIL(torch.autograd.Function):
    """autograd bridge: inputs (x, sex, 14 parameters in SLOTS order, wab, bab, drop_p, seed).

    forward = toad_mil_fwd_f32, backward = toad_mil_bwd_f32: one C call each (an empty bag has no kernels to fuse and takes
    the per-op route). Gradients of the parameters are fresh tensors handed to autograd, which accumulates into ``.grad`` the
    way the reference's optimiser expects; ``sex`` and ``x`` get their gradients too when they require them."""

    @staticmethod
    def forward(ctx, x, sex, *params):
        w = dict(zip(SLOTS, params[:14]))
        if params[14] is not None:
            w["wab"], w["bab"] = params[14], params[15]
        drop_p, seed = params[16], params[17]
        ctx.w = w
        ctx.need_dx = x.requires_grad
        ctx.need_dsex = sex.requires_grad
        ctx.drop = (drop_p, seed)
        ctx.done = False
        n, c = x.shape[0], w["wcls"].shape[0]
        if n == 0 or "wab" not in w:
            outs, s = mil_forward(w, x, sex, drop_p, seed)
            ctx.s, ctx.arena, ctx.x = s, None, None
            ret = (outs["logits"], outs["site_logits"], outs["A_nt"], outs["features"], outs["Y_prob"], outs["Y_hat"],
                   outs["site_prob"], outs["site_hat"])
        else:
            arena = ops.mil_fwd(w, x, sex, drop_p, seed)
            ctx.s, ctx.arena, ctx.x = None, arena, x
            v = arena.view
            ret = (v("logits", (1, c)), v("site_logits", (1, 2)), v("a_raw", (n, 2)), v("mcat", (2, 513)), v("y_prob", (1, c)),
                   v("y_hat", (1, 1), torch.int64), v("site_prob", (1, 2)), v("site_hat", (1, 1), torch.int64))
        ctx.mark_non_differentiable(ret[4], ret[5], ret[6], ret[7])
        return ret

    @staticmethod
    def backward(ctx, dlogits, dsite, da, dfeat, *unused):
        w = ctx.w
        c = w["wcls"].shape[0]
        dev = w["wcls"].device
        dlogits = torch.zeros((1, c), device=dev) if dlogits is None else dlogits.contiguous()
        dsite = torch.zeros((1, 2), device=dev) if dsite is None else dsite.contiguous()
        da = None if da is None else da.contiguous()
        dfeat = None if dfeat is None else dfeat.contiguous()
        # the saved activations stay alive with the graph, so backward(retain_graph=True) can be followed by another backward
        if ctx.arena is None:
            g, dx, dsex = mil_backward(w, ctx.s, dlogits, dsite, da, dfeat, need_dx=ctx.need_dx, need_dsex=True)
        else:
            g = {k: torch.empty_like(w[k]) for k in ops.STEP_SLOTS}
            drop_p, seed = ctx.drop
            dx, dsex = ops.mil_bwd(w, g, 0.0, ctx.x, ctx.arena, dlogits, dsite, da, dfeat, drop_p, seed,
                                   need_dx=ctx.need_dx, need_dsex=True)
            d = w["wa"].shape[0]
            g["wa"], g["wb"], g["ba"], g["bb"] = g["wab"][:d], g["wab"][d:], g["bab"][:d], g["bab"][d:]
        if not ctx.need_dsex:
            dsex = None
        return (dx, dsex) + tuple(g[k] for k in SLOTS) + (None, None, None, None)
