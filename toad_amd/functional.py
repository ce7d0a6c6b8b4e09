"""Forward / backward of the TOAD MIL path as a sequence of C-ABI kernel calls.

``mil_forward`` and ``mil_backward`` are plain functions over tensors (no autograd); the
``ToadMIL`` autograd.Function wraps them so ``loss.backward()`` in the reference's train loop
(utils/core_utils_mtl_concat.py:231) drives the HIP backward.  Kernel order:

  fwd:  linear+relu -> linear+relu -> linear([Wa;Wb]) -> fused gated pool -> heads
  bwd:  heads_bwd -> gated_pool_bwd -> wgrad(ab) -> dgrad(ab)+mask -> wgrad(2) -> dgrad(2)+mask -> wgrad(1)
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


def _stack_ab(w: Dict[str, torch.Tensor]):
    """[Wa;Wb] and [ba;bb] as single tensors. Zero-copy when the module keeps them adjacent in its
    flat parameter buffer (``wab``/``bab`` views supplied by the module), else one small cat."""
    if "wab" in w:
        return w["wab"], w["bab"]
    return torch.cat([w["wa"], w["wb"]], 0), torch.cat([w["ba"], w["bb"]], 0)


def trunk_scores(w: Dict[str, torch.Tensor], x: torch.Tensor, drop_p: float = 0.0, seed: int = 0):
    """models/model_toad.py:59-64 trunk (+Dropout when training with dropout=True) + :21,:25 stacked
    attention pre-activations."""
    s1, s2, _, _ = drop_seeds(seed)
    h1 = ops.linear_act_fwd(x, w["w1"], w["b1"], ops.ACT_RELU, drop_p=drop_p, drop_seed=s1)
    h = ops.linear_act_fwd(h1, w["w2"], w["b2"], ops.ACT_RELU, drop_p=drop_p, drop_seed=s2)
    wab, bab = _stack_ab(w)
    p = ops.linear_act_fwd(h, wab, bab, ops.ACT_NONE)
    return h1, h, p


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
    else:
        h1, h, p = trunk_scores(w, x, drop_p, seed)
        _, _, sa, sb = drop_seeds(seed)
        a_raw, m, stats = ops.gated_pool_fwd(p, d, h, w["wc"], w["bc"], drop_p, sa, sb)
    mcat, logits, y_prob, y_hat, site_logits, site_prob, site_hat = ops.heads_fwd(
        m, sex, w["wcls"], w["bcls"], w["wsite"], w["bsite"])
    saved = Saved(x=x, h1=h1, h=h, p=p, a_raw=a_raw, stats=stats, m=m, mcat=mcat, drop_p=drop_p, seed=seed)
    outs = dict(logits=logits, Y_prob=y_prob, Y_hat=y_hat, site_logits=site_logits, site_prob=site_prob,
                site_hat=site_hat, A_nt=a_raw, features=mcat)
    return outs, saved


def attention_scores(w: Dict[str, torch.Tensor], x: torch.Tensor, drop_p: float = 0.0, seed: int = 0) -> torch.Tensor:
    """attention_only path (models/model_toad.py:93-94): A_raw [N,T] without pooling."""
    if x.shape[0] == 0:
        return torch.empty((0, w["wc"].shape[0]), dtype=torch.float32, device=x.device)
    _, _, p = trunk_scores(w, x, drop_p, seed)
    _, _, sa, sb = drop_seeds(seed)
    a_raw, _, _ = ops.gated_pool_fwd(p, w["wa"].shape[0], None, w["wc"], w["bc"], drop_p, sa, sb)
    return a_raw


def mil_backward(w: Dict[str, torch.Tensor], s: Saved, dlogits: torch.Tensor, dsite: torch.Tensor,
                 da_ext: Optional[torch.Tensor] = None, dmcat_ext: Optional[torch.Tensor] = None,
                 grads: Optional[Dict[str, torch.Tensor]] = None, beta: float = 0.0,
                 need_dx: bool = False):
    """Backward of mil_forward. ``grads`` (slot -> destination, plus optional 'wab'/'bab' stacked
    views) receives ``beta*old + new``; when None fresh tensors are returned (beta ignored)."""
    d = w["wa"].shape[0]
    g: Dict[str, torch.Tensor] = {}
    if grads is None:
        beta = 0.0
    hg = None if grads is None else (grads["wcls"], grads["bcls"], grads["wsite"], grads["bsite"])
    g["wcls"], g["bcls"], g["wsite"], g["bsite"], dm = ops.heads_bwd(
        s.mcat, dlogits, dsite, w["wcls"], w["wsite"], dmcat_ext, hg, beta)
    if s.x.shape[0] == 0:
        # empty bag: nothing upstream of the pooled features received data, so those gradients are exactly zero
        for k in ("w1", "b1", "w2", "b2", "wa", "ba", "wb", "bb", "wc", "bc"):
            if grads is None:
                g[k] = torch.zeros_like(w[k])
            else:
                g[k] = grads[k].mul_(beta)
        return g, (torch.empty_like(s.x) if need_dx else None)
    _, _, sa, sb = drop_seeds(s.seed)
    mscale = 1.0 / (1.0 - s.drop_p) if s.drop_p > 0 else 1.0      # ReLU+Dropout outputs: zeros already carry the mask
    dp, dh, g["wc"], g["bc"] = ops.gated_pool_bwd(
        s.p, d, s.h, w["wc"], s.a_raw, s.stats, s.m, dm, da_ext,
        None if grads is None else grads["wc"], None if grads is None else grads["bc"], beta,
        drop_p=s.drop_p, seed_a=sa, seed_b=sb)
    # attention_a / attention_b Linear (stacked)
    wab, _ = _stack_ab(w)
    if grads is not None and "wab" in grads:
        dwab, dbab = ops.linear_wgrad(dp, s.h, grads["wab"], grads["bab"], beta)
    elif grads is not None:
        # destinations are not adjacent: reduce into a temporary and accumulate the halves
        dwab, dbab = ops.linear_wgrad(dp, s.h)
        for k, v in (("wa", dwab[:d]), ("wb", dwab[d:]), ("ba", dbab[:d]), ("bb", dbab[d:])):
            grads[k].mul_(beta).add_(v)
    else:
        dwab, dbab = ops.linear_wgrad(dp, s.h)
    g["wa"], g["wb"], g["ba"], g["bb"] = dwab[:d], dwab[d:], dbab[:d], dbab[d:]
    # dZ2 = (dP Wab + dH_pool) * (H > 0), written in place over dH_pool
    dz2 = ops.linear_dgrad(dp, ops.transpose(wab), addend=dh, relu_src=s.h, out=dh, mask_scale=mscale)
    del dp
    g["w2"], g["b2"] = ops.linear_wgrad(dz2, s.h1, None if grads is None else grads["w2"],
                                        None if grads is None else grads["b2"], beta)
    dz1 = ops.linear_dgrad(dz2, ops.transpose(w["w2"]), relu_src=s.h1, mask_scale=mscale)
    del dz2
    g["w1"], g["b1"] = ops.linear_wgrad(dz1, s.x, None if grads is None else grads["w1"],
                                        None if grads is None else grads["b1"], beta)
    dx = ops.linear_dgrad(dz1, ops.transpose(w["w1"])) if need_dx else None
    return g, dx


class ToadMIL(torch.autograd.Function):
    """autograd bridge: inputs (x, sex, 14 parameters in SLOTS order, wab, bab)."""

    @staticmethod
    def forward(ctx, x, sex, *params):
        w = dict(zip(SLOTS, params[:14]))
        if params[14] is not None:
            w["wab"], w["bab"] = params[14], params[15]
        drop_p, seed = params[16], params[17]
        outs, s = mil_forward(w, x, sex, drop_p, seed)
        ctx.w = w
        ctx.s = s
        ctx.need_dx = x.requires_grad
        ctx.mark_non_differentiable(outs["Y_prob"], outs["Y_hat"], outs["site_prob"], outs["site_hat"])
        return (outs["logits"], outs["site_logits"], outs["A_nt"], outs["features"], outs["Y_prob"], outs["Y_hat"],
                outs["site_prob"], outs["site_hat"])

    @staticmethod
    def backward(ctx, dlogits, dsite, da, dfeat, *unused):
        w, s = ctx.w, ctx.s
        c = w["wcls"].shape[0]
        dev = s.mcat.device
        dlogits = torch.zeros((1, c), device=dev) if dlogits is None else dlogits.contiguous()
        dsite = torch.zeros((1, 2), device=dev) if dsite is None else dsite.contiguous()
        da = None if da is None else da.contiguous()
        dfeat = None if dfeat is None else dfeat.contiguous()
        g, dx = mil_backward(w, s, dlogits, dsite, da, dfeat, need_dx=ctx.need_dx)
        ctx.s = None
        dsex = None
        if dfeat is not None and ctx.needs_input_grad[1]:
            dsex = dfeat[:, -1].sum().reshape(1)
        return (dx, dsex) + tuple(g[k] for k in SLOTS) + (None, None, None, None)
