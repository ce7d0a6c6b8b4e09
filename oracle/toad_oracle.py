"""TEST INFRASTRUCTURE ONLY — CPU restatement of TOAD's gated-attention MIL hot path.

This file is the *oracle* for ``toad_amd``: a plain fp32 PyTorch-CPU restatement of
``/root/reference/models/model_toad.py`` (forward) plus a hand-derived backward that
is decomposed exactly like the HIP kernels (so every kernel has a CPU twin).
It is never imported by the product package and never runs on the GPU.

Parity status: PINNED.  ``oracle/pin_against_reference.py`` imports the unmodified
reference model in the build container, checks this restatement against it
(forward dict, all 14 parameter gradients, ``attention_only``/``return_features``
for N in {0,1,2,63,64,65,256,777,1024 (x30, saturated softmax),300 (all rows equal),10000,100000}, C in {2,18}) and writes the golden vectors committed under
``tests/golden/``.  The reference has no tests or golden vectors of its own
(SURVEY.md §4), so the imported reference is the pin.

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

# models/model_toad.py:56  size_dict = {"small": [1024, 512, 256], "big": [1024, 512, 384]}
SIZE_DICT = {"small": (1024, 512, 256), "big": (1024, 512, 384)}
N_TASKS = 2  # models/model_toad.py:66  Attn_Net_Gated(..., n_tasks = 2)

# State-dict key names of the reference for dropout=False (listed by import).
PARAM_KEYS = (
    "attention_net.0.weight", "attention_net.0.bias",
    "attention_net.2.weight", "attention_net.2.bias",
    "attention_net.4.attention_a.0.weight", "attention_net.4.attention_a.0.bias",
    "attention_net.4.attention_b.0.weight", "attention_net.4.attention_b.0.bias",
    "attention_net.4.attention_c.weight", "attention_net.4.attention_c.bias",
    "classifier.weight", "classifier.bias",
    "site_classifier.weight", "site_classifier.bias",
)


def param_shapes(n_classes: int, size_arg: str = "big") -> Dict[str, Tuple[int, ...]]:
    """Shapes created by models/model_toad.py:59-73 (trunk, gated attention, two heads)."""
    l0, l, d = SIZE_DICT[size_arg]
    return {
        "attention_net.0.weight": (l, l0), "attention_net.0.bias": (l,),
        "attention_net.2.weight": (l, l), "attention_net.2.bias": (l,),
        "attention_net.4.attention_a.0.weight": (d, l), "attention_net.4.attention_a.0.bias": (d,),
        "attention_net.4.attention_b.0.weight": (d, l), "attention_net.4.attention_b.0.bias": (d,),
        "attention_net.4.attention_c.weight": (N_TASKS, d), "attention_net.4.attention_c.bias": (N_TASKS,),
        "classifier.weight": (n_classes, l + 1), "classifier.bias": (n_classes,),
        "site_classifier.weight": (2, l + 1), "site_classifier.bias": (2,),
    }


# --------------------------------------------------------------------------------------
# Closed-form deterministic inputs (numpy only; reproducible on any box without the
# reference).  SURVEY.md §8(c) golden-vector recipe.
# --------------------------------------------------------------------------------------
def closed_form_params(n_classes: int, size_arg: str = "big", bias_scale: float = 0.05) -> Dict[str, torch.Tensor]:
    """w[i,j] = sin(0.37 i + 1.13 j + phi)*sqrt(2/(fan_in+fan_out)); small non-zero biases.

    The scale is Xavier's (utils/utils.py:150-154 uses xavier_normal_); biases are made
    non-zero on purpose so a dropped bias add is detected.
    """
    out = {}
    for idx, (k, shp) in enumerate(param_shapes(n_classes, size_arg).items()):
        phi = 0.61 * (idx + 1)
        if len(shp) == 2:
            i = np.arange(shp[0], dtype=np.float64)[:, None]
            j = np.arange(shp[1], dtype=np.float64)[None, :]
            std = math.sqrt(2.0 / (shp[0] + shp[1]))
            # 1.7*std: a sine has rms 1/sqrt(2); keep pre-activations O(1) like xavier-normal
            w = np.sin(0.37 * i + 1.13 * j + phi) * (1.7 * std)
        else:
            i = np.arange(shp[0], dtype=np.float64)
            w = np.cos(0.23 * i + phi) * bias_scale
        out[k] = torch.from_numpy(w.astype(np.float32))
    return out


def closed_form_bag(n: int, l0: int = 1024, scale: float = 1.0, kind: str = "wave") -> torch.Tensor:
    """x[n,k] = cos(0.011 n (k+1)) + 0.1 sin(0.7 k)  (kind='wave'); 'equal' = all rows equal."""
    k = np.arange(l0, dtype=np.float64)[None, :]
    if kind == "equal":
        row = np.cos(0.05 * (k + 1)) + 0.1 * np.sin(0.7 * k)
        x = np.repeat(row, n, axis=0)
    else:
        r = np.arange(n, dtype=np.float64)[:, None]
        x = np.cos(0.011 * (r + 1) * (k + 1)) + 0.1 * np.sin(0.7 * k) + 0.3 * np.sin(1.3 * r + 0.05 * k)
    return torch.from_numpy((x * scale).astype(np.float32))


def random_params(n_classes: int, seed: int, size_arg: str = "big", bias_scale: float = 0.05) -> Dict[str, torch.Tensor]:
    """Xavier-normal weights (utils/utils.py:150-154: std = sqrt(2 / (fan_in + fan_out))) and small non-zero biases from numpy's PCG64 stream
    (bit-stable across platforms). The closed-form sine weights make layer 2's pre-activations 15x smaller than Xavier's (std 0.065), i.e.
    15x more of them at round-off of zero; the second golden family uses these instead."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k, shp in param_shapes(n_classes, size_arg).items():
        if len(shp) == 2:
            out[k] = torch.from_numpy((rng.standard_normal(shp) * math.sqrt(2.0 / (shp[0] + shp[1]))).astype(np.float32))
        else:
            out[k] = torch.from_numpy((rng.standard_normal(shp) * bias_scale).astype(np.float32))
    return out


def random_bag(n: int, seed: int, l0: int = 1024) -> torch.Tensor:
    """N(0,1) bag [n, l0] from numpy's PCG64 stream (bit-stable across platforms and numpy versions, unlike torch's CPU generator across
    torch versions): the second golden family (synthetic bags are N(0,1)). Pre-activations of a random bag are spread
    continuously, so - unlike the closed-form cos/sin bags, which park some within fp32 round-off of zero - ReLU-mask flips between two
    correct fp32 implementations are rare (expected ~0.7 per 10^7 mask elements)."""
    return torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal((n, l0), dtype=np.float32))


# --------------------------------------------------------------------------------------
# Forward, op for op as the reference runs it
# --------------------------------------------------------------------------------------
def linear_act(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, act: str) -> torch.Tensor:
    """nn.Linear (+ReLU): models/model_toad.py:59,62 (trunk), :21,25 (attention_a/b pre-activations)."""
    y = torch.addmm(b, x, w.t())
    if act == "relu":
        y = torch.relu(y)
    elif act != "none":
        raise ValueError(act)
    return y


def gated_scores(pa: torch.Tensor, pb: torch.Tensor, wc: torch.Tensor, bc: torch.Tensor,
                 ma: Optional[torch.Tensor] = None, mb: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attn_Net_Gated.forward, models/model_toad.py:36-41: A = (tanh(pa) * sigmoid(pb)) Wc^T + bc -> [N,T].
    ma/mb: train-mode Dropout(0.25) multipliers (0 or 1/(1-p)) after Tanh / Sigmoid (:27-29), or None."""
    a, b = torch.tanh(pa), torch.sigmoid(pb)
    if ma is not None:
        a, b = a * ma, b * mb
    g = a.mul(b)
    return torch.addmm(bc, g, wc.t())


def softmax_pool(a_raw_nt: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """models/model_toad.py:92,97-98: A = softmax(A^T, dim=1); M = A @ h -> [T,L]."""
    a = torch.softmax(a_raw_nt.t(), dim=1)
    return torch.mm(a, h)


def gated_pool_fwd(pa, pb, h, wc, bc, ma=None, mb=None):
    """CPU twin of the fused HIP pool kernel: (A_raw[N,T], M[T,L])."""
    a_raw = gated_scores(pa, pb, wc, bc, ma, mb)
    return a_raw, softmax_pool(a_raw, h)


def heads_fwd(m_tl, sex, wcls, bcls, wsite, bsite):
    """models/model_toad.py:99-107: concat sex, two Linear heads, topk, softmax."""
    mcat = torch.cat([m_tl, sex.repeat(m_tl.size(0), 1)], dim=1)          # :99
    logits = torch.addmm(bcls, mcat[0].unsqueeze(0), wcls.t())              # :101
    y_hat = torch.topk(logits, 1, dim=1)[1]                                  # :102
    y_prob = torch.softmax(logits, dim=1)                                    # :103
    site_logits = torch.addmm(bsite, mcat[1].unsqueeze(0), wsite.t())       # :105
    site_hat = torch.topk(site_logits, 1, dim=1)[1]                          # :106
    site_prob = torch.softmax(site_logits, dim=1)                            # :107
    return mcat, logits, y_prob, y_hat, site_logits, site_prob, site_hat


@dataclass
class Saved:
    x: torch.Tensor
    h1: torch.Tensor
    h: torch.Tensor
    p: torch.Tensor          # [N, 2D] pre-activations (pa | pb)
    a_raw: torch.Tensor      # [N, T]
    m: torch.Tensor          # [T, L] pooled (before the sex concat)
    mcat: torch.Tensor       # [T, L+1]
    sex: torch.Tensor
    masks: Optional[Dict[str, torch.Tensor]] = None   # dropout multipliers {"h1","h","a","b"} (train + dropout=True)


def forward(params: Dict[str, torch.Tensor], x: torch.Tensor, sex: torch.Tensor,
            return_features: bool = False, attention_only: bool = False,
            masks: Optional[Dict[str, torch.Tensor]] = None):
    """TOAD_fc_mtl_concat.forward, models/model_toad.py:90-116. ``masks`` = explicit Dropout(0.25)
    multipliers (0 or 4/3) for the four dropout sites (:61,:64,:27-29); None = dropout off / eval."""
    p = params
    h1 = linear_act(x, p["attention_net.0.weight"], p["attention_net.0.bias"], "relu")      # :59
    if masks is not None:
        h1 = h1 * masks["h1"]                                                                # :61
    h = linear_act(h1, p["attention_net.2.weight"], p["attention_net.2.bias"], "relu")      # :62
    if masks is not None:
        h = h * masks["h"]                                                                   # :64
    wab = torch.cat([p["attention_net.4.attention_a.0.weight"], p["attention_net.4.attention_b.0.weight"]], 0)
    bab = torch.cat([p["attention_net.4.attention_a.0.bias"], p["attention_net.4.attention_b.0.bias"]], 0)
    pre = linear_act(h, wab, bab, "none")                                                    # :21,:25
    d = wab.shape[0] // 2
    a_raw_nt = gated_scores(pre[:, :d], pre[:, d:], p["attention_net.4.attention_c.weight"],
                            p["attention_net.4.attention_c.bias"],
                            None if masks is None else masks["a"], None if masks is None else masks["b"])   # :36-41
    a_tn = a_raw_nt.t()                                                                      # :92
    if attention_only:
        return a_tn[0]                                                                       # :93-94
    m = softmax_pool(a_raw_nt, h)                                                            # :97-98
    mcat, logits, y_prob, y_hat, site_logits, site_prob, site_hat = heads_fwd(
        m, sex, p["classifier.weight"], p["classifier.bias"],
        p["site_classifier.weight"], p["site_classifier.bias"])
    out = {}
    if return_features:
        out["features"] = mcat                                                               # :110-111
    out.update({"logits": logits, "Y_prob": y_prob, "Y_hat": y_hat, "site_logits": site_logits,
                "site_prob": site_prob, "site_hat": site_hat, "A": a_tn})                     # :113-114
    saved = Saved(x=x, h1=h1, h=h, p=pre, a_raw=a_raw_nt, m=m, mcat=mcat, sex=sex, masks=masks)
    return out, saved


def loss_fn(logits, label, site_logits, site):
    """utils/core_utils_mtl_concat.py:111,213-215: 0.75*CE(logits,label)+0.25*CE(site_logits,site)."""
    ce = torch.nn.functional.cross_entropy
    return 0.75 * ce(logits, label) + 0.25 * ce(site_logits, site)


def loss_grad(logits, label, site_logits, site):
    """d loss / d logits for the weighted CE above (softmax - onehot, batch of one)."""
    def one(lg, y, wgt):
        g = torch.softmax(lg, dim=1).clone()
        g[0, int(y)] -= 1.0
        return g * wgt
    return one(logits, label, 0.75), one(site_logits, site, 0.25)


# --------------------------------------------------------------------------------------
# Hand-written backward, decomposed like the HIP kernels (autograd mirror of the path;
# reference: implicit via loss.backward(), utils/core_utils_mtl_concat.py:231)
# --------------------------------------------------------------------------------------
def heads_bwd(mcat, dlogits, dsite, wcls, wsite):
    """Backward of models/model_toad.py:99-105. Returns dWcls,dbcls,dWsite,dbsite,dM[T,L]."""
    l = mcat.shape[1] - 1
    dwcls = dlogits.t() @ mcat[0:1]
    dwsite = dsite.t() @ mcat[1:2]
    dm = torch.stack([(dlogits @ wcls)[0, :l], (dsite @ wsite)[0, :l]], 0)
    return dwcls, dlogits[0].clone(), dwsite, dsite[0].clone(), dm


def gated_pool_bwd(pa, pb, h, wc, a_raw, m, dm, da_ext: Optional[torch.Tensor] = None, ma=None, mb=None):
    """Backward of models/model_toad.py:36-41,92-98.

    dS[i,t] = p[i,t] * (dM[t].H[i] - dM[t].M[t]) (+ external dA_raw), p = softmax over i.
    Returns dPa, dPb, dH_pool, dWc, dbc.
    """
    p = torch.softmax(a_raw.t(), dim=1).t()                 # [N,T]
    c = (dm * m).sum(dim=1)                                  # [T]
    ds = p * (h @ dm.t() - c[None, :])                       # [N,T]
    if da_ext is not None:
        ds = ds + da_ext
    dh = p @ dm                                              # [N,L]
    a = torch.tanh(pa)
    b = torch.sigmoid(pb)
    ka = 1.0 if ma is None else ma                           # dropout multipliers after tanh / sigmoid
    kb = 1.0 if mb is None else mb
    ad, bd = a * ka, b * kb
    g = ad * bd
    dg = ds @ wc                                             # [N,D]
    dpa = dg * bd * ka * (1.0 - a * a)
    dpb = dg * ad * kb * b * (1.0 - b)
    dwc = ds.t() @ g
    dbc = ds.sum(dim=0)
    return dpa, dpb, dh, dwc, dbc


def linear_bwd(x, w, y_act, dy, act: str, dx_add: Optional[torch.Tensor] = None, need_dx: bool = True,
               mask_scale: float = 1.0):
    """Backward of y = dropout(act(x W^T + b)). ``dy`` is the grad wrt the saved output y_act.

    Returns (dx, dW, db).  For relu the mask is y_act > 0 (same as torch's threshold_backward); with
    dropout y_act's zeros already include the dropped elements and kept ones carry ``mask_scale`` = 1/(1-p).
    """
    if act == "relu":
        dy = dy * (y_act > 0).to(dy.dtype) * mask_scale
    dw = dy.t() @ x
    db = dy.sum(dim=0)
    dx = dy @ w if need_dx else None
    if dx is not None and dx_add is not None:
        dx = dx + dx_add
    return dx, dw, db


def backward(params: Dict[str, torch.Tensor], s: Saved, dlogits, dsite,
             da_ext: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Full backward in kernel order; returns {state-dict key: grad}."""
    p = params
    d = p["attention_net.4.attention_a.0.weight"].shape[0]
    wab = torch.cat([p["attention_net.4.attention_a.0.weight"], p["attention_net.4.attention_b.0.weight"]], 0)
    g: Dict[str, torch.Tensor] = {}
    dwcls, dbcls, dwsite, dbsite, dm = heads_bwd(s.mcat, dlogits, dsite, p["classifier.weight"],
                                                 p["site_classifier.weight"])
    g["classifier.weight"], g["classifier.bias"] = dwcls, dbcls
    g["site_classifier.weight"], g["site_classifier.bias"] = dwsite, dbsite
    mk = s.masks
    msc = 1.0 if mk is None else 1.0 / 0.75
    dpa, dpb, dh_pool, dwc, dbc = gated_pool_bwd(s.p[:, :d], s.p[:, d:], s.h,
                                                 p["attention_net.4.attention_c.weight"], s.a_raw, s.m, dm, da_ext,
                                                 None if mk is None else mk["a"], None if mk is None else mk["b"])
    g["attention_net.4.attention_c.weight"], g["attention_net.4.attention_c.bias"] = dwc, dbc
    dp = torch.cat([dpa, dpb], 1)
    # attention_a/b Linear: dH = dP Wab + dH_pool ; dWab = dP^T H
    dh, dwab, dbab = linear_bwd(s.h, wab, s.p, dp, "none", dx_add=dh_pool)
    g["attention_net.4.attention_a.0.weight"], g["attention_net.4.attention_b.0.weight"] = dwab[:d], dwab[d:]
    g["attention_net.4.attention_a.0.bias"], g["attention_net.4.attention_b.0.bias"] = dbab[:d], dbab[d:]
    dh1, dw2, db2 = linear_bwd(s.h1, p["attention_net.2.weight"], s.h, dh, "relu", mask_scale=msc)
    g["attention_net.2.weight"], g["attention_net.2.bias"] = dw2, db2
    _, dw1, db1 = linear_bwd(s.x, p["attention_net.0.weight"], s.h1, dh1, "relu", need_dx=False, mask_scale=msc)
    g["attention_net.0.weight"], g["attention_net.0.bias"] = dw1, db1
    return g


def fwd_bwd(params, x, sex, label, site, masks=None):
    """One training-step's worth of math: forward, weighted CE, backward. Returns (out, loss, grads)."""
    out, saved = forward(params, x, sex, masks=masks)
    loss = loss_fn(out["logits"], label, out["site_logits"], site)
    dlogits, dsite = loss_grad(out["logits"], label, out["site_logits"], site)
    grads = backward(params, saved, dlogits, dsite)
    return out, loss, grads


def xavier_params(n_classes: int, seed: int = 1, size_arg: str = "big") -> Dict[str, torch.Tensor]:
    """utils/utils.py:150-154 initialize_weights: xavier_normal_ weights, zero biases (seeded)."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(n_classes, size_arg).items():
        if len(shp) == 2:
            std = math.sqrt(2.0 / (shp[0] + shp[1]))
            out[k] = torch.randn(shp, generator=gen) * std
        else:
            out[k] = torch.zeros(shp)
    return out


def xavier_params_like_reference(n_classes: int, seed: int = 1, size_arg: str = "big") -> Dict[str, torch.Tensor]:
    """The parameters the reference model starts from under ``torch.manual_seed(seed)``, bit for bit (checked against the imported
    reference by oracle/pin_config1_against_reference.py): models/model_toad.py:54-75 builds its nn.Linear layers in this order - each
    constructor draws its default kaiming-uniform weight and uniform bias from the global generator - and then
    utils/utils.py:150-154 initialize_weights re-draws every weight with xavier_normal_ in ``modules()`` order and zeroes the biases."""
    l0, l, d = {"small": (1024, 512, 256), "big": (1024, 512, 384)}[size_arg]
    state = torch.get_rng_state()
    try:
        torch.manual_seed(seed)
        layers = [("attention_net.0", torch.nn.Linear(l0, l)), ("attention_net.2", torch.nn.Linear(l, l)),
                  ("attention_net.4.attention_a.0", torch.nn.Linear(l, d)), ("attention_net.4.attention_b.0", torch.nn.Linear(l, d)),
                  ("attention_net.4.attention_c", torch.nn.Linear(d, 2)), ("classifier", torch.nn.Linear(l + 1, n_classes)),
                  ("site_classifier", torch.nn.Linear(l + 1, 2))]
        out = {}
        for name, lin in layers:
            torch.nn.init.xavier_normal_(lin.weight)
            out[name + ".weight"] = lin.weight.detach().clone()
            out[name + ".bias"] = torch.zeros_like(lin.bias)
    finally:
        torch.set_rng_state(state)
    assert set(out) == set(PARAM_KEYS)
    return out

