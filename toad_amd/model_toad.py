"""MI355X-native drop-in for the reference's ``models/model_toad.py``.

Same public surface as the reference (so ``utils/core_utils_mtl_concat.py`` /
``utils/eval_utils_mtl_concat.py`` can ``from models.model_toad import TOAD_fc_mtl_concat``
unchanged — INTEGRATION.md):

  * ``Attn_Net_Gated(L=1024, D=256, dropout=False, n_tasks=1)``          (model_toad.py:17-41)
  * ``TOAD_fc_mtl_concat(gate=True, size_arg="big", dropout=False, n_classes=2)`` with
    ``relocate()`` and ``forward(h, sex, return_features=False, attention_only=False)``
    returning the same result dict                                         (model_toad.py:53-116)
  * identical sub-module layout, hence identical ``state_dict()`` keys/shapes
  * Xavier-normal weights / zero biases at construction                    (utils/utils.py:150-154)

What differs is everything underneath: ``forward`` never calls the nn.Linear/Tanh/Sigmoid
sub-modules (they only own the parameters); it runs hand-written gfx950 kernels from
libtoad_hip.so through ``toad_amd.functional``.  There is no CPU path — CPU tensors raise.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import functional as F_
from . import ops

_ALIGN = 64  # floats (256 B): every parameter starts 256-B aligned inside the flat buffer
_M64 = 0xFFFFFFFFFFFFFFFF


def initialize_weights(module: nn.Module) -> None:
    """Same initialisation as the reference's utils/utils.py:150-154."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight)
            m.bias.data.zero_()


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on {t.device}: toad_amd runs on MI355X (HIP) only and has no CPU fallback. "
            "Call model.relocate() and move the inputs to the GPU.")


def _draw_dropout(active: bool):
    """(drop_p, seed) for one forward. Train-mode Dropout(0.25) masks are a stateless hash of (seed, element
    index) inside the kernels; the seed is drawn from torch's CPU generator, so torch.manual_seed /
    seed_torch (main_mtl_concat.py:109-119) make runs reproducible. No host-device sync is involved."""
    if not active:
        return 0.0, 0
    return F_.DROP_P, int(torch.randint(0, 2 ** 62, (1,)).item())


def _scores_blocks(d: int, t: int):
    """(task blocks of <= 4, column blocks of <= 512) the pool kernels' covering instantiation takes (csrc/gated_pool.hip: shape_ok)."""
    return [(t0, min(t0 + 4, t)) for t0 in range(0, t, 4)], [(d0, min(d0 + 512, d)) for d0 in range(0, d, 512)]


class _ScoresFn(torch.autograd.Function):
    """Standalone Attn_Net_Gated: A = (tanh(xWa^T+ba) * sigmoid(xWb^T+bb)) Wc^T + bc, for ANY (L, D, n_tasks) the reference's constructor takes
    (models/model_toad.py:19). Within the kernels' envelope (L % 8 == 0, D <= 512 with D % 4 == 0, n_tasks <= 4: every shape TOAD itself builds)
    it is one GEMM + one scores launch. Outside it the same kernels run over blocks: the scores are linear in the gate columns and independent
    per task, so D is cut into column blocks of <= 512 (summed), n_tasks into blocks of <= 4 (concatenated); a D that is not a multiple of 4 /
    an L that is not a multiple of 8 is zero-padded (a zero weight row gives tanh(0) * sigmoid(0) = 0: exact)."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb, wc, bc, drop_p, seed):
        d, l = wa.shape
        t = wc.shape[0]
        dp_, lp_ = (d + 3) // 4 * 4, (l + 7) // 8 * 8
        if dp_ != d or lp_ != l:                                   # zero padding (rare shapes): extra weight rows / input columns that contribute nothing
            pad_w = lambda w: torch.nn.functional.pad(w, (0, lp_ - l, 0, dp_ - d))     # noqa: E731
            wa_, wb_ = pad_w(wa), pad_w(wb)
            ba_, bb_ = torch.nn.functional.pad(ba, (0, dp_ - d)), torch.nn.functional.pad(bb, (0, dp_ - d))
            wc_ = torch.nn.functional.pad(wc, (0, dp_ - d))
            xp = torch.nn.functional.pad(x, (0, lp_ - l)) if lp_ != l else x
        else:
            wa_, wb_, ba_, bb_, wc_, xp = wa, wb, ba, bb, wc, x
        wab, bab = torch.cat([wa_, wb_], 0), torch.cat([ba_, bb_], 0)
        p = ops.linear_act_fwd(xp.contiguous(), wab, bab, ops.ACT_NONE)
        _, _, sa, sb = F_.drop_seeds(seed)
        tb, db = _scores_blocks(dp_, t)
        if len(tb) == 1 and len(db) == 1:
            a_raw, _, _ = ops.gated_pool_fwd(p, dp_, None, wc_.contiguous(), bc, drop_p, sa, sb)
        else:
            a_raw = torch.empty((x.shape[0], t), dtype=torch.float32, device=x.device)
            zero_b = torch.zeros(4, dtype=torch.float32, device=x.device)
            for (t0, t1) in tb:
                acc = None
                for j, (d0, d1) in enumerate(db):
                    blk = ops.gate_scores_block_fwd(p, dp_, d0, wc_[t0:t1, d0:d1].contiguous(), bc[t0:t1].contiguous() if j == 0 else zero_b[:t1 - t0],
                                                    drop_p, sa + 2 * j * F_._GOLDEN & _M64, sb + 2 * j * F_._GOLDEN & _M64)
                    # (a mask stream per column block AND branch: sb = sa + G, so the blocks step by 2 G like the slides of a batch do in
                    #  csrc/step.hip - with a step of G block j + 1's tanh masks were block j's sigmoid masks)
                    acc = blk if acc is None else acc.add_(blk)
                a_raw[:, t0:t1] = acc
        ctx.save_for_backward(xp, p, wab, wc_)
        ctx.need_dx = x.requires_grad
        ctx.drop = (drop_p, sa, sb)
        ctx.dims = (d, l, dp_, lp_, t)
        return a_raw

    @staticmethod
    def backward(ctx, da):
        xp, p, wab, wc_ = ctx.saved_tensors
        d, l, dp_, lp_, t = ctx.dims
        n = xp.shape[0]
        drop_p, sa, sb = ctx.drop
        da = da.contiguous()
        tb, db = _scores_blocks(dp_, t)
        dp = torch.empty_like(p)
        dwc = torch.empty_like(wc_)
        dbc = torch.empty((t,), dtype=torch.float32, device=p.device)
        for j, (d0, d1) in enumerate(db):
            for i, (t0, t1) in enumerate(tb):
                dpa, dpb, dwc_blk, dbc_blk = ops.gate_scores_block_bwd(p, dp_, d0, wc_[t0:t1, d0:d1].contiguous(), da[:, t0:t1].contiguous(), drop_p,
                                                                       sa + 2 * j * F_._GOLDEN & _M64, sb + 2 * j * F_._GOLDEN & _M64)
                if i == 0:                                         # dP is linear in dA: the task blocks of one column block add up
                    dp[:, d0:d1], dp[:, dp_ + d0:dp_ + d1] = dpa, dpb
                else:
                    dp[:, d0:d1] += dpa; dp[:, dp_ + d0:dp_ + d1] += dpb
                dwc[t0:t1, d0:d1] = dwc_blk
                if j == 0:
                    dbc[t0:t1] = dbc_blk                           # column sums of dA: the same from every column block
        dwab, dbab = ops.linear_wgrad(dp, xp)
        dx = ops.linear_dgrad(dp, ops.transpose(wab)) if ctx.need_dx else None
        if dx is not None and lp_ != l:
            dx = dx[:, :l].contiguous()
        return (dx, dwab[:d, :l], dbab[:d], dwab[dp_:dp_ + d, :l], dbab[dp_:dp_ + d], dwc[:, :d], dbc, None, None)


class Attn_Net_Gated(nn.Module):
    """Attention network with sigmoid gating (3 fc layers) — reference models/model_toad.py:17-41."""

    def __init__(self, L: int = 1024, D: int = 256, dropout: bool = False, n_tasks: int = 1):
        super().__init__()
        a = [nn.Linear(L, D), nn.Tanh()]
        b = [nn.Linear(L, D), nn.Sigmoid()]
        if dropout:
            a.append(nn.Dropout(0.25))
            b.append(nn.Dropout(0.25))
        self.attention_a = nn.Sequential(*a)
        self.attention_b = nn.Sequential(*b)
        self.attention_c = nn.Linear(D, n_tasks)
        self._dropout = bool(dropout)

    def forward(self, x):
        _require_cuda(x, "x")
        drop_p, seed = _draw_dropout(self._dropout and self.training)
        A = _ScoresFn.apply(x.contiguous(), self.attention_a[0].weight, self.attention_a[0].bias,
                            self.attention_b[0].weight, self.attention_b[0].bias,
                            self.attention_c.weight, self.attention_c.bias, drop_p, seed)
        return A, x


class TOAD_fc_mtl_concat(nn.Module):
    """TOAD multi-task + concat MIL network with attention pooling — reference models/model_toad.py:53-116."""

    def __init__(self, gate: bool = True, size_arg: str = "big", dropout: bool = False, n_classes: int = 2):
        super().__init__()
        self.size_dict = {"small": [1024, 512, 256], "big": [1024, 512, 384]}
        size = self.size_dict[size_arg]
        if not gate:
            # the reference refers to an undefined Attn_Net here (model_toad.py:68 -> NameError)
            raise NameError("name 'Attn_Net' is not defined (gate=False is not supported by the reference either)")
        fc = [nn.Linear(size[0], size[1]), nn.ReLU()]
        if dropout:
            fc.append(nn.Dropout(0.25))
        fc.extend([nn.Linear(size[1], size[1]), nn.ReLU()])
        if dropout:
            fc.append(nn.Dropout(0.25))
        fc.append(Attn_Net_Gated(L=size[1], D=size[2], dropout=dropout, n_tasks=2))
        self.attention_net = nn.Sequential(*fc)
        self.classifier = nn.Linear(size[1] + 1, n_classes)
        self.site_classifier = nn.Linear(size[1] + 1, 2)
        initialize_weights(self)
        self._dropout = bool(dropout)
        self._idx2 = 3 if dropout else 2          # position of the second Linear inside attention_net
        self._flat: Optional[torch.Tensor] = None
        self._views: Dict[str, torch.Tensor] = {}

    # ---- parameter plumbing ---------------------------------------------------------------
    def _slot_params(self) -> Dict[str, nn.Parameter]:
        net = self.attention_net
        att = net[len(net) - 1]
        return {
            "w1": net[0].weight, "b1": net[0].bias,
            "w2": net[self._idx2].weight, "b2": net[self._idx2].bias,
            "wa": att.attention_a[0].weight, "ba": att.attention_a[0].bias,
            "wb": att.attention_b[0].weight, "bb": att.attention_b[0].bias,
            "wc": att.attention_c.weight, "bc": att.attention_c.bias,
            "wcls": self.classifier.weight, "bcls": self.classifier.bias,
            "wsite": self.site_classifier.weight, "bsite": self.site_classifier.bias,
        }

    # flat layout: Wa|Wb and ba|bb adjacent so the stacked 512->768 GEMM is zero-copy
    _FLAT_ORDER = ("w1", "b1", "w2", "b2", "wa", "wb", "ba", "bb", "wc", "bc", "wcls", "bcls", "wsite", "bsite")

    def _flat_layout(self):
        sp = self._slot_params()
        offs, off = {}, 0
        for k in self._FLAT_ORDER:
            offs[k] = off
            n = sp[k].numel()
            adjacent_next = k in ("wa", "ba")        # keep the b-half glued to the a-half
            off += n if adjacent_next else (n + _ALIGN - 1) // _ALIGN * _ALIGN
        return sp, offs, off

    def _is_flat(self) -> bool:
        if self._flat is None:
            return False
        sp, offs, _ = self._flat_layout()
        base = self._flat.data_ptr()
        return all(p.data_ptr() == base + 4 * offs[k] and p.device == self._flat.device for k, p in sp.items())

    def flatten_parameters(self) -> torch.Tensor:
        """Re-home every parameter as a view of one contiguous fp32 buffer (values preserved).
        Makes [Wa;Wb] a zero-copy view and gives data-parallel training one all-reduce bucket."""
        sp, offs, total = self._flat_layout()
        dev = sp["w1"].device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for k, p in sp.items():
                v = flat[offs[k]: offs[k] + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
        self._flat = flat
        d, l = sp["wa"].shape
        self._views = {
            "wab": flat[offs["wa"]: offs["wa"] + 2 * d * l].view(2 * d, l),
            "bab": flat[offs["ba"]: offs["ba"] + 2 * d],
        }
        return flat

    def flat_parameters(self) -> torch.Tensor:
        if not self._is_flat():
            self.flatten_parameters()
        return self._flat

    def flat_offsets(self):
        sp, offs, total = self._flat_layout()
        return {k: (offs[k], sp[k].numel()) for k in sp}, total

    def _weights(self) -> Dict[str, torch.Tensor]:
        """slot -> tensor for the library calls (the Parameter objects themselves + the stacked [Wa;Wb] / [ba;bb] views of the flat buffer).
        Called once per forward: walking the module tree through nn.Module.__getattr__ and re-deriving the flat layout cost 75 us per call, a
        tenth of the host time of the reference's loop on a 10k-patch bag (tools/dropin_prof.py; the loop is host-bound there). The dict is
        therefore cached and re-validated per call by identity: every (sub)module is still the registered child of its parent, every Parameter
        object is still registered under its name, and its storage still sits at its offset in the flat buffer - ~30 dict lookups and 15
        data_ptr() calls. Anything else (model.to(), load into new tensors, a replaced layer) takes the slow path and re-flattens."""
        c = self.__dict__.get("_w_cache")
        if c is not None:
            w, flat, base, mods, prm = c
            if self._flat is flat and flat.data_ptr() == base and all(par.get(key) is child for par, key, child in mods) \
                    and all(reg.get(name) is q and q.data_ptr() == ptr for reg, name, q, ptr in prm):
                return dict(w)
        if not self._is_flat():
            self.flatten_parameters()
        w: Dict[str, torch.Tensor] = dict(self._slot_params())
        w.update(self._views)
        # what the fast path re-validates: the module chain down to every Linear, and every parameter's registration + address
        net = self.attention_net
        att = net[len(net) - 1]
        lin = {"w1": net[0], "b1": net[0], "w2": net[self._idx2], "b2": net[self._idx2], "wa": att.attention_a[0], "ba": att.attention_a[0],
               "wb": att.attention_b[0], "bb": att.attention_b[0], "wc": att.attention_c, "bc": att.attention_c,
               "wcls": self.classifier, "bcls": self.classifier, "wsite": self.site_classifier, "bsite": self.site_classifier}
        mods = [(self._modules, "attention_net", net), (self._modules, "classifier", self.classifier), (self._modules, "site_classifier", self.site_classifier),
                (net._modules, "0", net[0]), (net._modules, str(self._idx2), net[self._idx2]), (net._modules, str(len(net) - 1), att),
                (att._modules, "attention_a", att.attention_a), (att._modules, "attention_b", att.attention_b), (att._modules, "attention_c", att.attention_c),
                (att.attention_a._modules, "0", att.attention_a[0]), (att.attention_b._modules, "0", att.attention_b[0])]
        prm = [(lin[k]._parameters, "weight" if k.startswith("w") else "bias", w[k], w[k].data_ptr()) for k in self._FLAT_ORDER]
        self.__dict__["_w_cache"] = (dict(w), self._flat, self._flat.data_ptr(), mods, prm)
        return w

    # ---- checkpoints ----------------------------------------------------------------------
    _DP_INFIX = "attention_net.module."

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Checkpoints written by a multi-GPU REFERENCE run carry nn.DataParallel's ``module.`` infix
        (models/model_toad.py:79-81 wraps attention_net whenever device_count() > 1, so its keys are
        ``attention_net.module.0.weight`` ...). This build never wraps, so those keys are renamed on load;
        without it eval_utils' non-strict load (eval_utils_mtl_concat.py:28-29) would silently leave the
        whole attention branch at its random initialisation."""
        infix = prefix + self._DP_INFIX
        for k in [k for k in state_dict if k.startswith(infix)]:
            state_dict[prefix + "attention_net." + k[len(infix):]] = state_dict.pop(k)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ---- reference API --------------------------------------------------------------------
    def relocate(self):
        """Reference: models/model_toad.py:77-88. Places the model on the current HIP device.
        Never wraps in nn.DataParallel: one process drives one GPU; slides, not patches, are
        sharded across GPUs (toad_amd.dp)."""
        if not torch.cuda.is_available():
            raise RuntimeError("toad_amd.relocate(): no HIP device visible; this package has no CPU fallback")
        device = torch.device("cuda", torch.cuda.current_device())
        self.to(device)
        self.flatten_parameters()

    @staticmethod
    def _bag_dtype(h: torch.Tensor) -> torch.Tensor:
        """fp32 bags pass; fp16 bags (feature stores kept in half precision) go to the kernels as they are when the whole-slide
        fp16 entry points take the shape (toad_mil_*_x16_f32: same results as the up-cast bag, two MFMA terms instead of three in the
        first layer); everything else (bf16, fp64, empty or oversized fp16 bags) is up-cast to fp32 like nn.Linear's caller would."""
        if getattr(h, "is_prepared_bag", False):                   # ops.prepare_bag: already in the form the first GEMMs consume
            if h.shape[1] != 1024 or not ops.x16_ok(h.shape[0]):
                raise ValueError("a PreparedBag must be [N, 1024] with 64 <= N and N * 4096 < 2^32")
            return h
        if h.dtype == torch.float32:
            return h
        if h.dtype == torch.float16 and h.dim() == 2 and h.shape[1] == 1024 and not h.requires_grad and ops.x16_ok(h.shape[0]):
            return h
        return h.float()

    def forward(self, h, sex, return_features=False, attention_only=False):
        _require_cuda(h, "h")
        drop_p, seed = _draw_dropout(self._dropout and self.training)
        w = self._weights()
        _require_cuda(w["w1"], "model parameters")
        h = self._bag_dtype(h.contiguous())
        if attention_only:
            with torch.no_grad():
                wd = {k: v.detach() for k, v in w.items()}
                if h.shape[0] == 0:
                    a_raw = F_.attention_scores(wd, h, drop_p, seed)
                else:                                             # trunk + scores in one library call, no pooling / heads
                    # cached arena + a copy of the scores: a view would pin the whole arena (7 KB per patch) behind an [N] result
                    a_raw = ops.mil_fwd(wd, h, None, drop_p, seed, attention_only=True, cached_arena=True).view("a_raw", (h.shape[0], 2))
                    return a_raw[:, 0].clone()                    # model_toad.py:92-94: raw task-0 scores, [N]
            return a_raw.t()[0]
        _require_cuda(sex, "sex")
        sex = sex.to(torch.float32).reshape(1).contiguous()
        sp = [w[k] for k in F_.SLOTS]
        need_grad = torch.is_grad_enabled() and (h.requires_grad or sex.requires_grad or any(p.requires_grad for p in sp))
        if not need_grad and h.shape[0] > 0:
            # no backward will follow (eval / no_grad / frozen model): the saved activations (H1, H, P: ~7 KB per patch) go to a cached
            # arena that the next such forward reuses, and the small outputs are COPIED out - a view would keep 0.7 GB alive per
            # 100k-patch slide for as long as a caller holds Y_prob (validate / summary append them for a whole epoch)
            wd = {k: v.detach() for k, v in w.items()}
            arena = ops.mil_fwd(wd, h, sex, drop_p, seed, cached_arena=True)
            n, c = h.shape[0], wd["wcls"].shape[0]
            v = arena.view
            small = torch.cat([v("logits", (c,)), v("y_prob", (c,)), v("site_logits", (2,)), v("site_prob", (2,)), v("mcat", (2 * 513,))])  # one small copy
            logits, y_prob, site_logits, site_prob, feats = small[:c].view(1, c), small[c:2 * c].view(1, c), small[2 * c:2 * c + 2].view(1, 2), \
                small[2 * c + 2:2 * c + 4].view(1, 2), small[2 * c + 4:].view(2, 513)
            hats = torch.cat([v("y_hat", (1,), torch.int64), v("site_hat", (1,), torch.int64)])
            y_hat, site_hat = hats[0:1].view(1, 1), hats[1:2].view(1, 1)
            a_nt = v("a_raw", (n, 2)).clone()
        else:
            logits, site_logits, a_nt, feats, y_prob, y_hat, site_prob, site_hat = F_.ToadMIL.apply(
                h, sex, *sp, w["wab"], w["bab"], drop_p, seed)
        results_dict = {}
        if return_features:
            results_dict.update({"features": feats})              # M after the sex concat, [2, L+1]
        results_dict.update({"logits": logits, "Y_prob": y_prob, "Y_hat": y_hat,
                             "site_logits": site_logits, "site_prob": site_prob, "site_hat": site_hat,
                             "A": a_nt.t()})                      # pre-softmax scores, [2, N] (transpose view)
        return results_dict

    @torch.no_grad()
    def forward_many(self, bags, sexes, return_features=False):
        """Forward-only pass over SEVERAL slides of different lengths (no reference counterpart: the reference's validate /
        summary loops call ``model(data, sex)`` once per slide with batch size 1, utils/core_utils_mtl_concat.py:284,393 and
        utils/eval_utils_mtl_concat.py:91). The rows of a bag never interact before the pooling, so the ragged batch is
        concatenated to one [sum N, 1024] operand and the three trunk / attention GEMMs (99 % of the flops and most of the
        launches of a small slide) run ONCE over it; the softmax pooling and the two heads then run per slide on row
        ranges of the shared activations. Returns one result dict per slide with the keys of ``forward``.

        Eval semantics only (dropout is not applied; gradients are not recorded). The GEMM operand scales are taken per
        256-row block of the CONCATENATED operand, so the values agree with the one-slide path to fp32 round-off, not bitwise."""
        if self.training and self._dropout:
            raise RuntimeError("forward_many is an inference path: call model.eval() first (dropout would be skipped)")
        if any(getattr(b, "is_prepared_bag", False) for b in bags):
            raise TypeError("forward_many concatenates fp32 bags; pass the tensors, not PreparedBag objects")
        bags = [b.contiguous() if b.dtype == torch.float32 else b.float().contiguous() for b in bags]   # (the per-op GEMMs of this path are fp32-only)
        if len(bags) != len(sexes):
            raise ValueError("forward_many: one sex entry per bag")
        if not bags:
            return []
        for b in bags:
            _require_cuda(b, "bag")
            if b.dim() != 2 or b.shape[0] == 0 or b.shape[1] != self.size_dict["big"][0]:
                raise ValueError("forward_many: every bag must be a non-empty [N, 1024] tensor")
        w = {k: v.detach() for k, v in self._weights().items()}
        d = w["wc"].shape[1]
        x = bags[0] if len(bags) == 1 else torch.cat(bags, 0)
        h1, a1 = ops.linear_act_fwd(x, w["w1"], w["b1"], ops.ACT_RELU, want_amax=True)
        h, a2 = ops.linear_act_fwd(h1, w["w2"], w["b2"], ops.ACT_RELU, x_amax=a1, want_amax=True)
        del h1
        p = ops.linear_act_fwd(h, w["wab"], w["bab"], ops.ACT_NONE, x_amax=a2)
        out, off = [], 0
        for b, sex in zip(bags, sexes):
            n = b.shape[0]
            a_raw, m, _ = ops.gated_pool_fwd(p[off:off + n], d, h[off:off + n], w["wc"], w["bc"])
            _require_cuda(sex, "sex")
            mcat, logits, y_prob, y_hat, s_logits, s_prob, s_hat = ops.heads_fwd(
                m, sex.to(torch.float32).reshape(1).contiguous(), w["wcls"], w["bcls"], w["wsite"], w["bsite"])
            res = {}
            if return_features:
                res["features"] = mcat
            res.update({"logits": logits, "Y_prob": y_prob, "Y_hat": y_hat, "site_logits": s_logits, "site_prob": s_prob,
                        "site_hat": s_hat, "A": a_raw.t()})
            out.append(res)
            off += n
        return out
