"""Slide-sharded data parallelism: one process per GPU, slides dealt across ranks, ONE all-reduce.

The reference's only multi-GPU code is nn.DataParallel *inside* a bag (models/model_toad.py:79-81:
scatter the N x 1024 bag, gather A and h on cuda:0).  That is not reproduced: one MI355X holds and
saturates on a whole 100k-patch bag, and slides are independent.  Instead:

  * every rank owns a full replica whose parameters are views of ONE flat fp32 buffer
    (TOAD_fc_mtl_concat.flat_parameters) and whose gradients are views of one flat grad buffer;
  * a step = each rank runs forward+loss+backward for its slides, the kernels accumulate
    (beta = 1) straight into the flat gradient; then a single all-reduce(SUM) of that 4.77 MB
    bucket over RCCL/xGMI, scale by 1/global_slides, identical optimiser step on every rank.

Semantic note: the reference steps once per slide; DP steps once per global batch
with the mean gradient.  Parity is therefore defined on gradients (tests/test_dp_gloo.py).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Slide = Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]   # (bag [N,1024], sex [1], label [1], site [1])


def shard_round_robin(n_slides: int, rank: int, world: int) -> List[int]:
    """slide i -> rank i mod world (BASELINE config 4)."""
    return list(range(rank, n_slides, world))


def shard_by_length(lengths: Sequence[int], rank: int, world: int) -> List[int]:
    """Longest-processing-time greedy: cost of a slide is proportional to its patch count."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    mine: List[int] = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += lengths[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


def hip_slide_grad(model, grads: Dict[str, torch.Tensor], slide: Slide, beta: float = 1.0, scale: float = 1.0,
                   w_cls: float = 0.75, w_site: float = 0.25):
    """forward + weighted CE (utils/core_utils_mtl_concat.py:213-215) + backward for one slide on the
    HIP kernels: grads = beta*grads + scale * d(loss)/d(params). ``scale`` (1/global_slides) is folded
    into the CE weights, so no separate gradient-scaling kernel runs. Returns the device loss vector
    [3] = (scale*loss, cls CE, site CE)."""
    from . import ops
    from .model_toad import _draw_dropout
    bag, sex, label, site = slide
    bag = model._bag_dtype(bag.contiguous())                  # fp32, or fp16 as stored (toad_mil_step_x16_f32)
    w = {k: v.detach() for k, v in model._weights().items()}
    drop_p, seed = _draw_dropout(model._dropout and model.training)
    if bag.shape[0] == 0:
        # empty bag (reference semantics: zero pooled features, the heads still see `sex`): per-op path, no trunk launches
        from . import functional as F_
        sexf = sex.to(torch.float32).reshape(1)
        outs, sv = F_.mil_forward(w, bag, sexf)
        loss, dl, ds = ops.mtl_ce_fwd_bwd(outs["logits"], outs["site_logits"], label, site, w_cls * scale, w_site * scale)
        F_.mil_backward(w, sv, dl, ds, grads=grads, beta=beta)
        return loss
    loss, _, _ = ops.mil_step(w, grads, beta, bag, sex.to(torch.float32).reshape(1), label, site,
                              w_cls * scale, w_site * scale, drop_p, seed)      # one C-ABI call per slide
    return loss


def hip_batch_grad(model, grads: Dict[str, torch.Tensor], slides: Sequence[Slide], beta: float = 1.0, scale: float = 1.0,
                   w_cls: float = 0.75, w_site: float = 0.25, xcat: Optional[torch.Tensor] = None, offsets=None):
    """One library call for a whole shard of (small) slides: the trunk / attention GEMMs of forward and backward run once over the
    concatenated bags, pooling / heads / loss per slide (toad_mil_multi_step_f32). grads = beta*grads + scale * sum_b d loss_b.
    ``xcat`` / ``offsets``: the bags already concatenated on the device (an ingest buffer); else they are concatenated here.
    Returns the per-slide loss vectors [B, 3]."""
    from . import ops
    from .model_toad import _draw_dropout
    w = {k: v.detach() for k, v in model._weights().items()}
    drop_p, seed = _draw_dropout(model._dropout and model.training)
    sex = torch.cat([s[1].to(torch.float32).reshape(1) for s in slides])
    label = torch.cat([s[2].reshape(1) for s in slides])
    site = torch.cat([s[3].reshape(1) for s in slides])
    if xcat is None:
        bags = [s[0].float() if s[0].dtype != torch.float32 else s[0] for s in slides]
        loss, _, _ = ops.mil_multi_step(w, grads, beta, bags, sex, label, site, w_cls * scale, w_site * scale, drop_p, seed)
    else:
        loss, _, _ = ops.mil_multi_step(w, grads, beta, xcat, sex, label, site, w_cls * scale, w_site * scale, drop_p, seed, offsets=offsets)
    return loss


class SlideShardedDP:
    def __init__(self, model, optimizer_factory: Callable[[Sequence[torch.nn.Parameter]], torch.optim.Optimizer],
                 process_group=None, slide_grad_fn: Optional[Callable] = None, broadcast_from: int = 0,
                 always_reduce: bool = False, batch_max_patches: Optional[int] = None, batch_rows: Optional[int] = None):
        """``batch_max_patches``: slides of at most this many patches are batched into one ragged multi-slide call (default
        ``BATCH_MAX_PATCHES``; 0 = never: every slide takes the per-slide call, whose train-mode dropout masks and operand scales are
        those of the slide alone). ``batch_rows``: most rows of one such call (default ``BATCH_ROWS``; bags that do not lie back to back in one
        allocation are concatenated by a copy of that size).
        ``always_reduce``: issue the gradient all-reduce whenever a process group exists, also at world size 1 (a sum over one
        rank: the values do not change, but the RCCL communicator and its kernel run on the launch stream exactly as they do at
        N > 1 - the single-GPU proof of the collective path, tests/test_gpu_nccl_world1.py and bench.py's allreduce_us)."""
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.always_reduce = bool(always_reduce) and dist.is_initialized()
        self.flat = model.flat_parameters()
        if self.world > 1:                      # identical replicas: one broadcast at construction, never again
            dist.broadcast(self.flat, src=broadcast_from, group=process_group)
        self.flat_grad = torch.zeros_like(self.flat)
        offs, _ = model.flat_offsets()
        sp = model._slot_params()
        self.grads: Dict[str, torch.Tensor] = {}
        for k, p in sp.items():
            o, n = offs[k]
            v = self.flat_grad[o:o + n].view_as(p)
            p.grad = v                          # per-parameter views of the bucket (inspection / hooks)
            self.grads[k] = v
        d, l = sp["wa"].shape
        oa, ob = offs["wa"][0], offs["ba"][0]
        self.grads["wab"] = self.flat_grad[oa:oa + 2 * d * l].view(2 * d, l)   # stacked [dWa;dWb]
        self.grads["bab"] = self.flat_grad[ob:ob + 2 * d]
        # the optimiser sees ONE tensor (the flat buffer; alignment padding has zero gradient and stays
        # zero), so the update is a single elementwise launch instead of a 14-tensor multi-tensor apply
        self.flat_param = torch.nn.Parameter(self.flat, requires_grad=True)
        self.flat_param.grad = self.flat_grad
        # optimizer_factory == "adam" / "sgd" (or a dict of FlatAdam / FlatSGD kwargs, {"opt": "sgd", ...} for SGD) selects the
        # one-launch HIP optimisers (get_optim's two branches, utils/utils.py:63-70); a callable receives [flat_param] and may
        # return any torch optimiser.
        if optimizer_factory == "sgd" or (isinstance(optimizer_factory, dict) and optimizer_factory.get("opt") == "sgd"):
            from .optim import FlatSGD
            kw = {k: v for k, v in optimizer_factory.items() if k != "opt"} if isinstance(optimizer_factory, dict) else {}
            self._flat_adam = FlatSGD(self.flat, **kw)          # same interface: .step(flat_grad)
            self.optimizer = None
        elif optimizer_factory == "adam" or isinstance(optimizer_factory, dict):
            from .optim import FlatAdam
            kw = {k: v for k, v in optimizer_factory.items() if k != "opt"} if isinstance(optimizer_factory, dict) else {}
            self._flat_adam = FlatAdam(self.flat, **kw)
            self.optimizer = None
        else:
            self._flat_adam = None
            self.optimizer = optimizer_factory([self.flat_param])
        self.slide_grad_fn = slide_grad_fn or hip_slide_grad
        self.batch_max_patches = self.BATCH_MAX_PATCHES if batch_max_patches is None else int(batch_max_patches)
        self.batch_rows = self.BATCH_ROWS if batch_rows is None else int(batch_rows)
        # validated HERE: toad_mil_multi_step_f32 takes at most MAX_BATCH_ROWS concatenated rows (32-bit row offsets of the 1024-wide fp32 operand,
        # csrc/gemm_f32.hip h2_nt_ok); a larger batch_rows would fail with a shape error in the middle of accumulate(), after earlier calls of the
        # same step had already written into the gradient bucket
        if not (0 < self.batch_rows <= self.MAX_BATCH_ROWS):
            raise ValueError(f"SlideShardedDP: batch_rows must be in [1, {self.MAX_BATCH_ROWS}] (the ragged multi-slide call's row limit), got {self.batch_rows}")
        if self.batch_max_patches < 0 or self.batch_max_patches > self.batch_rows:
            raise ValueError(f"SlideShardedDP: batch_max_patches must be in [0, batch_rows = {self.batch_rows}], got {self.batch_max_patches}")

    def zero_grad(self):
        self.flat_grad.zero_()

    def _check_flat(self):
        """The optimiser, the all-reduce and the gradient views were bound to the flat buffer at construction. model.to(),
        a deepcopy or a load that re-homes parameters makes the model re-flatten into a NEW buffer; training on would then
        update a buffer the forward no longer reads (a silent no-op), so that is an error here."""
        if self.model.flat_parameters() is not self.flat:
            raise RuntimeError("SlideShardedDP: the model's flat parameter buffer was replaced after construction "
                               "(model.to()/deepcopy/re-flatten); build a new SlideShardedDP for the moved model")

    # slides of at most BATCH_MAX_PATCHES patches are batched into ragged multi-slide calls of up to BATCH_ROWS rows. History: 65,536 / 131,072 let PAIRS
    # of 50,000-patch slides (BASELINE config 4) share their GEMM launches (one 50k-patch bag is 1.5 rounds of 256 x 256 tiles on 256 CUs, two are 3.05:
    # 1.29 -> 1.17 ms per slide, profiles/r04d). Round 5 sized the call for 288 GB of HBM instead of for one tile-plan round: ten 50k-patch slides per call
    # (524,288 rows, ~8 GB of workspace) amortise the per-call helpers (weight split, five K-split fix-ups, slab reduction, heads) and the tile tail over
    # five times the rows: 935 -> 1,022 slides/s on config 4 (profiles/r05g_config4_batch_rows.txt). Both are constructor arguments. The per-slide
    # threshold followed once the batched attention dgrad recomputed the pooling gradient instead of reading a materialised one (csrc/step.hip,
    # gemm_h2_epilogue.inc PBATCH): five 100k-patch slides in one call then run at 515 slides/s against 475 one by one on the same box
    # (profiles/r05r_batched_100k.txt; with the materialised gradient it was 459 against 461).
    BATCH_MAX_PATCHES = 262144
    BATCH_ROWS = 524288
    MAX_BATCH_ROWS = 1_048_575          # (2^32 - 1) // (1024 * 4): rows of a [N, 1024] fp32 operand the NT kernels address with 32-bit byte offsets

    def accumulate(self, slides: Sequence[Slide], global_slides: int, overwrite: bool = True, batched: Optional[bool] = None):
        """grads (+)= sum over ``slides`` of d(loss)/d(params) / global_slides. With ``overwrite`` the
        first slide is written with beta = 0, which replaces a zeroing pass over the bucket.
        ``batched`` (default: when this rank holds several fp32 bags of at most BATCH_MAX_PATCHES patches and the default slide
        function is in use): consecutive small slides go through ONE ragged multi-slide call per <= BATCH_ROWS rows
        (hip_batch_grad): same gradient to fp32 round-off, a fraction of the launches."""
        self._check_flat()
        if not slides:
            if overwrite:
                self.zero_grad()
            return []
        scale = 1.0 / float(global_slides)
        small = lambda s: torch.is_tensor(s[0]) and 0 < s[0].shape[0] <= self.batch_max_patches    # noqa: E731
        if batched is None:
            batched = self.slide_grad_fn is hip_slide_grad and len(slides) > 1 and sum(1 for s in slides if small(s)) > 1
        if batched:
            losses, first, i = [], True, 0
            while i < len(slides):
                j, rows = i, 0
                while j < len(slides) and small(slides[j]) and rows + slides[j][0].shape[0] <= self.batch_rows:
                    rows += slides[j][0].shape[0]; j += 1
                beta = 0.0 if (overwrite and first) else 1.0
                if j - i >= 2:
                    lb = hip_batch_grad(self.model, self.grads, slides[i:j], beta, scale)
                    losses.extend(lb[k] for k in range(j - i))
                else:
                    j = i + 1
                    losses.append(self.slide_grad_fn(self.model, self.grads, slides[i], beta, scale))
                first, i = False, j
            return losses
        return [self.slide_grad_fn(self.model, self.grads, s, 0.0 if (overwrite and i == 0) else 1.0, scale)
                for i, s in enumerate(slides)]

    def reduce(self):
        """ONE all-reduce(SUM) of the flat 4.77 MB gradient bucket, enqueued on the current stream behind the last slide's backward
        (RCCL orders it after the kernels that wrote the bucket; the optimiser launch that follows is ordered after it)."""
        if self.world > 1 or self.always_reduce:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)

    def step(self, slides: Sequence[Slide], global_slides: int):
        """One optimiser step over a global batch of ``global_slides`` slides, of which ``slides``
        are this rank's share. Returns the per-slide device loss vectors (no host sync)."""
        losses = self.accumulate(slides, global_slides)
        self.reduce()
        if self._flat_adam is not None:
            self._flat_adam.step(self.flat_grad)
        else:
            self.optimizer.step()
        return losses
