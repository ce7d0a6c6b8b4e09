"""Train / validate loops over the drop-in model, mirroring the reference harness.

Reference: ``utils/core_utils_mtl_concat.py`` — ``train_loop`` (:189-259), ``validate`` (:262-366),
``Accuracy_Logger`` (:13-43), ``calculate_error`` (``utils/utils.py:135-138``). The sequence per slide is
the reference's (to device, ``model(data, sex)``, ``0.75*CE + 0.25*CE``, ``backward``, ``step``,
``zero_grad``); what differs is bookkeeping: the reference synchronises the host three times per slide
(two ``.item()`` at :216-217 and ``int(Y_hat)`` at :24); here the running sums live on the device and are
read back once per epoch.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn


class AccuracyLogger:
    """Per-class hit / count tallies (reference ``Accuracy_Logger``), kept on the device."""

    def __init__(self, n_classes: int, device):
        self.n_classes = n_classes
        self.count = torch.zeros(n_classes, dtype=torch.int64, device=device)
        self.correct = torch.zeros(n_classes, dtype=torch.int64, device=device)

    def log(self, y_hat: torch.Tensor, y: torch.Tensor) -> None:
        y = y.reshape(-1).to(torch.int64)
        hit = (y_hat.reshape(-1).to(torch.int64) == y).to(torch.int64)
        self.count.index_add_(0, y, torch.ones_like(y))
        self.correct.index_add_(0, y, hit)

    def summary(self):
        """[(acc or None, correct, count)] per class — ``get_summary`` of the reference."""
        c, n = self.correct.cpu().tolist(), self.count.cpu().tolist()
        return [((ci / ni) if ni else None, ci, ni) for ci, ni in zip(c, n)]


def _to_device(batch, device):
    data, label, site, sex = batch
    return (data.to(device, non_blocking=True), label.to(device, non_blocking=True),
            site.to(device, non_blocking=True), sex.float().to(device, non_blocking=True))


class _FusedGrads:
    """The flat gradient buffer of a model and its per-slot views (what toad_mil_step_f32 writes), with every parameter's
    ``.grad`` pointing into it so that a torch optimiser over ``model.parameters()`` reads the same memory."""

    def __init__(self, model):
        self.flat = model.flat_parameters()
        self.flat_grad = torch.zeros_like(self.flat)
        offs, _ = model.flat_offsets()
        sp = model._slot_params()
        self.views: Dict[str, torch.Tensor] = {}
        for k, p in sp.items():
            o, n = offs[k]
            self.views[k] = self.flat_grad[o:o + n].view_as(p)
        d, l = sp["wa"].shape
        oa, ob = offs["wa"][0], offs["ba"][0]
        self.views["wab"] = self.flat_grad[oa:oa + 2 * d * l].view(2 * d, l)
        self.views["bab"] = self.flat_grad[ob:ob + 2 * d]
        self.params = sp

    def bind(self):
        for k, p in self.params.items():
            if p.grad is not self.views[k]:
                p.grad = self.views[k]


def _fused_grads(model) -> "_FusedGrads":
    st = getattr(model, "_fused_grads", None)
    if st is None or st.flat is not model.flat_parameters():
        st = _FusedGrads(model)
        model._fused_grads = st
    return st


def _fused_ok(model, optimizer, loss_fn) -> bool:
    """The fused slide step computes exactly 0.75*CE(logits, label) + 0.25*CE(site_logits, site) with default CE settings
    (core_utils_mtl_concat.py:213-215 with the loss_fn of :105)."""
    from .model_toad import TOAD_fc_mtl_concat
    if not isinstance(model, TOAD_fc_mtl_concat) or not next(model.parameters()).is_cuda:
        return False
    if loss_fn is not None:
        if type(loss_fn) is not nn.CrossEntropyLoss or loss_fn.weight is not None or loss_fn.reduction != "mean" \
                or getattr(loss_fn, "label_smoothing", 0.0) != 0.0 or loss_fn.ignore_index != -100:
            return False
    if any(not p.requires_grad for p in model.parameters()):
        return False
    # the fused step writes gradients straight into views every parameter's .grad aliases (beta = 0: whatever .grad held is
    # overwritten, autograd hooks never fire). That is only the reference's semantics when nothing hangs on the autograd path and
    # the optimiser steps exactly this model's parameters:
    if any(getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) for p in model.parameters()):
        return False                                             # gradient hooks (clipping, logging): take the autograd path
    from .optim import FlatAdam, FlatSGD
    if optimizer is not None and not isinstance(optimizer, (FlatAdam, FlatSGD)):
        groups = getattr(optimizer, "param_groups", None)
        if groups is None:
            return False
        owned = {id(p) for g in groups for p in g["params"]}
        mine = {id(p) for p in model.parameters()}
        if owned != mine:                                        # a subset / other tensors: the fused path would silently train something else
            return False
    return True


def train_loop(epoch: int, model, loader: Iterable, optimizer, n_classes: int, loss_fn=None, fused: Optional[bool] = None) -> Dict[str, object]:
    """One epoch, one optimiser step per slide (the reference's batch size is 1, utils/utils.py:51-55).

    ``fused`` (default: whenever the model is the HIP TOAD module and the loss is the reference's plain cross-entropy): forward,
    the weighted loss and the backward of a slide are ONE library call (``toad_mil_step_f32``) that writes the gradients into
    a flat buffer every parameter's ``.grad`` aliases; the optimiser then steps as usual (a torch optimiser over
    ``model.parameters()``, or the one-launch ``FlatAdam`` / ``FlatSGD`` of ``toad_amd.optim.get_optim``). The per-slide
    numbers are those of the unfused sequence below (tests/test_gpu_model.py::test_fused_loss_path_equals_autograd_path);
    what it removes is the host path of autograd + two CE modules + zero_grad, which at 256 patches costs three times the
    GPU work. ``fused=False`` runs the reference's sequence literally."""
    device = next(model.parameters()).device
    if fused is None:
        fused = _fused_ok(model, optimizer, loss_fn)
    elif fused and not _fused_ok(model, optimizer, loss_fn):
        raise ValueError("train_loop(fused=True) needs the HIP TOAD module on the device and the default CrossEntropyLoss")
    loss_fn = loss_fn or nn.CrossEntropyLoss()
    model.train()
    cls_logger, site_logger = AccuracyLogger(n_classes, device), AccuracyLogger(2, device)
    sums = torch.zeros(4, dtype=torch.float64, device=device)      # cls loss, site loss, cls error, site error
    n = 0
    if fused:
        from . import ops
        from .model_toad import _draw_dropout
        from .optim import FlatAdam, FlatSGD
        fg = _fused_grads(model)
        fg.bind()                                                     # .grad views for every optimiser kind: inspection / clipping see the gradients
        flat_opt = isinstance(optimizer, (FlatAdam, FlatSGD))
        if flat_opt and optimizer.p is not fg.flat:
            raise RuntimeError("train_loop: the flat optimiser was built for a parameter buffer the model no longer uses")
        if not flat_opt:
            fg.bind()
    for batch in loader:
        data, label, site, sex = _to_device(batch, device)
        if fused and data.shape[0] > 0:
            w = {k: v.detach() for k, v in model._weights().items()}
            drop_p, seed = _draw_dropout(model._dropout and model.training)
            loss3, logits, slog = ops.mil_step(w, fg.views, 0.0, model._bag_dtype(data.contiguous()), sex.reshape(1), label.reshape(1), site.reshape(1),
                                               0.75, 0.25, drop_p, seed, want_logits=True)
            y_hat, s_hat = logits.argmax(1), slog.argmax(1)
            cls_logger.log(y_hat, label)
            site_logger.log(s_hat, site)
            sums += torch.stack([loss3[1].double(), loss3[2].double(), (y_hat != label.reshape(-1)).double().mean(),
                                 (s_hat != site.reshape(-1)).double().mean()])
            if flat_opt:
                optimizer.step(fg.flat_grad)
            else:
                optimizer.step()                                      # reads the .grad views bound above; nothing to zero (beta = 0)
            n += 1
            continue
        res = model(data, sex)
        cls_loss = loss_fn(res["logits"], label)
        site_loss = loss_fn(res["site_logits"], site)
        loss = cls_loss * 0.75 + site_loss * 0.25                     # core_utils:213-215
        cls_logger.log(res["Y_hat"], label)
        site_logger.log(res["site_hat"], site)
        with torch.no_grad():
            sums += torch.stack([cls_loss.detach().double(), site_loss.detach().double(),
                                 (res["Y_hat"].reshape(-1) != label.reshape(-1)).double().mean(),   # calculate_error
                                 (res["site_hat"].reshape(-1) != site.reshape(-1)).double().mean()])
        if fused:                                                     # an empty bag inside a fused epoch: autograd path, same buffers
            from .optim import FlatAdam, FlatSGD
            for p in model.parameters():
                p.grad = None
            loss.backward()
            if isinstance(optimizer, (FlatAdam, FlatSGD)):
                offs, _ = model.flat_offsets()
                fg.flat_grad.zero_()
                for k, p in fg.params.items():
                    if p.grad is not None:
                        fg.views[k].copy_(p.grad)
                optimizer.step(fg.flat_grad)
            else:
                optimizer.step()
            fg.bind()
            n += 1
            continue
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
        n += 1
    s = (sums / max(n, 1)).cpu().tolist()                             # the epoch's only host sync
    return {"epoch": epoch, "slides": n, "cls_loss": s[0], "site_loss": s[1], "cls_error": s[2], "site_error": s[3],
            "cls_acc": cls_logger.summary(), "site_acc": site_logger.summary()}


def _auc(labels: np.ndarray, probs: np.ndarray, n_classes: int) -> float:
    """core_utils:311-330: binary -> AUC of class 1; multi-class -> mean one-vs-rest AUC over classes present."""
    from sklearn.metrics import auc as calc_auc, roc_auc_score, roc_curve
    from sklearn.preprocessing import label_binarize
    if n_classes == 2:
        return float(roc_auc_score(labels, probs[:, 1]))
    binary = label_binarize(labels, classes=list(range(n_classes)))
    aucs = []
    for c in range(n_classes):
        if c in labels:
            fpr, tpr, _ = roc_curve(binary[:, c], probs[:, c])
            aucs.append(calc_auc(fpr, tpr))
        else:
            aucs.append(float("nan"))
    return float(np.nanmean(np.array(aucs)))


@torch.no_grad()
def validate(model, loader: Iterable, n_classes: int, loss_fn=None, with_auc: bool = True, group_rows: int = 0) -> Dict[str, object]:
    """Forward-only pass (reference ``validate`` / ``summary``): losses, errors, per-slide probabilities, AUCs. ``group_rows`` > 0 opts into
    ``eval.forward_grouped``'s ragged multi-slide forward (default 0: one ``model(data, sex)`` per slide, like the reference)."""
    device = next(model.parameters()).device
    loss_fn = loss_fn or nn.CrossEntropyLoss()
    model.eval()
    cls_logger, site_logger = AccuracyLogger(n_classes, device), AccuracyLogger(2, device)
    sums = torch.zeros(4, dtype=torch.float64, device=device)
    probs, site_probs, labels, sites = [], [], [], []
    n = 0
    from .eval import forward_grouped
    for (data, label, site, sex), res in forward_grouped(model, (_to_device(b, device) for b in loader), group_rows):
        cls_logger.log(res["Y_hat"], label)
        site_logger.log(res["site_hat"], site)
        sums += torch.stack([loss_fn(res["logits"], label).double(), loss_fn(res["site_logits"], site).double(),
                             (res["Y_hat"].reshape(-1) != label.reshape(-1)).double().mean(),
                             (res["site_hat"].reshape(-1) != site.reshape(-1)).double().mean()])
        probs.append(res["Y_prob"]); site_probs.append(res["site_prob"]); labels.append(label); sites.append(site)
        n += 1
    s = (sums / max(n, 1)).cpu().tolist()
    out = {"slides": n, "cls_loss": s[0], "site_loss": s[1], "cls_error": s[2], "site_error": s[3],
           "cls_acc": cls_logger.summary(), "site_acc": site_logger.summary()}
    if n:
        out["prob"] = torch.cat(probs).cpu().numpy(); out["site_prob"] = torch.cat(site_probs).cpu().numpy()
        out["labels"] = torch.cat(labels).reshape(-1).cpu().numpy(); out["sites"] = torch.cat(sites).reshape(-1).cpu().numpy()
        if with_auc:
            try:
                out["cls_auc"] = _auc(out["labels"], out["prob"], n_classes)
                out["site_auc"] = _auc(out["sites"], out["site_prob"], 2)
            except Exception as e:                                   # e.g. a single class present
                out["auc_error"] = repr(e)
    return out


def train_loop_dp(epoch: int, dp, loader: Iterable, batch_slides: int) -> Dict[str, object]:
    """One epoch under DATA-PARALLEL semantics: one optimiser step per ``batch_slides`` slides of this rank (times the world size), through
    ``SlideShardedDP.step`` - small fp32 bags of a batch go through ONE ragged multi-slide library call (toad_mil_multi_step_f32: trunk /
    attention GEMMs once over the concatenated bags), the gradient is all-reduced once per step. The reference steps once per SLIDE
    (utils/core_utils_mtl_concat.py:200-234, batch size 1, utils/utils.py:51-55); this is the loop to use when its real bags - a few
    hundred to a few thousand patches - should fill a GPU: 77.6k instead of 4.5k slides/s at 256 patches (64 per step). Every rank must
    see the same number of batches. Returns the epoch's mean class / site losses (one host sync at the end)."""
    device = dp.flat.device
    world = dp.world
    sums = torch.zeros(2, dtype=torch.float64, device=device)
    n, batch = 0, []

    def flush():
        nonlocal n
        losses = dp.step([(b[0], b[3], b[1], b[2]) for b in batch], len(batch) * world)
        for lv in losses:
            sums.add_(torch.stack([lv[1].double(), lv[2].double()]))
        n += len(batch)
        batch.clear()
    dp.model.train()
    for b in loader:
        batch.append(_to_device(b, device))
        if len(batch) == batch_slides:
            flush()
    if batch:
        flush()
    s = (sums / max(n, 1)).cpu().tolist()
    return {"epoch": epoch, "slides": n, "cls_loss": s[0], "site_loss": s[1]}
