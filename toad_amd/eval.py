"""Evaluation over the drop-in model: checkpoint loading, per-slide summary, top-k / AUC metrics, heat-map scores.

Reference: ``utils/eval_utils_mtl_concat.py`` — ``initiate_model`` (:19-32), ``eval`` (:34-46), ``accuracy``
(:49-63), ``summary`` (:65-177); ``attention_only`` scoring is ``models/model_toad.py:93-94`` as used by the
reference's heat-map scripts. Names, argument meaning and result keys follow the reference so that its
``eval_mtl_concat.py`` reads the same dictionaries; the forward pass is the HIP path (``TOAD_fc_mtl_concat``),
the metrics are host-side numpy / scikit-learn exactly like the reference's.

What differs from the reference: per-slide probabilities are collected on the device and read back once
(the reference synchronises 4+ times per slide through ``.item()`` / ``.cpu()``), and ``summary`` does not
raise ``NameError`` for ``n_classes == 2`` (the reference's ``topk`` is only bound for ``n_classes > 2``,
``eval_utils:125-131,174`` — here the top-k block is simply skipped, as the reference intends).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence

import numpy as np
import torch

from .model_toad import TOAD_fc_mtl_concat
from .train import AccuracyLogger, _to_device


def initiate_model(args, ckpt_path: Optional[str] = None) -> TOAD_fc_mtl_concat:
    """``eval_utils:19-32``: build from ``args.drop_out`` / ``args.n_classes``, relocate, load (non-strict), eval()."""
    model = TOAD_fc_mtl_concat(dropout=args.drop_out, n_classes=args.n_classes)
    model.relocate()
    if ckpt_path is not None:
        ckpt = torch.load(ckpt_path, map_location="cpu")
        # non-strict like the reference (eval_utils:28-29), but never silently partial: DataParallel-era keys
        # (``attention_net.module.*``) are renamed by the model's loader, and anything still missing is an error
        res = model.load_state_dict(ckpt, strict=False)
        if res.missing_keys:
            raise RuntimeError(f"{ckpt_path}: checkpoint lacks {len(res.missing_keys)} model tensors "
                               f"(e.g. {res.missing_keys[:3]}); refusing to evaluate a partly random model")
        if res.unexpected_keys:
            import warnings
            warnings.warn(f"{ckpt_path}: ignoring {len(res.unexpected_keys)} unexpected keys, e.g. {res.unexpected_keys[:3]}")
    model.eval()
    return model


def accuracy(output: torch.Tensor, target: torch.Tensor, topk: Sequence[int] = (1,)):
    """Top-k accuracy over rows of ``output`` (``eval_utils:49-63``); returns one 1-element tensor per k."""
    with torch.no_grad():
        maxk = max(topk)
        n = target.size(0)
        pred = output.topk(maxk, 1, True, True)[1].t()                    # [maxk, n]
        hit = pred.eq(target.view(1, -1).expand_as(pred))
        return [hit[:k].reshape(-1).float().sum(0, keepdim=True) * (1.0 / n) for k in topk]


def _cls_auc(labels: np.ndarray, probs: np.ndarray, n_classes: int, micro_average: bool):
    """``eval_utils:133-158``: (-1, []) for a single class present; binary -> AUC of column 1;
    multi-class -> per-class one-vs-rest AUCs (nan for absent classes) and their nan-mean, or the micro average."""
    from sklearn.metrics import auc, roc_auc_score, roc_curve
    from sklearn.preprocessing import label_binarize
    if len(np.unique(labels)) == 1:
        return -1, []
    if n_classes == 2:
        return float(roc_auc_score(labels, probs[:, 1])), []
    binary = label_binarize(labels, classes=list(range(n_classes)))
    aucs = []
    for c in range(n_classes):
        if c in labels:
            fpr, tpr, _ = roc_curve(binary[:, c], probs[:, c])
            aucs.append(float(auc(fpr, tpr)))
        else:
            aucs.append(float("nan"))
    if micro_average:
        valid = np.where(np.any(binary, axis=0))[0]
        fpr, tpr, _ = roc_curve(binary[:, valid].ravel(), probs[:, valid].ravel())
        return float(auc(fpr, tpr)), aucs
    return float(np.nanmean(np.array(aucs))), aucs


def forward_grouped(model, batches: Iterable, group_rows: int = 0):
    """Yield ``(batch, result_dict)`` in loader order for an iterable of ``(data, label, site, sex)`` device batches.
    Consecutive slides are collected until their patch counts reach ``group_rows`` and forwarded with ONE pass of the
    trunk GEMMs (``TOAD_fc_mtl_concat.forward_many``): a 256-patch slide costs as many kernel launches as a 100k-patch one,
    so the reference's per-slide loop (eval_utils_mtl_concat.py:88-91, core_utils_mtl_concat.py:281-284) is launch-bound on
    small bags. A slide that alone reaches ``group_rows`` goes through ``model(data, sex)``. DEFAULT ``group_rows = 0``: no grouping - the
    reference's one ``model(data, sex)`` per slide, whose numbers do not depend on which neighbours the loader put next to a slide.
    Grouping is opt-in (e.g. 131072: 17.6k instead of 6.7k slides/s on 128..4096-patch bags): the GEMM operand scales are taken per
    256-row block of the CONCATENATED bags, so grouped results agree with the per-slide ones to fp32 round-off (2e-5), not bitwise,
    and the group holds its bags plus their concatenation on the device."""
    pending, rows = [], 0

    def flush():
        if len(pending) == 1:
            b = pending[0]
            yield b, model(b[0], b[3])
        elif pending:
            for b, r in zip(pending, model.forward_many([b[0] for b in pending], [b[3] for b in pending])):
                yield b, r

    for b in batches:
        n = int(b[0].shape[0])
        if group_rows <= 0 or n == 0 or n >= group_rows or not hasattr(model, "forward_many"):
            yield from flush(); pending, rows = [], 0
            yield b, model(b[0], b[3])
            continue
        if rows + n > group_rows and pending:
            yield from flush(); pending, rows = [], 0
        pending.append(b); rows += n
    yield from flush()


@torch.no_grad()
def summary(model, loader: Iterable, args, slide_ids: Optional[Sequence] = None, group_rows: int = 0) -> Dict[str, object]:
    """Forward every slide once and tabulate (``eval_utils:65-177``).

    ``slide_ids`` defaults to ``loader.dataset.slide_data['slide_id']`` like the reference; pass a list when the
    loader is a plain iterable. ``group_rows`` > 0: consecutive slides are forwarded together (``forward_grouped``) until
    their patch counts add up to it; 0 keeps the reference's one ``model(data, sex)`` per slide. Returns the reference's keys: ``patient_results, cls_test_error, cls_auc,
    cls_aucs, site_test_error, site_auc, loggers, df`` and ``top{k}_acc``.
    """
    import pandas as pd
    device = next(model.parameters()).device
    n_classes = args.n_classes
    model.eval()
    cls_logger, site_logger = AccuracyLogger(n_classes, device), AccuracyLogger(2, device)
    if slide_ids is None:
        slide_ids = loader.dataset.slide_data["slide_id"]
    ids = list(slide_ids)
    probs, site_probs, labels, sites, sexes, y_hats, s_hats = [], [], [], [], [], [], []
    for (data, label, site, sex), res in forward_grouped(model, (_to_device(b, device) for b in loader), group_rows):
        cls_logger.log(res["Y_hat"], label)
        site_logger.log(res["site_hat"], site)
        probs.append(res["Y_prob"]); site_probs.append(res["site_prob"])
        labels.append(label.reshape(-1)); sites.append(site.reshape(-1)); sexes.append(sex.reshape(-1))
        y_hats.append(res["Y_hat"].reshape(-1)); s_hats.append(res["site_hat"].reshape(-1))
    n = len(probs)
    if n == 0:
        raise ValueError("summary: empty loader")
    # the loop's only host read-backs
    all_cls_probs = torch.cat(probs).double().cpu().numpy()
    all_site_probs = torch.cat(site_probs).double().cpu().numpy()
    all_cls_labels = torch.cat(labels).double().cpu().numpy()
    all_site_labels = torch.cat(sites).double().cpu().numpy()
    all_sexes = torch.cat(sexes).double().cpu().numpy()
    y_hat = torch.cat(y_hats).cpu().numpy(); s_hat = torch.cat(s_hats).cpu().numpy()
    cls_test_error = float(np.mean(y_hat != all_cls_labels))                # mean of calculate_error per slide
    site_test_error = float(np.mean(s_hat != all_site_labels))

    patient_results = {}
    for i in range(n):
        sid = ids[i]
        patient_results[sid] = {"slide_id": np.array(sid), "cls_prob": all_cls_probs[i:i + 1].astype(np.float32),
                                "cls_label": int(all_cls_labels[i]), "site_prob": all_site_probs[i:i + 1].astype(np.float32),
                                "site_label": int(all_site_labels[i])}

    topk, topk_accs = (), []
    if n_classes > 2:
        topk = (1, 3, 5) if n_classes > 5 else (1, 3)
        topk_accs = accuracy(torch.from_numpy(all_cls_probs), torch.from_numpy(all_cls_labels), topk=topk)

    cls_auc, cls_aucs = _cls_auc(all_cls_labels, all_cls_probs, n_classes, bool(getattr(args, "micro_average", False)))
    if len(np.unique(all_site_labels)) == 1:
        site_auc = -1
    else:
        from sklearn.metrics import roc_auc_score
        site_auc = float(roc_auc_score(all_site_labels, all_site_probs[:, 1]))

    table = {"slide_id": ids[:n], "sex": all_sexes, "Y": all_cls_labels, "Y_hat": np.argmax(all_cls_probs, axis=1),
             "site": all_site_labels, "site_hat": np.argmax(all_site_probs, axis=1)}
    for c in range(n_classes):
        table["p_{}".format(c)] = all_cls_probs[:, c]
    table["site_p"] = all_site_probs[:, 1]
    out = {"patient_results": patient_results, "cls_test_error": cls_test_error, "cls_auc": cls_auc, "cls_aucs": cls_aucs,
           "site_test_error": site_test_error, "site_auc": site_auc, "loggers": (cls_logger, site_logger),
           "df": pd.DataFrame(table)}
    for k, a in zip(topk, topk_accs):
        out["top{}_acc".format(k)] = a.item()
    return out


def eval(loader: Iterable, args, ckpt_path: Optional[str], slide_ids: Optional[Sequence] = None):   # noqa: A001 (reference name)
    """``eval_utils:34-46``: model from checkpoint + ``summary``. Takes the loader (the reference builds it from
    its dataset with ``get_simple_loader``: sequential, batch size 1)."""
    model = initiate_model(args, ckpt_path)
    return model, summary(model, loader, args, slide_ids=slide_ids)


@torch.no_grad()
def attention_heatmap_scores(model: TOAD_fc_mtl_concat, data: torch.Tensor, percentile: bool = False) -> torch.Tensor:
    """Raw task-0 attention score per patch: ``model(data, sex, attention_only=True)`` (``model_toad.py:93-94``),
    which runs the trunk GEMMs and the fused gate/score kernel and skips pooling and heads. With ``percentile``
    the scores are converted to their rank in [0, 1] on the device (what the heat-map colour scale consumes)."""
    sex = torch.zeros(1, device=data.device)
    a = model(data, sex, attention_only=True)
    if not percentile:
        return a
    order = torch.argsort(a)
    ranks = torch.empty_like(a)
    ranks[order] = torch.arange(a.numel(), device=a.device, dtype=a.dtype)
    return ranks / max(a.numel() - 1, 1)
