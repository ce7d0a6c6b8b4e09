"""TEST INFRASTRUCTURE ONLY — pins toad_amd.train.EarlyStopping to the REAL reference class.

Runs ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/pin_earlystop_against_reference.py [--write]

Drives the unmodified ``EarlyStopping`` of ``utils/core_utils_mtl_concat.py:44-85`` over recorded validation-loss sequences
(improving, plateau, ties, noisy, early / late degradation, several (patience, stop_epoch) settings incl. the reference's hard-wired
(20, 50) of ``core_utils:134``) with a stub model whose ``state_dict()`` is the epoch number, and stores per call
(counter, best_score, early_stop, epoch of the checkpoint on disk) in tests/golden/toad_earlystop_golden.npz. With --write the
fixture is (re)written; without, this repo's class is replayed against the reference directly.
One shim is needed to import the reference under NumPy 2: ``np.Inf`` (removed) is aliased to ``np.inf`` before the class is built.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle.pin_against_reference import import_reference  # noqa: E402


def sequences():
    rng = np.random.default_rng(5)
    seqs = {}
    seqs["improving"] = np.linspace(2.0, 0.5, 40)
    seqs["plateau_then_worse"] = np.concatenate([np.linspace(2.0, 1.0, 10), np.full(30, 1.0), np.linspace(1.0, 1.5, 40)])
    seqs["ties"] = np.array([1.0, 1.0, 0.9, 0.9, 0.9, 1.1, 0.9, 1.2] * 10)
    seqs["noisy_walk"] = 1.5 + np.cumsum(rng.normal(0, 0.05, 120)) * 0.3
    seqs["early_minimum"] = np.concatenate([[0.3], np.linspace(0.8, 2.0, 100)])
    seqs["late_drop"] = np.concatenate([np.linspace(1.0, 1.4, 70), [0.2], np.linspace(0.5, 0.9, 40)])
    return {k: v.astype(np.float64) for k, v in seqs.items()}


SETTINGS = [(20, 50), (3, 5), (1, 0), (5, 60)]


class _Stub:
    def __init__(self):
        self.epoch = -1

    def state_dict(self):
        return {"epoch": torch.tensor(self.epoch)}


def drive(cls, losses, patience, stop_epoch):
    out = []
    with tempfile.TemporaryDirectory() as d:
        ck = os.path.join(d, "ck.pt")
        es = cls(patience=patience, stop_epoch=stop_epoch, verbose=False)
        m = _Stub()
        for epoch, v in enumerate(losses):
            m.epoch = epoch
            with contextlib.redirect_stdout(io.StringIO()):
                es(epoch, float(v), m, ckpt_name=ck)
            saved = int(torch.load(ck)["epoch"]) if os.path.exists(ck) else -1
            out.append((es.counter, float(es.best_score), int(es.early_stop), saved, float(es.val_loss_min)))
            if es.early_stop:
                break
    return np.array(out, dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    if not hasattr(np, "Inf"):
        np.Inf = np.inf                                   # the reference predates NumPy 2 (core_utils:61)
    import_reference()
    from utils.core_utils_mtl_concat import EarlyStopping as RefES  # type: ignore
    from toad_amd.train import EarlyStopping as MyES
    store = {}
    for name, losses in sequences().items():
        store["loss/" + name] = losses
        for (p, s) in SETTINGS:
            ref = drive(RefES, losses, p, s)
            mine = drive(MyES, losses, p, s)
            assert ref.shape == mine.shape and np.array_equal(ref, mine), (name, p, s)
            store[f"trace/{name}/{p}_{s}"] = ref
            print(f"{name:20s} patience {p:2d} stop_epoch {s:2d}: {len(ref):3d} calls, stopped={bool(ref[-1, 2])}, checkpoint epoch {int(ref[-1, 3])}")
    if args.write:
        path = os.path.join(REPO, "tests", "golden", "toad_earlystop_golden.npz")
        np.savez_compressed(path, **store)
        print("wrote", path)


if __name__ == "__main__":
    main()
