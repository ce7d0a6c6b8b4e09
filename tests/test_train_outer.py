"""The reference's train() outer loop (utils/core_utils_mtl_concat.py:87-187): early stopping -> checkpoint -> reload -> summaries.
CPU: toad_amd.train.EarlyStopping replayed against traces captured from the REAL reference class
(oracle/pin_earlystop_against_reference.py -> tests/golden/toad_earlystop_golden.npz). GPU: train() end to end on tiny loaders."""
import os
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class _Stub:
    epoch = -1

    def state_dict(self):
        return {"epoch": torch.tensor(self.epoch)}


def test_early_stopping_replays_the_reference_traces(tmp_path):
    from toad_amd.train import EarlyStopping
    g = np.load(os.path.join(HERE, "golden", "toad_earlystop_golden.npz"))
    traces = [k for k in g.files if k.startswith("trace/")]
    assert len(traces) == 24
    for key in traces:
        _, name, ps = key.split("/")
        patience, stop_epoch = (int(v) for v in ps.split("_"))
        losses, ref = g["loss/" + name], g[key]
        es, m, ck = EarlyStopping(patience=patience, stop_epoch=stop_epoch), _Stub(), str(tmp_path / (name + ps + ".pt"))
        for epoch, v in enumerate(losses):
            m.epoch = epoch
            es(epoch, float(v), m, ckpt_name=ck)
            saved = int(torch.load(ck)["epoch"])
            got = (es.counter, float(es.best_score), int(es.early_stop), saved, float(es.val_loss_min))
            assert got == tuple(ref[epoch]), (key, epoch, got, tuple(ref[epoch]))
            if es.early_stop:
                assert epoch == len(ref) - 1
                break
        else:
            assert len(ref) == len(losses) and not ref[-1, 2]


def _slides(n, seed, n_classes):
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(seed + i)
        rows = 64 + (i * 53) % 300
        # a learnable signal: the class shifts the mean of a few feature columns
        x = torch.randn(rows, 1024, generator=g)
        x[:, : 8] += (i % n_classes) * 0.75
        out.append((x, torch.tensor([i % n_classes]), torch.tensor([(i // 2) % 2]), torch.tensor([float(i % 2)])))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("early", [True, False])
def test_train_outer_loop(cuda, tmp_path, early):
    from toad_amd.train import train, validate
    from toad_amd import TOAD_fc_mtl_concat
    nc = 3
    args = types.SimpleNamespace(drop_out=False, n_classes=nc, opt="adam", lr=2e-4, reg=1e-5, early_stopping=early, max_epochs=8,
                                 results_dir=str(tmp_path / "res"))
    torch.manual_seed(11)
    loaders = (_slides(24, 100, nc), _slides(9, 500, nc), _slides(9, 900, nc))
    out = train(loaders, 0, args, patience=2, stop_epoch=1)
    assert len(out) == 10
    results, log = out[0], out[9]
    ckpt = os.path.join(args.results_dir, "s_0_checkpoint.pt")
    assert os.path.exists(ckpt)
    model = results["model"]
    sd = torch.load(ckpt, map_location="cpu")
    for k, v in model.state_dict().items():                       # the model that produced the summaries IS the checkpoint
        assert torch.equal(v.cpu(), sd[k]), k
    vl = [e["val_cls_loss"] for e in log]
    if early:
        # replay the reference state machine on the recorded validation losses: same stopping epoch, checkpoint = best epoch
        from toad_amd.train import EarlyStopping
        es, m = EarlyStopping(patience=2, stop_epoch=1), _Stub()
        stop_at = None
        for epoch, v in enumerate(vl):
            es(epoch, v, m, ckpt_name=str(tmp_path / "replay.pt"))
            if es.early_stop:
                stop_at = epoch
                break
        assert (stop_at is None and len(log) == args.max_epochs) or stop_at == len(log) - 1
        best = int(np.argmin(np.array(vl)))                        # ties resolve to the LAST best epoch (>= counts as improvement)
        best = max(i for i, v in enumerate(vl) if v == vl[best])
        # the reloaded weights reproduce the best epoch's validation loss
        again = validate(model, loaders[1], nc)
        assert abs(again["cls_loss"] - vl[best]) <= 1e-5 * max(1.0, abs(vl[best])), (again["cls_loss"], vl, best)
    else:
        assert len(log) == args.max_epochs
    # the 9 reference values: AUCs in [0, 1] (or nan when a class is absent), accuracies = 1 - error
    assert all((0.0 <= v <= 1.0) or v != v for v in out[1:9])
    assert abs(out[3] - (1.0 - results["test"]["cls_error"])) < 1e-12 and abs(out[4] - (1.0 - results["val"]["cls_error"])) < 1e-12
    # training moved the class loss down on the training signal
    assert log[-1]["train_cls_loss"] < log[0]["train_cls_loss"]
    # a model built afresh loads the checkpoint through the reference's path (eval_utils:28-29)
    m2 = TOAD_fc_mtl_concat(n_classes=nc); m2.load_state_dict(sd); m2.relocate()
    assert abs(validate(m2, loaders[1], nc)["cls_loss"] - results["val"]["cls_loss"]) <= 1e-6
