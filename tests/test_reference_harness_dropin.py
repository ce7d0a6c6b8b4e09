"""Build container only (skipped where /root/reference is absent): the reference's own harness, imported UNMODIFIED from /root/reference, bound to
this repository's model through a `models.model_toad` pre-registration - INTEGRATION.md Option B, executed.

  * test_main_mtl_concat_runs_on_the_dropin: the reference's entry script `main_mtl_concat.py` (:23-78 main, :81-168 parser / dataset / settings)
    run top to bottom through `runpy`, once as it is and once with the drop-in registered (tests/_runpy_main_probe.py: the DEVICE behind the two
    whole-slide C-ABI calls is substituted by a checking recorder that answers with the CPU oracle's values; every line of host code that runs is
    the product's or the reference's). Compared: summary.csv, the checkpoint, split_0_results.pkl.
  * test_reference_harness_drives_the_dropin_host_side: the harness pieces main() does not reach by default (EarlyStopping, eval_utils.initiate_model,
    a `module.`-infixed multi-GPU checkpoint), tests/_option_b_probe.py."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF + "/utils"), reason="the reference tree only exists in the build container")


@needs_reference
@pytest.mark.timeout(300)
def test_reference_harness_drives_the_dropin_host_side():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_option_b_probe.py")], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    for marker in ("BOUND core_utils and eval_utils to toad_amd.model_toad", "Total number of trainable parameters: 1192490", "OPTIM ok",
                   "CHECKPOINT keys ok: 14", "INITIATE_MODEL ok", "OPTION_B_OK"):
        assert marker in r.stdout, (marker, r.stdout[-2000:])
    assert "Validation loss decreased (inf --> 1.250000)" in r.stdout        # the reference's EarlyStopping printed it, saving OUR state_dict


PATCHES = 64                 # per bag (the harness does not care; BASELINE config 1's 256 only costs CPU time here)
PER_CLASS = {"train": 3, "val": 2, "test": 2}       # slides per class and split (at most): summary()'s one-vs-rest AUC needs all 18 classes in val and test


def _write_workdir(root):
    """A class-balanced subset of the reference's OWN dataset CSV and split (dataset_csv/dummy_dataset.csv, splits/dummy_mtl_concat_100/splits_0.csv),
    with the CSV's label typo patched (SURVEY.md 8c gotcha 5), and one seeded random bag per slide under <root>/data/DUMMY_DATA_DIR."""
    import pandas as pd
    import torch
    df = pd.read_csv(REF + "/dataset_csv/dummy_dataset.csv")
    df["label"] = df["label"].replace({"Esophagogogastric": "Esophagogastric"})
    split = pd.read_csv(REF + "/splits/dummy_mtl_concat_100/splits_0.csv", index_col=0)
    label_of = dict(zip(df["slide_id"], df["label"]))
    chosen, cols = set(), {}
    for col, k in PER_CLASS.items():
        seen, keep = {}, []
        for sid in split[col].dropna():
            lab = label_of[sid]
            if seen.get(lab, 0) < k:
                seen[lab] = seen.get(lab, 0) + 1
                keep.append(sid)
        assert len(seen) == 18, (col, seen)          # up to k per class, at least one of each
        cols[col] = keep
        chosen |= set(keep)
    os.makedirs(root + "/dataset_csv"); os.makedirs(root + "/splits/dummy_mtl_concat_100"); os.makedirs(root + "/data/DUMMY_DATA_DIR"); os.makedirs(root + "/results")
    df[df["slide_id"].isin(chosen)].to_csv(root + "/dataset_csv/dummy_dataset.csv", index=False)
    pd.concat([pd.Series(v, name=c) for c, v in cols.items()], axis=1).to_csv(root + "/splits/dummy_mtl_concat_100/splits_0.csv")
    for sid in sorted(chosen):
        g = torch.Generator().manual_seed(int(sid.split("_")[-1]))
        torch.save(torch.randn(PATCHES, 1024, generator=g), f"{root}/data/DUMMY_DATA_DIR/{sid}.pt")
    return {c: len(v) for c, v in cols.items()}


@needs_reference
@pytest.mark.timeout(600)
def test_main_mtl_concat_runs_on_the_dropin(tmp_path):
    """`python main_mtl_concat.py --task dummy_mtl_concat --k 1 --max_epochs 2 --opt sgd --log_data ...` (the reference's script through runpy), pure
    and on the drop-in: same summary.csv, same checkpoint keys / values, same per-slide probabilities. SGD, not the Adam default: the comparison
    is END of training, and Adam's first updates are sign-like (a gradient element at round-off of zero moves its weight by a full lr either
    way; config 1 under Adam is pinned per step in tests/test_config1_golden.py)."""
    import pandas as pd
    import torch
    root = str(tmp_path)
    sizes = _write_workdir(root)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONHASHSEED="1")
    out = {}
    for mode in ("reference", "dropin"):
        cmd = [sys.executable, os.path.join(HERE, "_runpy_main_probe.py"), mode, root, "--task", "dummy_mtl_concat", "--data_root_dir", root + "/data",
               "--results_dir", root + "/results", "--exp_code", mode, "--k", "1", "--max_epochs", "2", "--opt", "sgd", "--lr", "1e-3", "--log_data", "--seed", "1"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
        assert r.returncode == 0 and "RUNPY_MAIN_OK " + mode in r.stdout, (mode, r.stdout[-2500:], r.stderr[-3000:])
        assert "finished!" in r.stdout and "end script" in r.stdout                                 # main_mtl_concat.py:175-177
        assert "Total number of trainable parameters: 1192490" in r.stdout                        # utils/utils.py:72-84 print_network on the model
        res = os.path.join(root, "results", mode + "_s1")
        out[mode] = dict(stdout=r.stdout, probe=json.load(open(os.path.join(root, "probe_%s.json" % mode))),
                         summary=pd.read_csv(os.path.join(res, "summary.csv"), index_col=0),
                         ckpt=torch.load(os.path.join(res, "s_0_checkpoint.pt")),
                         results=pickle.load(open(os.path.join(res, "split_0_results.pkl"), "rb")),
                         files=sorted(os.listdir(res)))
    ref, dro = out["reference"], out["dropin"]
    # the harness really ran on the drop-in, and called the device exactly as often as the reference ran its model
    calls = dro["probe"]["calls"]
    assert dro["probe"]["model_class"] == "toad_amd.model_toad.TOAD_fc_mtl_concat" and calls["relocate"] == 1
    assert calls["mil_fwd"] == calls["mil_bwd"] == 2 * sizes["train"]                             # two epochs of train_loop, one backward per slide
    assert calls["mil_fwd_nograd"] == 2 * sizes["val"] + sizes["val"] + sizes["test"]             # validate x 2 epochs + summary(val) + summary(test)
    # same artefacts (main_mtl_concat.py:66-78,166-168; core_utils:104,149-151)
    assert [f.replace("reference", "X") for f in ref["files"]] == [f.replace("dropin", "X") for f in dro["files"]] == \
        ["0", "experiment_X.txt", "s_0_checkpoint.pt", "split_0_results.pkl", "splits_0.csv", "summary.csv"]
    # summary.csv: same columns, same values (the metrics are ratios of counts and AUCs of probabilities that agree to ~1e-7)
    assert list(ref["summary"].columns) == list(dro["summary"].columns) == ["folds", "cls_test_auc", "cls_val_auc", "cls_test_acc", "cls_val_acc", "site_test_auc",
                                                                             "site_val_auc", "site_test_acc", "site_val_acc"]
    np.testing.assert_allclose(dro["summary"].to_numpy(dtype=float), ref["summary"].to_numpy(dtype=float), rtol=0, atol=1e-9)
    # checkpoint: the reference's keys in the reference's order, fp32, values after 2 x 54 SGD steps within 1e-6 of the reference's
    assert list(dro["ckpt"]) == list(ref["ckpt"]) and len(ref["ckpt"]) == 14
    moved = 0.0
    for k in ref["ckpt"]:
        a, b = dro["ckpt"][k], ref["ckpt"][k]
        assert a.dtype == b.dtype == torch.float32 and a.shape == b.shape, k
        assert (a - b).abs().max().item() <= 1e-6, (k, (a - b).abs().max().item())
        moved = max(moved, b.abs().max().item())
    # per-slide results of the test split
    assert list(dro["results"]) == list(ref["results"]) and len(ref["results"]) == sizes["test"]
    for sid, r in ref["results"].items():
        d = dro["results"][sid]
        assert d["cls_label"] == r["cls_label"] and d["site_label"] == r["site_label"]
        assert d["cls_prob"].shape == r["cls_prob"].shape == (1, 18) and d["cls_prob"].dtype == r["cls_prob"].dtype
        assert np.abs(d["cls_prob"] - r["cls_prob"]).max() <= 1e-6 and np.abs(d["site_prob"] - r["site_prob"]).max() <= 1e-6, sid
    # and the two runs printed the same epoch lines (core_utils:252-253,329-330: losses / errors to 4 decimals)
    import re
    pat = re.compile(r"^(Epoch: \d+, cls train_loss.*|Val Set, cls val_loss.*|Cls (Val|Test) error.*)$", re.M)
    assert [m.group(0) for m in pat.finditer(ref["stdout"])] == [m.group(0) for m in pat.finditer(dro["stdout"])] != []
