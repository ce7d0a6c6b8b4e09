"""Run by tests/test_reference_harness_dropin.py in a subprocess (build container only: needs /root/reference).

    python tests/_runpy_main_probe.py {reference|dropin} WORKDIR [main_mtl_concat.py arguments ...]

Executes the reference's OWN entry script, ``/root/reference/main_mtl_concat.py`` (argument parser, ``seed_torch``, ``Generic_MIL_MTL_Dataset``
on ``dataset_csv/dummy_dataset.csv``, ``train()`` -> ``train_loop`` / ``validate`` / ``summary``, checkpoint, ``split_0_results.pkl``,
``summary.csv``), unmodified, through ``runpy`` with WORKDIR as the current directory (the script reads ``dataset_csv/`` and ``splits/``
relative to it; the test writes a class-balanced subset of the reference's own CSV / split there).

  * ``reference``: nothing but the import shims of SURVEY.md 8(c) (modules absent from the image and unused on the path).
  * ``dropin``: INTEGRATION.md Option B - ``models.model_toad`` is pre-registered with THIS repository's classes, so the harness constructs,
    relocates, optimises, trains, evaluates and checkpoints ``toad_amd.TOAD_fc_mtl_concat``. The build container has no GPU and the product
    has no CPU path, so the test substitutes the DEVICE and nothing else: the two whole-slide C-ABI calls behind ``model(data, sex)`` and
    ``loss.backward()`` (``ops.mil_fwd`` = toad_mil_fwd_f32, ``ops.mil_bwd`` = toad_mil_bwd_f32) are replaced by a recorder that checks what
    the host side hands to the library (contiguous fp32 tensors of the documented shapes, the 16 parameter slots as views of the flat
    buffer, gradient destinations) and answers with the CPU oracle's values in the library's own arena layout; the CUDA-only guards and the
    device move of ``relocate()`` are lifted. Everything else that runs is product host code driven by reference code: constructor kwargs,
    ``relocate``, flat parameter buffer, ``parameters()`` for ``get_optim``, ``train()`` / ``eval()``, the autograd bridge (``ToadMIL``),
    the no-grad forward with its copied-out outputs, result-dict keys / shapes / dtypes, ``state_dict`` -> checkpoint.
This file is test infrastructure (it imports ``oracle/``); the kernels behind the same two calls are covered by the -m gpu tests."""
import importlib.machinery
import json
import os
import runpy
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
mode, workdir, argv = sys.argv[1], sys.argv[2], sys.argv[3:]
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)


class _Writer:                                                  # tensorboardX.SummaryWriter (core_utils:96-98; --log_data is effectively mandatory, :186)
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


for name in ("torchvision", "torchvision.transforms", "h5py", "tensorboardX", "torchsummary"):      # absent from the image, unused on the path
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchsummary"].summary = lambda *a, **k: None
sys.modules["tensorboardX"].SummaryWriter = _Writer
for pkg in ("models", "utils", "datasets"):                     # the reference's namespace packages (HF `datasets` would shadow its own)
    m = types.ModuleType(pkg)
    m.__path__ = [os.path.join(REF, pkg)]
    m.__spec__ = importlib.machinery.ModuleSpec(pkg, None, is_package=True)
    sys.modules[pkg] = m
sys.path.insert(0, REF)
if not hasattr(np, "Inf"):
    np.Inf = np.inf                                             # core_utils:61 (removed in NumPy 2)

record = {"mode": mode}
if mode == "dropin":
    import toad_amd
    from toad_amd import functional as F_, model_toad as dropin, ops
    from oracle import toad_oracle as orc                       # the device stand-in's arithmetic (test infrastructure)

    shim = types.ModuleType("models.model_toad")                # ---- Option B: the drop-in under the reference's module name
    shim.TOAD_fc_mtl_concat = dropin.TOAD_fc_mtl_concat
    shim.Attn_Net_Gated = dropin.Attn_Net_Gated
    sys.modules["models.model_toad"] = shim

    calls = {"relocate": 0, "mil_fwd": 0, "mil_fwd_nograd": 0, "mil_bwd": 0, "rows": 0}
    dropin._require_cuda = lambda t, what: None                 # no HIP device in the build container (see the docstring)

    def relocate(self):                                         # models/model_toad.py:77-88 without the device move
        calls["relocate"] += 1
        self.flatten_parameters()
    dropin.TOAD_fc_mtl_concat.relocate = relocate

    SLOT2KEY = dict(zip(F_.SLOTS, orc.PARAM_KEYS))

    def _params(w):
        d = w["wc"].shape[1]
        assert w["wab"].data_ptr() == w["wa"].data_ptr() and w["wab"].shape == (2 * d, 512), "[Wa;Wb] must be a zero-copy view of the flat buffer"
        assert w["bab"].data_ptr() == w["ba"].data_ptr()
        for k in ops.STEP_SLOTS:
            t = w[k]
            assert t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0, k
        return {SLOT2KEY[s]: w[s].detach() for s in F_.SLOTS}

    class _Arena:                                               # ops.MilArena's surface: .n .c .d and typed views by slot name
        def __init__(self, n, c, d, t):
            self.n, self.c, self.d, self.t = n, c, d, t

        def view(self, name, shape, dtype=torch.float32):
            v = self.t[name]
            assert v.dtype == dtype and v.numel() == int(np.prod(shape)), (name, v.shape, shape)
            return v.view(*shape)

    def mil_fwd(w, bag, sex, drop_p=0.0, seed=0, attention_only=False, x_amax=None, cached_arena=False):
        assert bag.dtype == torch.float32 and bag.is_contiguous() and bag.dim() == 2 and bag.shape[1] == 1024
        assert sex.dtype == torch.float32 and sex.shape == (1,) and drop_p == 0.0 and not attention_only
        calls["mil_fwd_nograd" if cached_arena else "mil_fwd"] += 1
        calls["rows"] += bag.shape[0]
        p = _params(w)
        with torch.no_grad():
            out, sv = orc.forward(p, bag, sex)
        t = {"h1": sv.h1, "h": sv.h, "p": sv.p, "a_raw": sv.a_raw.contiguous(), "m": sv.m, "mcat": sv.mcat.contiguous(), "logits": out["logits"],
             "y_prob": out["Y_prob"], "y_hat": out["Y_hat"].to(torch.int64), "site_logits": out["site_logits"], "site_prob": out["site_prob"],
             "site_hat": out["site_hat"].to(torch.int64)}
        return _Arena(bag.shape[0], p["classifier.weight"].shape[0], w["wc"].shape[1], {k: v.contiguous() for k, v in t.items()})

    def mil_bwd(w, grads, beta, bag, arena, dlogits, dsite, da_ext=None, dmcat_ext=None, drop_p=0.0, seed=0, need_dx=False, need_dsex=False):
        calls["mil_bwd"] += 1
        # (autograd materialises the gradients of the unused outputs A and features as zeros)
        assert beta == 0.0 and not need_dx and all(e is None or not e.any() for e in (da_ext, dmcat_ext)), "the reference's loss reaches the model through the two logits only"
        assert dlogits.shape == (1, arena.c) and dsite.shape == (1, 2) and dlogits.is_contiguous() and dsite.is_contiguous()
        assert set(grads) >= set(ops.STEP_SLOTS)
        p = _params(w)
        t = arena.t
        sv = orc.Saved(x=bag, h1=t["h1"], h=t["h"], p=t["p"], a_raw=t["a_raw"], m=t["m"], mcat=t["mcat"], sex=None)
        g = orc.backward(p, sv, dlogits, dsite)
        d = arena.d
        key = {v: k for k, v in SLOT2KEY.items()}
        for k, v in g.items():
            s = key[k]
            if s in ("wa", "wb", "ba", "bb"):
                dst = grads["wab" if s[0] == "w" else "bab"]
                (dst[:d] if s[1] == "a" else dst[d:]).copy_(v)
            else:
                assert grads[s].shape == v.shape, s
                grads[s].copy_(v)
        dsex = (dlogits @ p["classifier.weight"])[0, -1:] + (dsite @ p["site_classifier.weight"])[0, -1:]
        return None, (dsex if need_dsex else None)

    ops.mil_fwd, ops.mil_bwd = mil_fwd, mil_bwd
    record["calls"] = calls

os.makedirs(workdir, exist_ok=True)
os.chdir(workdir)
sys.argv = [os.path.join(REF, "main_mtl_concat.py")] + argv
ns = runpy.run_path(sys.argv[0], run_name="__main__")           # the reference's script, top to bottom, unmodified

if mode == "dropin":
    import utils.core_utils_mtl_concat as core                   # type: ignore
    assert core.TOAD_fc_mtl_concat is toad_amd.TOAD_fc_mtl_concat, "the harness did not bind to the drop-in"
    record["model_class"] = core.TOAD_fc_mtl_concat.__module__ + "." + core.TOAD_fc_mtl_concat.__name__
record["results_dir"] = ns["args"].results_dir
with open(os.path.join(workdir, "probe_%s.json" % mode), "w") as f:
    json.dump(record, f)
print("RUNPY_MAIN_OK", mode)
