"""Run by tests/test_reference_harness_dropin.py in a subprocess (build container only: needs /root/reference).

INTEGRATION.md "Option B": the reference's OWN harness modules import ``models.model_toad`` - pre-register a module of that name that exports
this repository's classes and the unmodified ``utils/core_utils_mtl_concat.py`` / ``utils/eval_utils_mtl_concat.py`` bind to the drop-in.
Everything of the harness that runs on the host is driven here through the reference's code: construction with the kwargs of
core_utils:114-116, print_network (utils/utils.py:72-84), get_optim (:63-70), EarlyStopping.save_checkpoint (core_utils:80-85),
initiate_model (eval_utils:19-32: construct, relocate, print_network, torch.load, load_state_dict(strict=False), eval()).
There is no GPU in the build container and toad_amd has no CPU fallback, so ``relocate()`` - which would move the parameters to the HIP
device - is replaced by a recorder for this probe and no forward is run; on a GPU box the same objects are driven by tests/test_gpu_*.py."""
import importlib.machinery
import os
import sys
import tempfile
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

import numpy as np
import torch

import toad_amd
from toad_amd import model_toad as dropin

for name in ("torchvision", "torchvision.transforms", "h5py", "tensorboardX", "torchsummary"):      # absent from the image, unused on the path
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchsummary"].summary = lambda *a, **k: None
for pkg in ("models", "utils", "datasets"):                     # the reference's namespace packages (HF `datasets` would shadow its own)
    m = types.ModuleType(pkg)
    m.__path__ = [os.path.join(REF, pkg)]
    m.__spec__ = importlib.machinery.ModuleSpec(pkg, None, is_package=True)
    sys.modules[pkg] = m
sys.path.insert(0, REF)
if not hasattr(np, "Inf"):
    np.Inf = np.inf                                             # core_utils:61 (removed in NumPy 2; SURVEY.md 8c gotcha 3)

# ---- Option B: the drop-in under the reference's module name
shim = types.ModuleType("models.model_toad")
shim.TOAD_fc_mtl_concat = dropin.TOAD_fc_mtl_concat
shim.Attn_Net_Gated = dropin.Attn_Net_Gated
sys.modules["models.model_toad"] = shim

import utils.core_utils_mtl_concat as core      # noqa: E402   (the reference's files, unmodified)
import utils.eval_utils_mtl_concat as ev        # noqa: E402
from utils.utils import get_optim, print_network  # noqa: E402

assert core.TOAD_fc_mtl_concat is toad_amd.TOAD_fc_mtl_concat and ev.TOAD_fc_mtl_concat is toad_amd.TOAD_fc_mtl_concat
print("BOUND core_utils and eval_utils to", core.TOAD_fc_mtl_concat.__module__)

relocated = []
dropin.TOAD_fc_mtl_concat.relocate = lambda self: relocated.append(type(self).__name__)      # no HIP device here (see the docstring)

args = types.SimpleNamespace(drop_out=True, n_classes=18, opt="adam", lr=1e-4, reg=1e-5)
model_dict = {"dropout": args.drop_out, "n_classes": args.n_classes}                          # core_utils:114
torch.manual_seed(1)
model = core.TOAD_fc_mtl_concat(**model_dict)                                                   # core_utils:116
model.relocate()                                                                                # core_utils:118
print_network(model)                                                                            # core_utils:120
n_params = sum(p.numel() for p in model.parameters())
assert n_params == 1192490, n_params
for opt_name, cls in (("adam", torch.optim.Adam), ("sgd", torch.optim.SGD)):
    args.opt = opt_name
    opt = get_optim(model, args)                                                                # core_utils:123
    assert isinstance(opt, cls) and sum(len(g["params"]) for g in opt.param_groups) == 14
    assert opt.param_groups[0]["lr"] == args.lr and opt.param_groups[0]["weight_decay"] == args.reg
print("OPTIM ok")

with tempfile.TemporaryDirectory() as d:
    ck = os.path.join(d, "s_0_checkpoint.pt")
    stopper = core.EarlyStopping(patience=2, stop_epoch=0, verbose=True)                        # the reference's class, not a port
    stopper(0, 1.25, model, ckpt_name=ck)                                                       # first call saves (core_utils:70-72)
    with torch.no_grad():
        next(model.parameters()).add_(1.0)
    stopper(1, 1.50, model, ckpt_name=ck)                                                       # worse: no save, counter 1
    assert stopper.counter == 1 and not stopper.early_stop and os.path.exists(ck)
    saved = torch.load(ck)
    ref_keys = [ln.split("|")[2] for ln in str(np.load(os.path.join(REPO, "tests", "golden", "toad_golden.npz"))["api/state_dict"]).split("\n")
                if ln.startswith("2|1|")]                                                       # the reference's own keys for dropout=True
    assert list(saved) == ref_keys, "state-dict keys / order differ from the reference's (dropout=True layout)"
    print("CHECKPOINT keys ok:", len(saved))
    # eval side: initiate_model builds a fresh drop-in, relocates, prints, loads with strict=False, switches to eval (eval_utils:19-32)
    m2 = ev.initiate_model(types.SimpleNamespace(drop_out=True, n_classes=18), ck)
    assert isinstance(m2, toad_amd.TOAD_fc_mtl_concat) and not m2.training and relocated == ["TOAD_fc_mtl_concat"] * 2
    for (k, a), (_, b) in zip(saved.items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # a checkpoint written by the reference on a multi-GPU box carries nn.DataParallel's `module.` infix (model_toad.py:79-81); the
    # reference masks the mismatch with strict=False (eval_utils:29) and silently evaluates an untrained trunk - the drop-in renames
    infixed = {k.replace("attention_net.", "attention_net.module.", 1): v for k, v in saved.items()}
    torch.save(infixed, ck)
    m3 = ev.initiate_model(types.SimpleNamespace(drop_out=True, n_classes=18), ck)
    for (k, a), (_, b) in zip(saved.items(), m3.state_dict().items()):
        assert torch.equal(a, b), ("module.-infixed checkpoint", k)
    print("INITIATE_MODEL ok")
print("OPTION_B_OK")
