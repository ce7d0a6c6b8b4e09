"""TEST INFRASTRUCTURE ONLY — pins oracle/resnet_oracle.py to the real reference extractor and emits goldens.

Runs ONLY in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/pin_resnet_against_reference.py [--write]

Imports the unmodified ``models/resnet_custom.py`` (``resnet50_baseline``, :111-119; `torchsummary` stubbed — it
is imported at :5 and never used), loads the seeded state dict of ``resnet_oracle.make_params`` (strict), runs it
under eval()/no_grad on seeded tiles in fp32 and in fp64, and checks the oracle against it. With --write the
REFERENCE's features go to tests/golden/resnet_golden.npz together with its own fp32-vs-fp64 deviation (the
yardstick for what "equal in fp32" means after 43 stacked convolutions). Inputs and weights are regenerated
from their seeds by the tests; nothing of the reference travels.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import resnet_oracle as ro  # noqa: E402
from oracle.pin_against_reference import import_reference  # noqa: E402

# (name, batch, H, W, weight seed, tile seed)
CASES = [
    ("b2_64", 2, 64, 64, 11, 101),
    ("b3_96x64", 3, 96, 64, 11, 102),        # non-square
    ("b1_100", 1, 100, 100, 12, 103),        # odd intermediate sizes: 100 -> 50 -> 25 -> 13 -> 7
    ("b1_33", 1, 33, 33, 12, 104),           # 33 -> 17 -> 9 -> 5 -> 3
    ("b2_256", 2, 256, 256, 11, 105),        # the extractor's real tile size
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    import_reference()
    sys.modules["torchsummary"].summary = lambda *a, **k: None
    from models.resnet_custom import resnet50_baseline          # type: ignore

    w = {}
    for name, b, h, wd, wseed, xseed in CASES:
        sd = ro.make_params(wseed)
        x = ro.make_tiles(b, h, wd, xseed)
        ref = resnet50_baseline(pretrained=False)
        ref.load_state_dict(sd, strict=True)
        ref.eval()
        with torch.no_grad():
            f32 = ref(x)
        ref64 = resnet50_baseline(pretrained=False).double()
        ref64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, strict=True)
        ref64.eval()
        with torch.no_grad():
            f64 = ref64(x.double())
        o = ro.forward(sd, x)
        scale = f64.abs().max().item()
        e_or = (o - f32).abs().max().item()
        dev = (f32.double() - f64).abs().max().item()
        print(f"  {name:10s} B={b} {h}x{wd}: |feat|max {scale:.3e}  oracle-vs-ref {e_or:.2e}  ref32-vs-ref64 {dev:.2e}  "
              f"(relative {e_or / scale:.1e} / {dev / scale:.1e})")
        assert f32.shape == (b, 1024)
        assert e_or <= 1e-4 * max(scale, 1.0), (name, e_or)
        w[name + "/meta"] = np.array([b, h, wd, wseed, xseed], dtype=np.int64)
        w[name + "/feat"] = f32.numpy().astype(np.float32)
        w[name + "/feat64"] = f64.numpy()
        w[name + "/dev64"] = np.float64(dev)
    if args.write:
        out = os.path.join(REPO, "tests", "golden", "resnet_golden.npz")
        np.savez_compressed(out, **w)
        print("  wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
