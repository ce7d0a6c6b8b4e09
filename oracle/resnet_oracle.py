"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's truncated ResNet-50 feature extractor.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
(toad_amd/) never does.

Follows /root/reference/models/resnet_custom.py:
  * Bottleneck_Baseline.forward (:35-56): 1x1 conv -> BN -> ReLU -> 3x3 conv (stride s, pad 1) -> BN -> ReLU ->
    1x1 conv (x4 channels) -> BN, plus the residual (through downsample = strided 1x1 conv + BN when the shape
    changes, :79-85), then ReLU.
  * ResNet_Baseline (:58-108): 7x7/2 conv (pad 3) -> BN -> ReLU -> 3x3/2 max-pool (pad 1) -> layer1 (3 blocks,
    64 planes) -> layer2 (4 blocks, 128 planes, stride 2) -> layer3 (6 blocks, 256 planes, stride 2) ->
    AdaptiveAvgPool2d(1) -> flatten: [B, 3, H, W] -> [B, 1024]. (layers[3] of [3,4,6,3] is never built.)
  * resnet50_baseline (:111-119).
Batch-norm is the inference form (running statistics): the extractor is only ever run under eval() / no_grad
to produce the .pt bags the MIL model trains on. Pinned against the imported reference by
oracle/pin_resnet_against_reference.py (golden: tests/golden/resnet_golden.npz).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

LAYERS: Tuple[Tuple[int, int, int], ...] = ((64, 3, 1), (128, 4, 2), (256, 6, 2))    # (planes, blocks, stride)
EPS = 1e-5                                                                             # nn.BatchNorm2d default


def conv_specs() -> List[Tuple[str, str, int, int, int, int, int]]:
    """(conv key, bn key, Cin, Cout, kernel, stride, pad) of all 43 convolutions, in execution order."""
    out = [("conv1", "bn1", 3, 64, 7, 2, 3)]
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(LAYERS, start=1):
        for b in range(blocks):
            s = stride if b == 0 else 1
            pre = f"layer{li}.{b}."
            out.append((pre + "conv1", pre + "bn1", inplanes, planes, 1, 1, 0))
            out.append((pre + "conv2", pre + "bn2", planes, planes, 3, s, 1))
            out.append((pre + "conv3", pre + "bn3", planes, planes * 4, 1, 1, 0))
            if b == 0 and (s != 1 or inplanes != planes * 4):
                out.append((pre + "downsample.0", pre + "downsample.1", inplanes, planes * 4, 1, s, 0))
            inplanes = planes * 4
    return out


def make_params(seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """A full state dict (reference key names) from one CPU generator seed: Kaiming-normal(fan_out) conv weights
    like the reference's init (:70-72), and NON-trivial batch-norm affine / running statistics so that folding
    mistakes cannot hide (the reference's init of 1/0 with fresh running stats would make BN the identity)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for conv, bn, cin, cout, k, _, _ in conv_specs():
        std = (2.0 / (cout * k * k)) ** 0.5
        sd[conv + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std
        sd[bn + ".weight"] = 0.6 + 0.5 * torch.rand(cout, generator=g)
        sd[bn + ".bias"] = 0.1 * torch.randn(cout, generator=g)
        sd[bn + ".running_mean"] = 0.1 * torch.randn(cout, generator=g)
        sd[bn + ".running_var"] = 0.6 + 0.8 * torch.rand(cout, generator=g)
        sd[bn + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def make_tiles(b: int, h: int, w: int, seed: int) -> torch.Tensor:
    """Synthetic normalised RGB tiles [b, 3, h, w] (what transforms.Normalize hands the extractor)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(b, 3, h, w, generator=g)


def _bn(x, sd, key):
    scale = sd[key + ".weight"] / torch.sqrt(sd[key + ".running_var"] + EPS)
    shift = sd[key + ".bias"] - sd[key + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def _bottleneck(x, sd, pre, stride, has_down):
    out = torch.relu(_bn(F.conv2d(x, sd[pre + "conv1.weight"]), sd, pre + "bn1"))                       # :38-40
    out = torch.relu(_bn(F.conv2d(out, sd[pre + "conv2.weight"], stride=stride, padding=1), sd, pre + "bn2"))   # :42-44
    out = _bn(F.conv2d(out, sd[pre + "conv3.weight"]), sd, pre + "bn3")                                 # :46-47
    res = x
    if has_down:
        res = _bn(F.conv2d(x, sd[pre + "downsample.0.weight"], stride=stride), sd, pre + "downsample.1")   # :49-50
    return torch.relu(out + res)                                                                        # :52-53


@torch.no_grad()
def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, return_stages: bool = False):
    """[B,3,H,W] -> [B,1024] (ResNet_Baseline.forward, :95-108)."""
    stages = {}
    x = torch.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    stages["stem"] = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    stages["pool"] = x
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(LAYERS, start=1):
        for b in range(blocks):
            s = stride if b == 0 else 1
            x = _bottleneck(x, sd, f"layer{li}.{b}.", s, b == 0 and (s != 1 or inplanes != planes * 4))
            inplanes = planes * 4
        stages[f"layer{li}"] = x
    feat = x.mean(dim=(2, 3))
    return (feat, stages) if return_stages else feat
