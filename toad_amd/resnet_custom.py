"""Drop-in for the reference's ``models/resnet_custom.py``: the truncated ResNet-50 that turns 256x256 tiles into the
1024-d patch embeddings of a bag, running on the HIP kernels (``toad_resnet50_trunc_fwd_f32``).

Same classes, constructor arguments, module tree and state-dict keys as the reference (``Bottleneck_Baseline``
:19-56, ``ResNet_Baseline`` :58-108, ``resnet50_baseline`` :111-119), so torchvision ResNet-50 checkpoints load exactly
like they do there (``load_state_dict(..., strict=False)``, :121-124). The ``nn.Conv2d`` / ``nn.BatchNorm2d`` children
are parameter containers only: ``forward`` folds eval-mode BatchNorm into the convolution weights once (fp64 on the
device), lays each convolution out as the [Cout, kh*kw*Cin] operand of the NHWC GEMM, and hands the whole network
to one C-ABI call. There is no CPU path and no train-mode BatchNorm: the reference only ever runs the extractor under
``eval()`` to write the ``.pt`` bags (it is never optimised), and this module refuses anything else loudly.

``pretrained=True`` needs the network in the reference too (model_zoo download, :122); here it raises with that
explanation — load a local checkpoint with ``load_state_dict`` instead.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib

__all__ = ["Bottleneck_Baseline", "ResNet_Baseline", "resnet50_baseline"]

STEM_K = 192            # 4 x 4 space-to-depth taps x 12 channels (147 real taps + zero slots)
MAX_TILES_PER_CALL = 512     # at 256x256 tiles. 32-bit byte offsets inside the kernels: B*(H/4)*(W/4)*128*4 < 2^31 (layer2.0 conv2 input)


def max_tiles_per_call(h: int, w: int) -> int:
    """Tiles per C call for H x W tiles: the kernels' 32-bit offsets bound B * H * W, so the cap scales with the tile area
    (512 at the reference's 256 x 256; 128 at 512 x 512; more for smaller tiles, bounded by the workspace to 4096)."""
    return max(1, min(4096, (MAX_TILES_PER_CALL * 256 * 256) // max(h * w, 1)))


class Bottleneck_Baseline(nn.Module):
    """Parameter container of one bottleneck block, children named and shaped as in the reference (resnet_custom.py:19-33):
    conv1 1x1 -> bn1 -> conv2 3x3 (carries the stride) -> bn2 -> conv3 1x1 (x4 channels) -> bn3, optional downsample."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        out = planes * self.expansion
        for i, (cin, cout, k, s) in enumerate(((inplanes, planes, 1, 1), (planes, planes, 3, stride), (planes, out, 1, 1)), start=1):
            self.add_module(f"conv{i}", nn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=k // 2, bias=False))
            self.add_module(f"bn{i}", nn.BatchNorm2d(cout))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("Bottleneck_Baseline runs only inside ResNet_Baseline.forward (one fused HIP call)")


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d, stem: bool):
    """eval-mode BN folded into the convolution: W' = W * g/sqrt(var+eps), b' = beta - mean * g/sqrt(var+eps)."""
    w = conv.weight.detach().double()
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    w = w * scale.view(-1, 1, 1, 1)
    if stem:                                            # space-to-depth operand of toad_stem_conv_s2d_f32: [64, 4(qy), 4(qx), 2(ry), 2(rx), 3(c)]
        w8 = torch.zeros(w.shape[0], 3, 8, 8, dtype=torch.float64, device=w.device)
        w8[:, :, 1:, 1:] = w                            # slot (2q + r) holds tap 2q + r - 1; slot 0 (tap -1) stays zero
        w2 = w8.view(w.shape[0], 3, 4, 2, 4, 2).permute(0, 2, 4, 3, 5, 1).reshape(w.shape[0], STEM_K)
    else:                                               # (ky, kx, c) order = the NHWC gather's column order
        w2 = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    return w2.float().contiguous(), shift.float().contiguous()


class ResNet_Baseline(nn.Module):
    def __init__(self, block, layers):
        super().__init__()
        self.inplanes = 64
        if block is not Bottleneck_Baseline or list(layers[:3]) != [3, 4, 6]:
            raise NotImplementedError("the HIP extractor implements resnet50_baseline: Bottleneck_Baseline, layers [3, 4, 6, (3)]")
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        for m in self.modules():                                            # reference init, :70-75
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._folded = None           # (weights, biases, ctypes arrays), built lazily on the device
        self._folded_sig = None
        self._ws: Optional[torch.Tensor] = None

    def _make_layer(self, block, planes, blocks, stride=1):
        """`blocks` bottlenecks; the first one carries the stride and, when the shape changes, the 1x1 projection of the skip."""
        width = planes * block.expansion
        proj = None
        if stride != 1 or self.inplanes != width:
            proj = nn.Sequential(nn.Conv2d(self.inplanes, width, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(width))
        stack = [block(self.inplanes, planes, stride, proj)] + [block(width, planes) for _ in range(blocks - 1)]
        self.inplanes = width
        return nn.Sequential(*stack)

    # ---- folded-weight cache ------------------------------------------------------------------------------------
    def refold(self) -> None:
        """Drop the folded weights (call after changing parameters in place; load_state_dict / .to() do it themselves)."""
        self._folded = None

    def load_state_dict(self, *args, **kwargs):
        self._folded = None
        return super().load_state_dict(*args, **kwargs)

    def _param_signature(self):
        """(storage address, in-place version) of every parameter and BN statistic: changes whenever weights are loaded
        (also as a child of a parent module, which bypasses this class's load_state_dict), moved, or edited in place."""
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _apply(self, fn, *args, **kwargs):
        self._folded = None
        self._ws = None
        return super()._apply(fn, *args, **kwargs)

    def _conv_bn_pairs(self):
        yield self.conv1, self.bn1, True
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                yield blk.conv1, blk.bn1, False
                yield blk.conv2, blk.bn2, False
                yield blk.conv3, blk.bn3, False
                if blk.downsample is not None:
                    yield blk.downsample[0], blk.downsample[1], False

    def _fold_all(self):
        ws: List[torch.Tensor] = []
        bs: List[torch.Tensor] = []
        for conv, bn, stem in self._conv_bn_pairs():
            w, b = _fold(conv, bn, stem)
            ws.append(w); bs.append(b)
        assert len(ws) == 43
        wp = (ctypes.c_void_p * 43)(*[t.data_ptr() for t in ws])
        bp = (ctypes.c_void_p * 43)(*[t.data_ptr() for t in bs])
        self._folded = (ws, bs, wp, bp)
        self._folded_sig = self._param_signature()

    def relocate(self):
        if not torch.cuda.is_available():
            raise RuntimeError("toad_amd.resnet_custom needs a HIP device (no CPU fallback)")
        return self.to(torch.device("cuda", torch.cuda.current_device()))

    # ---- forward ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] fp32 NCHW on the HIP device -> [B,1024] (ResNet_Baseline.forward, :95-108)."""
        if self.training:
            raise RuntimeError("the HIP extractor is inference-only: call .eval() (the reference never trains it)")
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("expected a float32 [B,3,H,W] tensor on the HIP device (no CPU fallback)")
        if self.conv1.weight.device != x.device:
            raise RuntimeError("model and input are on different devices; call model.relocate()")
        lib = _lib.load()
        if self._folded is None or self._folded_sig != self._param_signature():
            self._fold_all()
        _, _, wp, bp = self._folded
        x = x.contiguous()
        B, _, H, W = x.shape
        out = torch.empty(B, 1024, device=x.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        cap = max_tiles_per_call(H, W)
        for b0 in range(0, B, cap):
            nb = min(cap, B - b0)
            need = lib.toad_resnet50_trunc_ws_bytes(nb, H, W)
            if need == 0:
                raise RuntimeError(f"unsupported tile shape {H}x{W}")
            if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
                self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            _lib.check(lib.toad_resnet50_trunc_fwd_f32(x[b0:b0 + nb].data_ptr(), wp, bp, out[b0:b0 + nb].data_ptr(), nb, H, W,
                                                       self._ws.data_ptr(), self._ws.numel(), stream), "toad_resnet50_trunc_fwd_f32")
        return out


def resnet50_baseline(pretrained: bool = False) -> ResNet_Baseline:
    """Constructs the modified (truncated) ResNet-50 (reference :111-119)."""
    model = ResNet_Baseline(Bottleneck_Baseline, [3, 4, 6, 3])
    if pretrained:
        raise RuntimeError("pretrained=True downloads ImageNet weights in the reference (model_zoo, :122); there is no network here - "
                           "load a local torchvision resnet50 checkpoint with model.load_state_dict(sd, strict=False)")
    return model
