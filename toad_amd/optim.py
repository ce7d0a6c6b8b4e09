"""Optimisers for the flat parameter buffer: one HIP launch per step.

``FlatAdam`` / ``FlatSGD`` apply the update rules and defaults of the reference's ``get_optim`` (utils/utils.py:63-70:
``optim.Adam(params, lr=args.lr, weight_decay=args.reg)`` and ``optim.SGD(params, lr=args.lr, momentum=0.9,
weight_decay=args.reg)``) to all 1.19 M parameters at once; any torch optimiser keeps working on ``model.parameters()``
unchanged (the drop-in path).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class FlatAdam:
    def __init__(self, flat_param: torch.Tensor, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-5):
        if flat_param.dim() != 1 or flat_param.numel() % 4 != 0:
            raise ValueError("FlatAdam needs a 1-D buffer whose length is a multiple of 4")
        self.p = flat_param
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        self.t = 0
        self.grad: Optional[torch.Tensor] = None       # bound gradient buffer for the argument-less step() of the harness loop

    def zero_grad(self, set_to_none: bool = True) -> None:
        """No-op: the fused slide step overwrites the gradient buffer (beta = 0); kept so that the reference's loop body
        (``optimizer.step(); optimizer.zero_grad()``, utils/core_utils_mtl_concat.py:233-234) runs unchanged."""

    def step(self, flat_grad: Optional[torch.Tensor] = None) -> None:
        if not self.p.is_cuda:
            raise RuntimeError("FlatAdam runs on the HIP device only")
        flat_grad = self.grad if flat_grad is None else flat_grad
        if flat_grad is None:
            raise RuntimeError("FlatAdam.step: no gradient buffer (pass one or bind it with .grad = buffer)")
        self.t += 1
        lib = _lib.load()
        _lib.check(lib.toad_adam_step_f32(self.p.data_ptr(), flat_grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                          self.p.numel(), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                                          float(self.eps), float(self.weight_decay), self.t,
                                          torch.cuda.current_stream().cuda_stream), "toad_adam_step_f32")

    def state_dict(self):
        return {"m": self.m, "v": self.v, "t": self.t, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                "weight_decay": self.weight_decay}

    def load_state_dict(self, sd) -> None:
        """Restore the moments and the step count (checkpoint resume); hyper-parameters follow the checkpoint."""
        if sd["m"].numel() != self.p.numel() or sd["v"].numel() != self.p.numel():
            raise ValueError("FlatAdam.load_state_dict: moment buffers do not match the flat parameter buffer")
        self.m.copy_(sd["m"].to(self.m.device).reshape(-1))
        self.v.copy_(sd["v"].to(self.v.device).reshape(-1))
        self.t = int(sd["t"])
        self.lr, self.betas, self.eps = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"])
        self.weight_decay = float(sd["weight_decay"])


class FlatSGD:
    """torch.optim.SGD(lr, momentum, weight_decay) over the flat buffer (toad_sgd_step_f32): get_optim's SGD branch."""

    def __init__(self, flat_param: torch.Tensor, lr: float = 1e-4, momentum: float = 0.9, weight_decay: float = 1e-5):
        if flat_param.dim() != 1 or flat_param.numel() % 4 != 0:
            raise ValueError("FlatSGD needs a 1-D buffer whose length is a multiple of 4")
        self.p = flat_param
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.buf = torch.zeros_like(flat_param) if momentum != 0.0 else None
        self.t = 0
        self.grad: Optional[torch.Tensor] = None

    def zero_grad(self, set_to_none: bool = True) -> None:
        """No-op (see FlatAdam.zero_grad)."""

    def step(self, flat_grad: Optional[torch.Tensor] = None) -> None:
        if not self.p.is_cuda:
            raise RuntimeError("FlatSGD runs on the HIP device only")
        flat_grad = self.grad if flat_grad is None else flat_grad
        if flat_grad is None:
            raise RuntimeError("FlatSGD.step: no gradient buffer (pass one or bind it with .grad = buffer)")
        self.t += 1
        lib = _lib.load()
        _lib.check(lib.toad_sgd_step_f32(self.p.data_ptr(), flat_grad.data_ptr(), None if self.buf is None else self.buf.data_ptr(),
                                         self.p.numel(), float(self.lr), float(self.momentum), float(self.weight_decay), self.t,
                                         torch.cuda.current_stream().cuda_stream), "toad_sgd_step_f32")

    def state_dict(self):
        return {"buf": self.buf, "t": self.t, "lr": self.lr, "momentum": self.momentum, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd) -> None:
        if self.buf is not None and sd["buf"] is not None:
            self.buf.copy_(sd["buf"].to(self.buf.device).reshape(-1))
        self.t = int(sd["t"])
        self.lr, self.momentum, self.weight_decay = float(sd["lr"]), float(sd["momentum"]), float(sd["weight_decay"])


def get_optim(model, args, flat: bool = True):
    """``get_optim`` of the reference (utils/utils.py:63-70): ``args.opt`` in {"adam", "sgd"}, ``args.lr``, ``args.reg``.
    ``flat`` (default): the one-launch HIP optimiser over the model's flat parameter buffer, which ``toad_amd.train.train_loop``
    drives with the fused slide step; ``flat=False``: the reference's torch optimiser over ``model.parameters()``."""
    if args.opt not in ("adam", "sgd"):
        raise NotImplementedError
    if not flat:
        params = filter(lambda p: p.requires_grad, model.parameters())
        if args.opt == "adam":
            return torch.optim.Adam(params, lr=args.lr, weight_decay=args.reg)
        return torch.optim.SGD(params, lr=args.lr, momentum=0.9, weight_decay=args.reg)
    buf = model.flat_parameters()
    if args.opt == "adam":
        return FlatAdam(buf, lr=args.lr, weight_decay=args.reg)
    return FlatSGD(buf, lr=args.lr, momentum=0.9, weight_decay=args.reg)
