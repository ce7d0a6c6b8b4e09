"""Optimiser for the flat parameter buffer: one HIP launch per step (toad_adam_step_f32).

Same update rule and defaults as the reference's ``get_optim`` Adam branch (utils/utils.py:63-70:
``optim.Adam(params, lr=args.lr, weight_decay=args.reg)``); the reference's SGD branch and any other
torch optimiser keep working on ``model.parameters()`` unchanged.
"""
from __future__ import annotations

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

    def step(self, flat_grad: torch.Tensor) -> None:
        if not self.p.is_cuda:
            raise RuntimeError("FlatAdam runs on the HIP device only")
        self.t += 1
        lib = _lib.load()
        _lib.check(lib.toad_adam_step_f32(self.p.data_ptr(), flat_grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                          self.p.numel(), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                                          float(self.eps), float(self.weight_decay), self.t,
                                          torch.cuda.current_stream().cuda_stream), "toad_adam_step_f32")

    def state_dict(self):
        return {"m": self.m, "v": self.v, "t": self.t, "lr": self.lr, "betas": self.betas, "eps": self.eps,
                "weight_decay": self.weight_decay}
