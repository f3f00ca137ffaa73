"""Multi-tensor Adam for the per-Gaussian parameter groups.

The reference builds one ``torch.optim.Adam`` per parameter group through nerfstudio's ``AdamOptimizerConfig``
(``street_gaussians_ns/sgn_config.py:71-108``: xyz, features_dc, features_rest, opacity, scaling, rotation, sky_sphere;
eps 1e-15; the scene graph multiplies that by the number of sub-models) and steps them one after the other.
:class:`FusedAdam` is a ``torch.optim.Optimizer`` with Adam's constructor and **state layout** (``state[p]['step']``,
``['exp_avg']``, ``['exp_avg_sq']`` — the tensors the reference's densification rewrites in place,
``sgn_splatfacto.py:459-511``), whose ``step()`` updates every parameter of every group in ONE kernel launch
(``sgn_adam_step``, ``csrc/optim.hip``).  :func:`step_many` does the same across several optimizer objects (the
reference's one-optimizer-per-group layout).  No CPU fallback: parameters must live on the ROCm device.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List

import torch

from . import _lib as L


def _launch(rows: List[tuple]) -> None:
    """rows: (param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step)."""
    if not rows:
        return
    n = len(rows)
    ptrs = lambda k: (C.c_void_p * n)(*[r[k].data_ptr() for r in rows])
    f32 = lambda k: (C.c_double * n)(*[float(r[k]) for r in rows])   # doubles: torch's Python scalars
    numel = (C.c_int64 * n)(*[r[0].numel() for r in rows])
    steps = (C.c_int64 * n)(*[int(r[8]) for r in rows])
    L.check(L.load().sgn_adam_step(n, ptrs(0), ptrs(1), ptrs(2), ptrs(3), numel, f32(4), f32(5), f32(6), f32(7), steps,
                                   L.stream_ptr()), "sgn_adam_step")


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr, betas, eps)`` (no weight decay / amsgrad / maximize — the reference uses none)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False, maximize: bool = False):
        if weight_decay != 0.0 or amsgrad or maximize:
            raise NotImplementedError("FusedAdam implements the configuration the reference uses: plain Adam")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    def _rows(self) -> List[tuple]:
        rows = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                L.require_device(p, p.grad)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("FusedAdam needs contiguous fp32 parameters")
                st = self.state[p]
                if len(st) == 0:                      # same lazy initialisation as torch.optim.Adam
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                rows.append((p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2,
                             group["eps"], int(st["step"].item())))
        return rows

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _launch(self._rows())
        return loss


@torch.no_grad()
def step_many(optimizers: Iterable[FusedAdam]) -> None:
    """One launch for several :class:`FusedAdam` objects (nerfstudio keeps one optimizer per parameter group)."""
    rows: List[tuple] = []
    for opt in optimizers:
        rows.extend(opt._rows())
    _launch(rows)
