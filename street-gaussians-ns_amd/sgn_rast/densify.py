"""Per-step densification statistics — ``SplatfactoModel.after_train``
(``street_gaussians_ns/sgn_splatfacto.py:513-541``) as one HIP pass.

The reference updates ``xys_grad_norm`` / ``vis_counts`` / ``max_2Dsize`` every training step with boolean-mask
indexing (``t[mask] = t[mask] + ...``: nonzero + gather + index_put per line, each with a host sync).  :class:`Stats`
keeps the same three tensors (same names, same values) and updates them with ``sgn_densify_stats``; under data
parallelism :meth:`Stats.sync` reduces them SUM / SUM / MAX so every replica takes identical split / dup / cull
decisions in ``refinement_after`` (``:550-646``), which is left to the caller — it is control-plane torch code that
runs every ``refine_every`` steps.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L


class Stats:
    def __init__(self):
        self.xys_grad_norm: Optional[torch.Tensor] = None
        self.vis_counts: Optional[torch.Tensor] = None
        self.max_2Dsize: Optional[torch.Tensor] = None

    def reset(self) -> None:                 # end of refinement_after (:644-646)
        self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None

    @torch.no_grad()
    def update(self, xys_grad: torch.Tensor, radii: torch.Tensor, last_size) -> None:
        """``xys_grad`` = ``self.xys.grad`` [N,2]; ``radii`` [N] int; ``last_size`` = (H, W) of the render."""
        L.require_device(xys_grad, radii)
        n = xys_grad.shape[0]
        g = xys_grad.detach().contiguous().float()
        r = radii.detach().to(torch.int32).contiguous()
        first = self.xys_grad_norm is None
        if first:
            f32 = dict(dtype=torch.float32, device=g.device)
            self.xys_grad_norm, self.vis_counts = torch.empty(n, **f32), torch.empty(n, **f32)
            self.max_2Dsize = torch.empty(n, **f32)
        L.check(L.load().sgn_densify_stats(n, L.ptr(g), L.ptr(r), float(max(last_size[0], last_size[1])), int(first),
                                           L.ptr(self.xys_grad_norm), L.ptr(self.vis_counts), L.ptr(self.max_2Dsize),
                                           L.stream_ptr()), "sgn_densify_stats")

    def sync(self, group=None) -> None:
        from .dp import sync_densify_stats
        if self.xys_grad_norm is not None:
            sync_densify_stats(self.xys_grad_norm, self.vis_counts, self.max_2Dsize, group=group)
