"""Data-parallel (view-parallel) training over the GPUs of one node — SURVEY.md §8e.

New capability relative to the reference (which is single-GPU: ``scripts/shells/train.sh:6``;
``FullImageDatamanager`` stores ``world_size`` but never uses it, ``sgn_datamanager.py:79-86``).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` in the
CPU tests).  Parameters are replicated; rank ``r`` renders camera ``perm[world*step + r]`` of a
seed-synchronised permutation (replaces the ``random.randint`` pop at
``sgn_datamanager.py:281``); the only exchange step is a SUM all-reduce of the per-Gaussian
gradients (59 floats = 236 B per Gaussian at SH degree 3).  xGMI is point-to-point, so a ring
all-reduce is per-link bound: the big SH gradient (192 of the 236 B) is reduced *as soon as its
autograd node finishes*, overlapping ``project_gaussians`` backward; the four small tensors go
out as one flat bucket after backward.  Densification statistics are reduced (SUM/SUM/MAX) so
replicas take bit-identical split/dup/cull decisions.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun).
    Returns (rank, world, local_rank).  No-op single-process fallback when WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def view_permutation(n_views: int, seed: int, epoch: int) -> torch.Tensor:
    """Same shuffled camera order on every rank (seed-synchronised)."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + epoch)
    return torch.randperm(n_views, generator=g)


def view_for_rank(step: int, rank: int, world: int, n_views: int, seed: int = 0) -> int:
    """Camera index rank ``rank`` renders at ``step``: ``perm[world*step + rank]`` with a fresh
    permutation per epoch; over one epoch every view is rendered exactly once across ranks."""
    flat = world * step + rank
    epoch, pos = divmod(flat, n_views)
    return int(view_permutation(n_views, seed, epoch)[pos])


class GradAllReducer:
    """Bucketed, overlapped all-reduce of per-Gaussian gradients.

    ``big`` parameters (the SH coefficients) get a post-accumulate-grad hook that launches an
    async all-reduce the moment their gradient is final; everything else is flattened into one
    bucket in :meth:`finish`.  ``average=True`` divides by world size (the loss is a per-image
    mean, so DP over views averages)."""

    def __init__(self, params: Sequence[torch.Tensor], big: Iterable[torch.Tensor] = (),
                 average: bool = True, group=None):
        self.params = list(params)
        self.big_ids = {id(p) for p in big}
        self.average = average
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending: List = []
        self._handles = []
        if self.world > 1:
            for p in self.params:
                if id(p) in self.big_ids:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def _hook(self, p: torch.Tensor) -> None:
        work = dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, p))

    def finish(self) -> None:
        """Call after ``loss.backward()``: reduces the small bucket, waits for the async ones."""
        if self.world == 1:
            return
        small = [p for p in self.params if id(p) not in self.big_ids and p.grad is not None]
        if small:
            flat = torch.cat([p.grad.reshape(-1) for p in small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                flat /= self.world
            off = 0
            for p in small:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        for work, p in self._pending:
            work.wait()
            if self.average:
                p.grad /= self.world
        self._pending.clear()

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles.clear()


def sync_densify_stats(xys_grad_norm: torch.Tensor, vis_counts: torch.Tensor, max_2dsize: torch.Tensor,
                       group=None) -> None:
    """In-place SUM / SUM / MAX all-reduce of the statistics ``refinement_after`` consumes
    (``sgn_splatfacto.py:513-541,550-646``) so every replica splits/dups/culls identically."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(xys_grad_norm, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(vis_counts, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_2dsize, op=dist.ReduceOp.MAX, group=group)


def broadcast_params(params: Dict[str, torch.Tensor], src: int = 0, group=None) -> None:
    """Make replicas bit-identical at start / after a checkpoint load."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for k in sorted(params):
        dist.broadcast(params[k].data, src=src, group=group)
