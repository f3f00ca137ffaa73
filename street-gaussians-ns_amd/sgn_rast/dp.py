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

Optionally (``SHGradExchange``) the SH gradient is not all-reduced at all: its low-rank factors (3-float colour
gradient + view direction or camera position) are all-gathered and the summed dense gradient is rebuilt on every
rank — 2-4x fewer bytes on the links, which is what takes view-parallel scaling past 6x at 8 GPUs when the
compute step is only ~2-3 ms.
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
                 average: bool = True, group=None, sh_exchange: "Optional[SHGradExchange]" = None,
                 force: bool = False):
        self.sh_exchange = sh_exchange
        skip = sh_exchange.leaf_ids() if sh_exchange is not None else set()
        self.params = [p for p in params if id(p) not in skip]
        self.big_ids = {id(p) for p in big if id(p) not in skip}
        self.average = average
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: run the collectives at world 1
        self._pending: List = []
        self._handles = []
        if self.active:
            for p in self.params:
                if id(p) in self.big_ids:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    def _hook(self, p: torch.Tensor) -> None:
        work = dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, p))

    def finish(self) -> None:
        """Call after ``loss.backward()``: reduces the small bucket, waits for the async ones.  The bucket's
        all-reduce is launched first and runs on the collective stream while the SH exchange rebuilds the dense SH
        gradient on the compute stream."""
        small, flat, work = [], None, None
        if self.sh_exchange is not None:
            self.sh_exchange.start_if_silent()      # its all-gathers come first on every rank (the taps fire in backward)
        if self.active:
            # every rank must make the same collective calls in the same order: a parameter that received no gradient
            # on this rank (its view saw nothing) takes part with zeros, and a "big" one whose hook therefore never
            # fired is reduced here, before the bucket, where the other ranks' hooks put it
            pending_ids = {id(p) for _w, p in self._pending}
            for p in self.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                if id(p) in self.big_ids and id(p) not in pending_ids:
                    self._pending.append((dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group,
                                                          async_op=True), p))
            small = [p for p in self.params if id(p) not in self.big_ids]
            if small:
                flat = torch.cat([p.grad.reshape(-1) for p in small])
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.sh_exchange is not None:
            self.sh_exchange.finish()
        if not self.active:
            return
        if work is not None:
            work.wait()
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


def _sh_multi_hip(degree, k, dirs_all, means, cam_all, object_ids, poses, v_all, scale):
    """[R,N,3] gathered factors -> summed SH gradient on the GPU (sgn_sh_bwd_multi), already split into the two
    leaves the reference keeps: (band 0 [N,1,3], bands 1.. [N,K-1,3]) — no slicing copies afterwards."""
    from . import _lib as L
    R, n = v_all.shape[0], v_all.shape[1]
    dc = torch.empty(n, 1, 3, dtype=torch.float32, device=v_all.device)
    rest = torch.empty(n, k - 1, 3, dtype=torch.float32, device=v_all.device)
    L.check(L.load().sgn_sh_bwd_multi(n, k, degree, R, L.ptr(dirs_all), L.ptr(means), L.ptr(cam_all),
                                      L.ptr(object_ids), L.ptr(poses), L.ptr(v_all.contiguous()), float(scale),
                                      L.ptr(rest) if k > 1 else None, L.ptr(dc), L.stream_ptr()), "sgn_sh_bwd_multi")
    return dc, rest


class SHGradExchange:
    """Low-rank exchange of the SH-coefficient gradient (192 of the 236 B/Gaussian).

    One view's SH gradient is ``basis(viewdir)[k] * v_rgb[c]``: ranks all-gather the 3-float colour gradient
    plus either the view directions (drop-in ops: 24 B/Gaussian/rank) or just the camera position (fused ops:
    12 B/Gaussian/rank) and rebuild the SUMMED dense gradient locally (``sgn_sh_bwd_multi``), instead of
    all-reducing 192 B/Gaussian.  On the point-to-point xGMI mesh that is 2x / 4x fewer bytes per link.
    Install once; it taps the SH backward, starts the all-gathers as soon as the colour gradient exists
    (overlapping the rest of backward) and :meth:`finish` (called by ``GradAllReducer.finish``) overwrites the
    leaf gradients of ``features_dc`` / ``features_rest`` with the cross-rank result."""

    def __init__(self, features_dc: torch.Tensor, features_rest: torch.Tensor, average: bool = True, group=None,
                 multi_fn=_sh_multi_hip, force: bool = False):
        self.dc, self.rest = features_dc, features_rest
        self.average, self.group, self.multi_fn = average, group, multi_fn
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())  # force: exercise the path at world 1
        self._stash = None
        self._last = None        # layout of the last exchange (kind, degree, K, means, object ids ...), see finish()
        self._works = []
        self._view = None

    def set_view(self, means: torch.Tensor, cam_pos: torch.Tensor) -> "SHGradExchange":
        """Drop-in ops only see view directions; a trainer that knows its camera can say so: with the (replicated)
        world means and this rank's camera position registered, the exchange gathers the 12-byte camera position
        instead of the [N,3] directions — half the bytes of the direction form.  ``means`` may be the parameter leaf
        itself (read at exchange time); call again when the camera changes."""
        self._view = (means, cam_pos)
        return self

    def leaf_ids(self):
        return {id(self.dc), id(self.rest)}

    def install(self) -> "SHGradExchange":
        from . import fused, ops
        ops._sh_bwd_tap = self._tap_dirs
        fused._sh_bwd_tap = self._tap_fused
        return self

    def remove(self) -> None:
        from . import fused, ops
        ops._sh_bwd_tap = None
        fused._sh_bwd_tap = None

    def _gather(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        try:
            w = dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)
        except Exception:  # backend without the fused form
            chunks = list(out.unbind(0))
            w = dist.all_gather(chunks, t, group=self.group, async_op=True)
        self._works.append(w)
        return out

    def _tap_dirs(self, viewdirs, v_colors, degree, k):
        """Returns True when the exchange takes the SH gradient over: the caller then skips its own dense backward
        (the local [N,K,3] gradient would be overwritten by the cross-rank sum anyway)."""
        if not self.active:
            return False
        if self._view is not None and self._view[0].shape[0] == v_colors.shape[0]:
            means, cam_pos = self._view
            cam_pos = cam_pos.detach().reshape(3).to(v_colors.device, torch.float32)
            self._stash = dict(kind="cam", degree=degree, k=k, v_all=self._gather(v_colors),
                               cam_all=self._gather(cam_pos), means=means.detach().contiguous(), object_ids=None,
                               poses=None, idft=None, keep=(v_colors, cam_pos))
            return True
        self._stash = dict(kind="dirs", degree=degree, k=k, v_all=self._gather(v_colors),
                           dirs_all=self._gather(viewdirs), keep=(viewdirs, v_colors))
        return True

    def _tap_fused(self, means, cam_pos, v_eff, degree, k, object_ids, poses, idft):
        if not self.active:
            return False
        self._stash = dict(kind="cam", degree=degree, k=k, v_all=self._gather(v_eff), cam_all=self._gather(cam_pos),
                           means=means, object_ids=object_ids, poses=poses, idft=idft, keep=(v_eff, cam_pos))
        return True

    def _participate_empty(self) -> None:
        """This rank's SH backward never ran this step (its view saw no Gaussian, so no colour gradient exists):
        the other ranks are already inside the all-gathers, so join them with a zero colour gradient — every rank
        must make the same collective calls — using what the last tap told us about the layout."""
        last = self._last
        if last is None:
            raise RuntimeError("SHGradExchange: the SH backward did not run on this rank and no earlier step is "
                               "known to take the layout from; the other ranks are waiting in all_gather")
        n = self.dc.shape[0]
        zeros = torch.zeros(n, 3, dtype=torch.float32, device=self.dc.device)
        if last["kind"] == "dirs":
            self._stash = dict(last, v_all=self._gather(zeros), dirs_all=self._gather(zeros + 1.0), keep=(zeros,))
        else:
            cam = (self._view[1] if self._view is not None else last["cam_pos"]).detach().reshape(3).to(
                zeros.device, torch.float32)
            self._stash = dict(last, v_all=self._gather(zeros), cam_all=self._gather(cam), keep=(zeros, cam))

    def start_if_silent(self) -> None:
        """Issue this step's all-gathers now if the SH backward never tapped in (see _participate_empty)."""
        if self.active and self._stash is None:
            self._participate_empty()

    def finish(self) -> None:
        if not self.active:
            return
        self.start_if_silent()
        self._last = {k: v for k, v in self._stash.items() if k not in ("v_all", "dirs_all", "cam_all", "keep")}
        if self._stash["kind"] == "cam":
            self._last["cam_pos"] = self._stash["keep"][-1]
        for w in self._works:
            w.wait()
        self._works.clear()
        s, self._stash = self._stash, None
        scale = 1.0 / self.world if self.average else 1.0
        if s["kind"] == "dirs":
            v = self.multi_fn(s["degree"], s["k"], s["dirs_all"], None, None, None, None, s["v_all"], scale)
        else:
            v = self.multi_fn(s["degree"], s["k"], None, s["means"], s["cam_all"], s["object_ids"], s["poses"],
                              s["v_all"], scale)
        if isinstance(v, tuple):                       # already split into the two leaves (HIP path)
            v0, v_rest = v
        else:                                          # dense [N,K,3] (the torch stand-in used by the gloo tests)
            v0, v_rest = v[:, 0:1, :], v[:, 1:, :]
        F = self.dc.shape[1]
        if F == 1:
            dc_grad = v0
        else:  # Fourier DC: d dc_eff / d features_dc[:, f] = idft[object, f]
            idft, oid = s["idft"], s["object_ids"]
            w = idft[oid.long()] if oid is not None else idft[:1].expand(v0.shape[0], F)
            dc_grad = w[:, :, None] * v0
        for leaf, g in ((self.dc, dc_grad), (self.rest, v_rest)):
            g = g if g.is_contiguous() else g.contiguous()
            if leaf.grad is None:
                leaf.grad = g if g.shape == leaf.shape else g.reshape(leaf.shape)
            else:
                leaf.grad.copy_(g)


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
