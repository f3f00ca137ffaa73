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
    """The three running statistics of ``after_train`` for ONE model (the scene graph keeps one per sub-model).

    View-parallel training (one view per rank and step) must equal ONE process that accumulates the same views in the
    order ``world * step + rank``.  The reference's first update after a refinement counts EVERY Gaussian once, visible
    or not (``:524-527``: ``vis_counts = torch.ones_like(...)``); the later ones count the visible Gaussians only.  The
    statistics of the ranks are SUMMED (:meth:`sync`), so exactly one view of the interval may play "first" — and it has
    to be the globally first one that SAW the model: in the scene graph a sub-model is updated only on the steps whose
    frame shows it (``sgn_splatfacto_scene_graph.py:186-190`` sets ``xys = None`` on the others and ``after_train``
    returns), so rank 0 need not hold that view.  Every rank therefore starts an interval from zeros, counts visible
    Gaussians only, and remembers the key ``world * step + rank`` and the visibility mask of ITS first view;
    :meth:`sync` finds the smallest key (one MIN all-reduce of a scalar) and the rank that holds it adds the missing
    ``1 - visible`` — after which SUM / SUM / MAX give what the single process has (counts and sizes bit for bit, the
    gradient-norm sums up to the association of the fp32 additions; the replicas agree bit for bit either way).  A rank
    that never saw the model in the interval joins the collectives with zeros (a rank that skipped them would hang the
    others).
    """

    _NO_VIEW = (1 << 62)

    def __init__(self, group=None, force: bool = False):
        # the process group the statistics are reduced over (:meth:`sync`); None = the default group
        self.group = group
        # force: keep the view-parallel bookkeeping and make the collectives in a 1-rank group too (self-test of the
        # N-rank path on one GPU, like GradAllReducer(force=True); same values as the plain form)
        self.force = bool(force)
        self.xys_grad_norm: Optional[torch.Tensor] = None
        self.vis_counts: Optional[torch.Tensor] = None
        self.max_2Dsize: Optional[torch.Tensor] = None
        self._first_key: Optional[int] = None          # world > 1: key of this rank's first view of the interval ...
        self._first_visible: Optional[torch.Tensor] = None   # ... and which Gaussians it saw
        self._updates = 0                              # fallback step counter for callers that pass no step
        self.last_dim = 0.0            # larger image dimension of the most recent render that updated the statistics
        self.synced_dim: Optional[float] = None        # ... and its MAX over the ranks, from the last :meth:`sync`

    def reset(self) -> None:                 # end of refinement_after (:644-646)
        self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None
        self._first_key = self._first_visible = None

    def _world_rank(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1, 0
        return dist.get_world_size(self.group), dist.get_rank(self.group)

    def _accumulate(self, g: torch.Tensor, r: torch.Tensor, max_dim: float, first: bool) -> None:
        """The arithmetic of ``:520-541`` as one HIP pass (tests substitute the reference's torch expressions)."""
        L.require_device(g, r)
        L.check(L.load().sgn_densify_stats(g.shape[0], L.ptr(g), L.ptr(r), float(max_dim), int(first),
                                           L.ptr(self.xys_grad_norm), L.ptr(self.vis_counts), L.ptr(self.max_2Dsize),
                                           L.stream_ptr()), "sgn_densify_stats")

    @torch.no_grad()
    def update(self, xys_grad: torch.Tensor, radii: torch.Tensor, last_size, step: Optional[int] = None) -> None:
        """``xys_grad`` = ``self.xys.grad`` [N,2]; ``radii`` [N] int; ``last_size`` = (H, W) of the render; ``step`` =
        the training step (orders the views of different ranks; defaults to a per-object call counter, which is the
        same thing when every rank updates on every step)."""
        n = xys_grad.shape[0]
        g = xys_grad.detach().contiguous().float()
        r = radii.detach().to(torch.int32).contiguous()
        self._updates += 1
        first = self.xys_grad_norm is None
        if first:
            f32 = dict(dtype=torch.float32, device=g.device)
            world, rank = self._world_rank()
            if world > 1 or self.force:
                self.xys_grad_norm, self.vis_counts = torch.zeros(n, **f32), torch.zeros(n, **f32)
                self.max_2Dsize = torch.zeros(n, **f32)
                self._first_key = world * int(self._updates if step is None else step) + rank
                self._first_visible = r > 0
                first = False
            else:
                self.xys_grad_norm, self.vis_counts = torch.empty(n, **f32), torch.empty(n, **f32)
                self.max_2Dsize = torch.empty(n, **f32)
        self.last_dim = float(max(last_size[0], last_size[1]))
        self._accumulate(g, r, self.last_dim, first)

    def sync(self, group=None, n: Optional[int] = None, device=None) -> bool:
        """Reduce over the ranks (call on EVERY rank, also on one that has no statistics: pass ``n`` / ``device`` so it
        can join with zeros).  Returns False when no rank saw the model since the last reset (nothing to decide on)."""
        import torch.distributed as dist
        from .dp import sync_densify_stats
        group = group if group is not None else self.group
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not self.force):
            self.synced_dim = None
            return self.xys_grad_norm is not None
        dev = self.xys_grad_norm.device if self.xys_grad_norm is not None else torch.device(device or "cpu")
        key = self._NO_VIEW if self._first_key is None else self._first_key
        # (the image size the thresholds are scaled by rides along, negated: a sub-model's `last_size` is set by the
        # frames that show it (scene_graph.py:222-224), which differ per rank — replicas must scale by the same number)
        lowest = torch.tensor([key, -int(self.last_dim)], dtype=torch.int64, device=dev)
        dist.all_reduce(lowest, op=dist.ReduceOp.MIN, group=group)
        lowest, dim = (int(v) for v in lowest.tolist())
        self.synced_dim = float(-dim)
        if lowest == self._NO_VIEW:
            return False
        if self.xys_grad_norm is None:
            if n is None:
                raise RuntimeError("Stats.sync: this rank has no statistics for a model another rank saw; pass n and "
                                   "device so it can take part in the reduction with zeros")
            f32 = dict(dtype=torch.float32, device=dev)
            self.xys_grad_norm, self.vis_counts, self.max_2Dsize = (torch.zeros(n, **f32), torch.zeros(n, **f32),
                                                                    torch.zeros(n, **f32))
        elif key == lowest:
            # this rank rendered the interval's first view of the model: count every Gaussian once, like `ones_like`
            self.vis_counts += (~self._first_visible).to(self.vis_counts.dtype)
        sync_densify_stats(self.xys_grad_norm, self.vis_counts, self.max_2Dsize, group=group)
        return True


# ======================================================================================================================
# Split / duplicate / cull + optimiser-state surgery, replicated under data parallelism
# ======================================================================================================================
from dataclasses import dataclass
from typing import Dict

PARAM_NAMES = ("means", "log_scales", "quats", "features_dc", "features_rest", "opacity_logits")


@dataclass
class DensifyConfig:
    """The thresholds of ``SplatfactoModelConfig`` that ``refinement_after`` reads (sgn_splatfacto.py:153-226)."""
    warmup_length: int = 500
    refine_every: int = 100
    cull_alpha_thresh: float = 0.1
    cull_scale_thresh: float = 0.5
    continue_cull_post_densification: bool = True
    reset_alpha_every: int = 30
    densify_grad_thresh: float = 0.0002
    densify_size_thresh: float = 0.01
    n_split_samples: int = 2
    cull_screen_size: float = 0.15
    split_screen_size: float = 0.05
    stop_screen_size_at: int = 4000
    stop_split_at: int = 15000
    num_train_data: int = 0


def _mix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's-complement wrap-around = arithmetic modulo 2^64; the logical right
    shifts are arithmetic shifts with the sign-extension masked off)."""
    x = x * -7046029254386353131                                             # 0x9E3779B97F4A7C15
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * -4658895280553007687                # 0xBF58476D1CE4E5B9
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * -7723592293110705685               # 0x94D049BB133111EB
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def _hash_normal(base: torch.Tensor, component: int) -> torch.Tensor:
    """Standard normal deviates from a counter-based hash (Box-Muller on two 53-bit uniforms), float32."""
    import math
    u1 = ((_mix64(base + (2 * component + 1)) >> 11) & 0x1FFFFFFFFFFFFF).to(torch.float64) / float(1 << 53)
    u2 = ((_mix64(base + (2 * component + 2)) >> 11) & 0x1FFFFFFFFFFFFF).to(torch.float64) / float(1 << 53)
    return (torch.sqrt(-2.0 * torch.log(u1 + 2.0 ** -54)) * torch.cos(2.0 * math.pi * u2)).to(torch.float32)


class Densifier:
    """``SplatfactoModel.refinement_after`` / ``split_gaussians`` / ``dup_gaussians`` / ``cull_gaussians`` and the
    optimiser-state surgery ``dup_in_optim`` / ``remove_from_optim`` (``street_gaussians_ns/sgn_splatfacto.py:459-511,
    550-720``) for a parameter dictionary (the names of :mod:`sgn_rast.step`) and one optimiser per parameter — the
    reference's layout (``sgn_config.py:71-108``) — whose state uses torch.optim.Adam's keys (``exp_avg``,
    ``exp_avg_sq``; :class:`sgn_rast.optim.FusedAdam` or ``torch.optim.Adam`` alike).

    View-parallel training keeps N replicas of the Gaussians; they must take BIT-IDENTICAL decisions or they drift apart
    at the first refinement (step 600) and the gradient all-reduce stops making sense.  Everything the decisions read is
    made identical first: the statistics are reduced across ranks (``Stats.sync``: SUM of gradient norms and visibility
    counts — the average over all views rendered since the last refinement — and MAX of the screen-space size); the
    parameters are identical by construction (same initial values, same all-reduced gradients, same optimiser); and the
    one random draw (``split_gaussians``' ``torch.randn``, ``:680``) comes from a generator seeded with
    ``(seed, step)`` on every rank.  Quirks of the reference kept on purpose: ``split_gaussians`` shrinks the scales of
    the split Gaussians IN PLACE before the duplicate mask is computed (``:699`` then ``:603``), so a Gaussian just
    above the size threshold can be both split and duplicated; new optimiser rows are zeros.
    """

    def __init__(self, params: Dict[str, torch.Tensor], optimizers: Dict[str, torch.optim.Optimizer],
                 config: DensifyConfig = DensifyConfig(), seed: int = 0, group=None, stats: Optional[Stats] = None,
                 rng_device=None, split_noise: str = "stream"):
        assert set(params) == set(PARAM_NAMES) and set(optimizers) >= set(PARAM_NAMES)
        assert split_noise in ("stream", "hashed")
        # "stream" (default) draws the split offsets as the reference does: ONE torch.randn((n_splits * samps, 3)) —
        # which Gaussian gets which sample then depends on every other Gaussian's split decision.  "hashed" gives
        # every Gaussian a persistent 62-bit id (children derive theirs from the parent's) and draws its offsets from
        # a counter-based hash of (seed, step, id, sample): a Gaussian's samples no longer depend on the rest of the
        # set, so two runs whose threshold decisions differ for a handful of Gaussians stay comparable sample by sample
        # (trajectory comparisons across devices / world sizes).
        self.split_noise = split_noise
        self.ids = torch.arange(next(iter(params.values())).shape[0], dtype=torch.int64,
                                device=next(iter(params.values())).device)
        self.params, self.optimizers, self.cfg, self.seed, self.group = params, optimizers, config, seed, group
        # where the split samples are drawn: None = on the parameters' device (the reference's `torch.randn(...,
        # device=self.device)`); "cpu" = on the host and copied over — the same numbers on any device, for
        # trajectory comparisons between a GPU run and a CPU run
        self.rng_device = rng_device
        self.stats = stats if stats is not None else Stats(group=group)
        self.last_size = (1, 1)
        self.record: Dict[str, float] = {}

    # ------------------------------------------------------------------------------------------------ after_train
    def after_train(self, step: int, xys_grad: Optional[torch.Tensor], radii: torch.Tensor, last_size) -> None:
        """:513-541 — accumulate this step's statistics (stops at ``stop_split_at`` like the reference)."""
        self.last_size = (int(last_size[0]), int(last_size[1]))
        if step >= self.cfg.stop_split_at:
            return
        if xys_grad is None:                      # this rank's view saw nothing: zero gradient, nothing visible
            xys_grad = torch.zeros(radii.shape[0], 2, dtype=torch.float32, device=radii.device)
        self.stats.update(xys_grad, radii, self.last_size, step=step)

    # ------------------------------------------------------------------------------------------ optimiser surgery
    @staticmethod
    def _leaf_like(old: torch.Tensor, data: torch.Tensor) -> torch.Tensor:
        """A new autograd leaf of the kind ``old`` was (nn.Parameter, as the reference re-creates them, or plain)."""
        return torch.nn.Parameter(data) if isinstance(old, torch.nn.Parameter) else data.requires_grad_(True)

    def _rebind(self, name: str, new_param: torch.Tensor, state_fn) -> None:
        """Swap parameter ``name`` for ``new_param`` in its optimiser, transforming exp_avg / exp_avg_sq with
        ``state_fn`` (:459-475, :483-504)."""
        opt, old = self.optimizers[name], self.params[name]
        group = opt.param_groups[0]
        idx = next(i for i, p in enumerate(group["params"]) if p is old)
        st = opt.state.pop(old, {})
        if "exp_avg" in st:
            st["exp_avg"], st["exp_avg_sq"] = state_fn(st["exp_avg"]), state_fn(st["exp_avg_sq"])
        group["params"][idx] = new_param
        opt.state[new_param] = st
        self.params[name] = new_param

    # ----------------------------------------------------------------------------------------------------- pieces
    def _split(self, mask: torch.Tensor, samps: int, step: int) -> Dict[str, torch.Tensor]:
        """:674-710.  The sample offsets are drawn from a generator seeded identically on every rank."""
        P = self.params
        n_splits = int(mask.sum().item())
        dev = P["means"].device
        rdev = dev if self.rng_device is None else torch.device(self.rng_device)
        if self.split_noise == "hashed":
            pid = self.ids[mask].to(rdev)                                             # [n_splits]
            k = torch.arange(samps, dtype=torch.int64, device=rdev)[:, None]           # sample-major, like .repeat(samps, 1)
            base = _mix64(_mix64(pid[None, :] * 8 + k) ^ (self.seed * 1_000_003 + step))
            centered = torch.stack([_hash_normal(base, c) for c in range(3)], dim=-1).reshape(samps * n_splits, 3)
            centered = centered.to(dev)
            self._child_ids = (base.reshape(-1) & 0x3FFFFFFFFFFFFFFF).to(dev)
        else:
            gen = torch.Generator(device=rdev)
            gen.manual_seed((self.seed * 1_000_003 + step) & 0x7FFFFFFFFFFFFFFF)
            centered = torch.randn((samps * n_splits, 3), device=rdev, generator=gen).to(dev)
            self._child_ids = torch.zeros(samps * n_splits, dtype=torch.int64, device=dev)
        scaled = torch.exp(P["log_scales"][mask].repeat(samps, 1)) * centered
        q = P["quats"][mask] / P["quats"][mask].norm(dim=-1, keepdim=True)
        from .ops import quat_to_rotmat
        rots = quat_to_rotmat(q.repeat(samps, 1))
        rotated = torch.bmm(rots, scaled[..., None]).squeeze()
        out = {"means": rotated + P["means"][mask].repeat(samps, 1),
               "features_dc": P["features_dc"][mask].repeat(samps, 1, 1),
               "features_rest": P["features_rest"][mask].repeat(samps, 1, 1),
               "opacity_logits": P["opacity_logits"][mask].repeat(samps, 1),
               "log_scales": torch.log(torch.exp(P["log_scales"][mask]) / 1.6).repeat(samps, 1),
               "quats": P["quats"][mask].repeat(samps, 1)}
        P["log_scales"][mask] = torch.log(torch.exp(P["log_scales"][mask]) / 1.6)      # in place, like :699
        return out

    def _cull(self, step: int, extra: Optional[torch.Tensor]) -> torch.Tensor:
        """:648-672 — returns the deleted mask; parameters are replaced by their kept rows."""
        P, c = self.params, self.cfg
        culls = (torch.sigmoid(P["opacity_logits"]) < c.cull_alpha_thresh).squeeze(-1)
        if extra is not None:
            culls = culls | extra
        if step > c.refine_every * c.reset_alpha_every:
            toobig = (torch.exp(P["log_scales"]).max(dim=-1).values > c.cull_scale_thresh)
            if step < c.stop_screen_size_at:
                toobig = toobig | (self.stats.max_2Dsize > c.cull_screen_size)
            culls = culls | toobig
            self.record["refine_culls_toobigs_count"] = int(toobig.sum().item())
        keep = ~culls
        for name in PARAM_NAMES:
            self._rebind(name, self._leaf_like(P[name], P[name].detach()[keep]), lambda t: t[keep])
        self.ids = self.ids[keep]
        return culls

    # ------------------------------------------------------------------------------------------- refinement_after
    @torch.no_grad()
    def refinement_after(self, step: int) -> bool:
        """:550-646.  Returns True when the set of Gaussians changed (callers then rebuild whatever is sized by N:
        gradient reducers, cached object-id tables).  Call on EVERY rank at the same step."""
        c, S = self.cfg, self.stats
        if step <= c.warmup_length:
            return False
        P = self.params
        # replicas decide on the statistics of ALL views; a rank that never saw this model since the last refinement
        # (a scene-graph object outside its frames) joins the reduction with zeros, and when NO rank saw it every rank
        # returns here, as the reference does on `xys_grad_norm is None` (:554-555)
        if not S.sync(self.group, n=P["means"].shape[0], device=P["means"].device):
            return False
        self.record.clear()
        n_before = P["means"].shape[0]
        reset_interval = c.reset_alpha_every * c.refine_every
        do_densify = step < c.stop_split_at and step % reset_interval > c.num_train_data + c.refine_every
        deleted = None
        if do_densify:
            # (view-parallel: the size of the frames that showed this model, agreed between the ranks by Stats.sync)
            dim = S.synced_dim if getattr(S, "synced_dim", None) else max(self.last_size[0], self.last_size[1])
            avg = (S.xys_grad_norm / S.vis_counts) * 0.5 * dim
            high = avg > c.densify_grad_thresh
            splits = torch.exp(P["log_scales"]).max(dim=-1).values > c.densify_size_thresh
            if step < c.stop_screen_size_at:
                splits = splits | (S.max_2Dsize > c.split_screen_size)
            splits = splits & high
            nsamps = c.n_split_samples
            new_split = self._split(splits, nsamps, step)
            dups = (torch.exp(P["log_scales"]).max(dim=-1).values <= c.densify_size_thresh) & high
            self.record.update(high_grads_count=int(high.sum().item()), refine_splits_count=int(splits.sum().item()),
                               refine_dups_count=int(dups.sum().item()))
            n_split_new, n_dup = new_split["means"].shape[0], int(dups.sum().item())
            for name in PARAM_NAMES:
                old = P[name]
                cat = torch.cat([old.detach(), new_split[name], old.detach()[dups]], dim=0)
                pad = n_split_new + n_dup
                self._rebind(name, self._leaf_like(old, cat),
                             lambda t: torch.cat([t, t.new_zeros((pad,) + tuple(t.shape[1:]))], dim=0))
            dup_ids = _mix64(self.ids[dups] * 8 + 7 + step) & 0x3FFFFFFFFFFFFFFF
            self.ids = torch.cat([self.ids, self._child_ids, dup_ids])
            S.max_2Dsize = torch.cat([S.max_2Dsize, S.max_2Dsize.new_zeros(n_split_new + n_dup)], dim=0)
            splits_mask = torch.cat([splits, splits.new_zeros(n_split_new + n_dup)])
            deleted = self._cull(step, splits_mask)      # the split originals go, plus the usual culls
        elif step >= c.stop_split_at and c.continue_cull_post_densification:
            deleted = self._cull(step, None)
        if step < c.stop_split_at and step % reset_interval == c.refine_every:          # opacity reset (:625-641)
            reset_value = c.cull_alpha_thresh * 2.0       # fp32 logit read back as a Python float, like :631
            P["opacity_logits"].data = torch.clamp(
                P["opacity_logits"].data,
                max=torch.logit(torch.tensor(reset_value, device=P["opacity_logits"].device)).item())
            st = self.optimizers["opacity_logits"].state.get(P["opacity_logits"], {})
            if "exp_avg" in st:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(st["exp_avg"]), torch.zeros_like(st["exp_avg_sq"])
        S.reset()
        return deleted is not None or P["means"].shape[0] != n_before


class SceneGraphDensifier:
    """The scene graph's densification: ONE :class:`Densifier` (and one :class:`Stats`) per sub-model over the SHARED
    per-name optimisers, as ``SplatfactoSceneGraphModel.get_training_callbacks`` registers every sub-model's own
    ``after_train`` / ``refinement_after`` (``sgn_splatfacto_scene_graph.py:127-135``) against optimisers whose single
    param group lists the sub-models' parameters in model order (``get_gaussian_param_groups``, ``:110-119``;
    ``dup_in_optim`` / ``remove_from_optim`` address them by ``_model_idx_in_scene_graph``, ``sgn_splatfacto.py:459-505``
    — here by identity).

    ``after_train`` takes the step's VISIBLE sub-models only: the reference sets ``xys = None`` on the others
    (``:186-190``) and their ``after_train`` returns.  Under view-parallel training the ranks see different frames, hence
    different visible sets; every sub-model's statistics are reduced on their own (:meth:`Stats.sync`: the interval's
    first view of THAT model counts every Gaussian once, a rank that never saw it joins with zeros), in model order on
    every rank, so replicas stay bit-identical.  ``models`` is the live list of parameter dictionaries (entries are
    replaced in place when a sub-model's Gaussians change)."""

    def __init__(self, models, optimizers: Dict[str, torch.optim.Optimizer], config=DensifyConfig(), seed: int = 0,
                 group=None, stats_factory=Stats, **densifier_kw):
        cfgs = list(config) if isinstance(config, (list, tuple)) else [config] * len(models)
        assert len(cfgs) == len(models)
        self.models = models
        self.parts = [Densifier(m, optimizers, cfgs[i], seed=seed * 1009 + i, group=group,
                                stats=stats_factory(group=group), **densifier_kw) for i, m in enumerate(models)]

    def after_train(self, step: int, visible, xys_grads, radii_parts, last_size) -> None:
        """``visible``: indices of the sub-models in this step's render, in render order; ``xys_grads[j]`` /
        ``radii_parts[j]``: the retained ``xys.grad`` (None: the view saw nothing) and the radii of ``visible[j]``."""
        for j, i in enumerate(visible):
            self.parts[i].after_train(step, xys_grads[j], radii_parts[j], last_size)

    def refinement_after(self, step: int):
        """Every sub-model in model order (the same collectives in the same order on every rank).  Returns the list of
        "changed" flags; ``models[i]`` is that sub-model's current parameter dictionary afterwards."""
        changed = []
        for i, d in enumerate(self.parts):
            changed.append(d.refinement_after(step))
            self.models[i] = d.params
        return changed
