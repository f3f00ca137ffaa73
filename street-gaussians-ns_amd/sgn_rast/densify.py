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
    def __init__(self, group=None):
        # the process group the statistics are reduced over (:meth:`sync`); None = the default group.  Which rank plays
        # "first view of the step" below is decided INSIDE that group (a sub-group that does not hold global
        # rank 0 must still have exactly one rank that counts every Gaussian once).
        self.group = group
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
            if self._is_follower_rank():
                # The reference's first update after a refinement counts EVERY Gaussian once, visible or not (:524-527:
                # `vis_counts = torch.ones_like(...)`).  Under view-parallel training the statistics of the ranks are
                # SUMMED (`sync`), so only ONE rank's first view may do that — otherwise a Gaussian no rank saw would
                # count `world` times and N ranks would not equal one rank accumulating N views per step.  Ranks > 0
                # therefore start from zeros and take the "later view" branch (visible Gaussians only).
                self.xys_grad_norm, self.vis_counts = torch.zeros(n, **f32), torch.zeros(n, **f32)
                self.max_2Dsize = torch.zeros(n, **f32)
                first = False
            else:
                self.xys_grad_norm, self.vis_counts = torch.empty(n, **f32), torch.empty(n, **f32)
                self.max_2Dsize = torch.empty(n, **f32)
        L.check(L.load().sgn_densify_stats(n, L.ptr(g), L.ptr(r), float(max(last_size[0], last_size[1])), int(first),
                                           L.ptr(self.xys_grad_norm), L.ptr(self.vis_counts), L.ptr(self.max_2Dsize),
                                           L.stream_ptr()), "sgn_densify_stats")

    def _is_follower_rank(self) -> bool:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 and dist.get_rank(self.group) > 0

    def sync(self, group=None) -> None:
        from .dp import sync_densify_stats
        if self.xys_grad_norm is not None:
            sync_densify_stats(self.xys_grad_norm, self.vis_counts, self.max_2Dsize,
                               group=group if group is not None else self.group)


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
        self.stats.update(xys_grad, radii, self.last_size)

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
        if step <= c.warmup_length or S.xys_grad_norm is None:
            return False
        S.sync(self.group)                               # replicas decide on the statistics of ALL views
        self.record.clear()
        P = self.params
        n_before = P["means"].shape[0]
        reset_interval = c.reset_alpha_every * c.refine_every
        do_densify = step < c.stop_split_at and step % reset_interval > c.num_train_data + c.refine_every
        deleted = None
        if do_densify:
            avg = (S.xys_grad_norm / S.vis_counts) * 0.5 * max(self.last_size[0], self.last_size[1])
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
