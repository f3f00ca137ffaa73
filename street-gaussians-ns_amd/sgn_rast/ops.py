"""Host side of the hot path: the gsplat 0.1.x operator surface over libsgnrast.so.

Mirrors (names, argument order/meaning, return tuples, assertions) the third-party
``gsplat`` modules the reference imports at ``sgn_splatfacto.py:11-14`` and
``sgn_splatfacto_scene_graph.py:8``:

* ``gsplat/project_gaussians.py``  -> :func:`project_gaussians`  (``_ProjectGaussians``)
* ``gsplat/rasterize.py``          -> :func:`rasterize_gaussians` (``_RasterizeGaussians``)
* ``gsplat/sh.py``                 -> :func:`spherical_harmonics`, :func:`num_sh_bases`
* ``gsplat/utils.py``              -> binning helpers
* ``gsplat/_torch_impl.py``        -> :func:`quat_to_rotmat` (the only symbol the reference uses)

PyTorch is plumbing here (device memory, current stream, autograd graph); all
arithmetic happens in the hand-written HIP kernels behind the C ABI.
"""
from __future__ import annotations

import contextlib
import contextvars
import ctypes as C
import os
import threading
import weakref
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import _lib as L
from . import config
from . import proofs

# ------------------------------------------------------------- upstream-variant semantics (round 6)
# gsplat is absent where this library is built, so three behaviours of gsplat 0.1.x are DECIDED from recollection of
# upstream rather than read from its source (SURVEY.md Appendix A [verify] / [decide]).  Each is a
# CALL-TIME switch with the decided behaviour as default — if tests/golden/make_upstream_golden.py, run against the real
# gsplat, disagrees on one of them, the fix is a flag flip here, not a rewrite of a kernel:
#   tile_bbox_add_after_cast   False: tile-box max side (int)(c + r + 1), gsplat helpers.cuh (CUDA); True: (int)(c + r) + 1,
#                              gsplat/_torch_impl.py.  Read by project_gaussians AND by the binning inside
#                              rasterize_gaussians (they must agree: set it around both).
#   ewa_vjp_clamped            False: the projection backward uses the Jacobian of the UN-clamped view-space point
#                              (upstream CUDA project_cov3d_ewa_vjp); True: it differentiates through the forward's
#                              +-1.3 tan(fov/2) clamp (autograd through _torch_impl).  Read by project_gaussians.
#   alpha_clamp_bwd            0.99: upstream's backward clamp (0.999 in the forward); 0.999: the self-consistent variant.
#                              Read by rasterize_gaussians (at the call: the backward runs with the call's value).
# `with ops.upstream_variant(tile_bbox_add_after_cast=True): ...` changes them for the current thread / context only
# (contextvars); `ops.semantics()` is what the next call will use; the C ABI takes them as arguments (SGN_SEM_* bits
# of include/sgn_rast.h, alpha_clamp_bwd of sgn_raster_bwd): the library holds no state.
UPSTREAM_ALPHA_CLAMP_BWD = 0.99
SEM_BBOX_ADD_AFTER_CAST, SEM_EWA_VJP_CLAMPED = 1, 2       # SGN_SEM_* of include/sgn_rast.h


class Semantics:
    __slots__ = ("tile_bbox_add_after_cast", "ewa_vjp_clamped", "alpha_clamp_bwd")

    def __init__(self, tile_bbox_add_after_cast=False, ewa_vjp_clamped=False, alpha_clamp_bwd=UPSTREAM_ALPHA_CLAMP_BWD):
        self.tile_bbox_add_after_cast = bool(tile_bbox_add_after_cast)
        self.ewa_vjp_clamped = bool(ewa_vjp_clamped)
        self.alpha_clamp_bwd = float(alpha_clamp_bwd)

    def flags(self) -> int:
        return (SEM_BBOX_ADD_AFTER_CAST if self.tile_bbox_add_after_cast else 0) | (
            SEM_EWA_VJP_CLAMPED if self.ewa_vjp_clamped else 0)

    def copy(self) -> "Semantics":
        return Semantics(self.tile_bbox_add_after_cast, self.ewa_vjp_clamped, self.alpha_clamp_bwd)

    def __repr__(self):
        return (f"Semantics(tile_bbox_add_after_cast={self.tile_bbox_add_after_cast}, ewa_vjp_clamped="
                f"{self.ewa_vjp_clamped}, alpha_clamp_bwd={self.alpha_clamp_bwd})")


_process_semantics = Semantics()
_ctx_semantics: contextvars.ContextVar = contextvars.ContextVar("sgn_semantics", default=None)


def semantics() -> Semantics:
    """The upstream-variant semantics the next operator call of this context carries."""
    cur = _ctx_semantics.get()
    return cur if cur is not None else _process_semantics


@contextlib.contextmanager
def upstream_variant(**kw):
    """`with upstream_variant(tile_bbox_add_after_cast=True, ewa_vjp_clamped=True, alpha_clamp_bwd=0.999): ...` — a
    private copy of the semantics for this thread / context (see the note above)."""
    s = semantics().copy()
    for k, v in kw.items():
        if k not in Semantics.__slots__:
            raise AttributeError(k)
        setattr(s, k, float(v) if k == "alpha_clamp_bwd" else bool(v))
    token = _ctx_semantics.set(s)
    try:
        yield s
    finally:
        _ctx_semantics.reset(token)


def set_alpha_clamp_bwd(value: float) -> None:
    """0.99 (default) reproduces upstream; 0.999 is the self-consistent variant.  (Process default; inside
    `upstream_variant` the context's own copy.)"""
    semantics().alpha_clamp_bwd = float(value)


def _i32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype is torch.int32 and t.is_contiguous():
        return t.detach()
    return t.detach().to(torch.int32).contiguous()


def _contig(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()      # (`.contiguous()` on a contiguous tensor still costs a dispatch)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype is torch.float32 and t.is_contiguous():          # the usual case: one call instead of three
        return t.detach()
    return t.detach().to(torch.float32).contiguous()



# ------------------------------------------------------------------ host state
# Everything the host side remembers between calls — the binning cache and its pending prefetch, the speculative
# capacities, deferred argument checks, pinned read-back slots, the depth-channel and early-rank speculation — lives in
# ONE object per (device, stream) (these were module globals with one entry each).  Two models,
# cameras or threads that work on different streams never see each other's state; callers that interleave on ONE
# stream are served by small LRUs (four binnings, four depth channels, four launch orders) instead of one slot each.
# A single (device, stream) must not be driven from two threads at once (as with any stream-ordered library).
import collections


class _State:
    BIN_ENTRIES = 4

    def __init__(self):
        # the most recent binning (window matching); "info": what the autograd graph said about its tensors
        self.bin_cache = {"key": None, "keep": None, "val": None, "info": None}
        self.bin_older = collections.OrderedDict()                     # key -> (keep, val) of the three before it
        self.bin_pending = {"key": None, "state": None, "keep": None}
        self.order_cache = collections.OrderedDict()                   # (id(tile_bins), thresh) -> (tile_bins, order)
        self.last_count: dict = {}     # key -> slowly decaying maximum of the counts seen (views of one scene differ)
        self.pending_checks: list = []
        self.side = None               # [pinned int32[4,8] (count, up to 6 deferred-check flags, walk statistic), next slot]
        self.walk_stat = None          # device int32[1]: the last backward's walked / listed permille, not yet read back
        self.walked_permille = None    # ... and the last value the host has seen (quadrant-mask policy)
        self.stat_skipped = 0          # binnings since the statistic was last read back (every eighth one carries it)
        self.eager_side = None         # [pinned int32[8] ring for the eager flag read-back, next slot]
        self.quat_flag = None          # [device int32[1] zeroed once, stamp of the last call] (sgn_project_fwd_all)
        self.quat_ring = None          # [pinned int32[32]: eight (failed, landed, complete, -) slots of that call, next slot]
        self.depth_state = {"want": False, "unused": 0, "cache": None}
        self.depth_caches = collections.OrderedDict()                  # binning key -> first pass's channel + state
        self.early = {"entry": None, "misses": 0, "pause": 0}
        self.order_scratch: dict = {}  # n_tiles -> persistent zero-filled scratch of the multi-workgroup tile order
        self.no_speculation_once = False   # the one-call forward missed its capacity: the next binning runs its plain form

    # -- binning cache: most recent entry + a short LRU behind it
    def find_binning(self, key):
        if self.bin_cache["key"] == key:
            return self.bin_cache["val"]
        hit = self.bin_older.pop(key, None)
        if hit is None:
            return None
        self._retire_current()
        self.bin_cache["key"], (self.bin_cache["keep"], self.bin_cache["val"]) = key, hit
        self.bin_cache["info"] = None
        return hit[1]

    def has_binning(self, key) -> bool:
        return self.bin_cache["key"] == key or key in self.bin_older

    def store_binning(self, key, keep, val, info=None) -> None:
        self._retire_current()
        self.bin_cache["key"], self.bin_cache["keep"], self.bin_cache["val"] = key, keep, val
        self.bin_cache["info"] = info        # holds autograd nodes of the step: only ever on the most recent entry

    def _retire_current(self) -> None:
        k = self.bin_cache["key"]
        if k is not None:
            self.bin_older[k] = (self.bin_cache["keep"], self.bin_cache["val"])
            while len(self.bin_older) > self.BIN_ENTRIES - 1:
                old, _ = self.bin_older.popitem(last=False)
                self.depth_caches.pop(old, None)

    def clear_binning(self) -> None:
        self.bin_cache["key"] = self.bin_cache["keep"] = self.bin_cache["val"] = self.bin_cache["info"] = None
        self.bin_older.clear()
        self.order_cache.clear()
        self.depth_caches.clear()
        self.depth_state["cache"] = None


_have_gpu = None
_states: "collections.OrderedDict" = collections.OrderedDict()
_MAX_STATES = 16        # programs that keep creating streams must not pin a cache (device memory) per stream for ever


def _S() -> _State:
    """The state object of the current device's current stream."""
    global _have_gpu
    if _have_gpu is None:
        _have_gpu = torch.cuda.is_available()
    k = (L.current_device(), L.stream_handle()) if _have_gpu else (-1, 0)
    st = _states.get(k)
    if st is not None:
        if len(_states) > 1:
            _states.move_to_end(k)                  # true LRU: the stream in use is the last to be evicted
        return st
    if st is None:
        st = _states[k] = _State()
        while len(_states) > _MAX_STATES:          # least recently USED goes (its caches are only speed)
            _, old = _states.popitem(last=False)
            old.clear_binning()
            failed = False                          # deferred argument checks of that stream are settled, not dropped
            while old.pending_checks:
                failed |= bool(int(old.pending_checks.pop().item()))
            assert not failed, "quats must be normalized"
    return st


_STATE_ALIASES = {"_bin_cache": "bin_cache", "_bin_pending": "bin_pending", "_last_count": "last_count",
                  "_pending_checks": "pending_checks", "_depth_state": "depth_state", "_early": "early",
                  "_order_cache": "order_cache"}


def __getattr__(name):          # ops._bin_cache etc.: the CURRENT (device, stream)'s objects (tests, diagnostics)
    if name in _STATE_ALIASES:
        return getattr(_S(), _STATE_ALIASES[name])
    raise AttributeError(name)


# --------------------------------------------------------------------- sh
def num_sh_bases(degree: int) -> int:
    """gsplat/sh.py num_sh_bases (reference: sgn_splatfacto.py:268)."""
    if degree == 0:
        return 1
    if degree == 1:
        return 4
    if degree == 2:
        return 9
    if degree == 3:
        return 16
    return 25


def deg_from_sh(num_bases: int) -> int:
    for d, n in enumerate((1, 4, 9, 16, 25)):
        if n == num_bases:
            return d
    assert False, "Invalid number of SH bases"


# Optional hook for data-parallel training (sgn_rast.dp.SHGradExchange).  At forward time the exchange says whether it
# may take this node's gradient over (`claims_coeffs`: the coefficients are literally torch.cat of its two registered
# leaves); at backward time `tap_dirs` records the low-rank factors (view directions, colour gradient) so they can be
# exchanged instead of the dense [N,K,3] tensor.  None in normal operation.
_sh_exchange = None


class _SphericalHarmonics(Function):
    @staticmethod
    def forward(ctx, degrees_to_use: int, viewdirs: torch.Tensor, coeffs: torch.Tensor, claimed: bool = False):
        L.require_device(viewdirs, coeffs)
        num_points, k = coeffs.shape[0], coeffs.shape[-2]
        ctx.degrees_to_use, ctx.k, ctx.claimed = degrees_to_use, k, bool(claimed)
        ctx.set_materialize_grads(False)
        deg_from_sh(k)
        viewdirs = _f32c(viewdirs)
        coeffs_c = _f32c(coeffs)
        colors = torch.empty(num_points, 3, dtype=torch.float32, device=coeffs.device)
        L.check(L.load().sgn_sh_fwd(num_points, k, degrees_to_use, L.ptr(viewdirs), L.ptr(coeffs_c),
                                    L.ptr(colors), L.stream_ptr()), "sgn_sh_fwd")
        ctx.save_for_backward(viewdirs)
        return colors

    @staticmethod
    def backward(ctx, v_colors: torch.Tensor):
        if v_colors is None and _sh_exchange is None:   # the colours took no part in the loss (accumulation-only pass)
            return None, None, None, None
        (viewdirs,) = ctx.saved_tensors
        if v_colors is None:
            v_colors = torch.zeros(viewdirs.shape[0], 3, dtype=torch.float32, device=viewdirs.device)
        n = v_colors.shape[0]
        v_colors = _f32c(v_colors)
        if _sh_exchange is not None and _sh_exchange.tap_dirs(viewdirs, v_colors, ctx.degrees_to_use, ctx.k,
                                                              ctx.claimed):
            return None, None, None, None    # the data-parallel exchange rebuilds the (summed) gradient itself
        v_coeffs = torch.empty(n, ctx.k, 3, dtype=torch.float32, device=v_colors.device)
        L.check(L.load().sgn_sh_bwd(n, ctx.k, ctx.degrees_to_use, L.ptr(viewdirs), L.ptr(v_colors),
                                    L.ptr(v_coeffs), L.stream_ptr()), "sgn_sh_bwd")
        return None, None, v_coeffs, None


# The reference builds its coefficient tensor as `torch.cat((features_dc, features_rest), dim=1)` of two leaf parameters
# (sgn_splatfacto.py:858) — 192 MB at 1 M Gaussians.  A dense SH gradient [N,K,3] is then cut back into the two leaves by
# autograd: CatBackward hands out strided views of it and AccumulateGrad copies each view into a contiguous `.grad`
# (a 180 MB read + write per step, ~50 us, plus the 192 MB allocation).  In the scene graph the two halves are themselves
# row-wise concatenations over the sub-models, the objects' DC term a Fourier sum (scene_graph.py:239-247, 355-360): two
# more levels of strided views, ~20 copy kernels and ~40 graph nodes.  When the graph behind `coeffs` PROVES that shape
# (proofs.sh_source; no hook on the concatenation, nobody retaining its gradient) the SH node takes the LEAVES as its
# autograd inputs and its backward writes their gradients directly: one kernel for band 0 / the rest, one more for all
# Fourier fan-outs; same values, no dense tensor, no copies.  Anything else takes the dense path.
# Option `graph_proofs` (sgn_rast/config.py) switches it off.
sh_split_backward = bool(config.value("graph_proofs"))
sh_split_stats = {"split": 0, "dense": 0}
_sh_memo = None        # (weakref to the last proven coefficient tensor, its ShSource)


def _proofs_on(flag: bool) -> bool:
    return flag and proofs.enabled()


class _SphericalHarmonicsSplit(Function):
    """spherical_harmonics over a proven `cat((DC, REST), 1)`: reads the concatenation, differentiates into the leaves
    (`src`: proofs.ShSource; the leaves follow as autograd inputs, DC parts first, then the REST leaves)."""
    @staticmethod
    def forward(ctx, degrees_to_use: int, viewdirs: torch.Tensor, coeffs: torch.Tensor, src, *leaves):
        ctx.bypassed = _bypassed_refs()          # (the caller's torch.cat((dc, rest), 1) tensor)
        L.require_device(viewdirs, coeffs)
        num_points, k = coeffs.shape[0], coeffs.shape[-2]
        ctx.degrees_to_use, ctx.k, ctx.src = degrees_to_use, k, src
        ctx.set_materialize_grads(False)
        deg_from_sh(k)
        viewdirs = _f32c(viewdirs)
        colors = torch.empty(num_points, 3, dtype=torch.float32, device=coeffs.device)
        L.check(L.load().sgn_sh_fwd(num_points, k, degrees_to_use, L.ptr(viewdirs), L.ptr(_f32c(coeffs)),
                                    L.ptr(colors), L.stream_ptr()), "sgn_sh_fwd")
        ctx.save_for_backward(viewdirs)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        src = ctx.src
        n_leaves = len(src.dc) + len(src.rest)
        if v_colors is None:                     # the colours took no part in the loss (an accumulation-only pass)
            return (None,) * (4 + n_leaves)
        (viewdirs,) = ctx.saved_tensors
        n = v_colors.shape[0]
        v_colors = _f32c(v_colors)
        f32 = dict(dtype=torch.float32, device=v_colors.device)
        lib = L.load()
        by = getattr(ctx, "bypassed", ())
        asked = _asked_for(by[0]) if by else None
        if asked is not None:
            # a hook / retain_grad placed on the concatenated coefficients AFTER the call: the dense gradient upstream
            # returns, handed to autograd at that tensor (its hooks fire; cat / Fourier-sum backward reach the leaves)
            v_coeffs = torch.empty(n, ctx.k, 3, **f32)
            L.check(lib.sgn_sh_bwd(n, ctx.k, ctx.degrees_to_use, L.ptr(viewdirs), L.ptr(v_colors), L.ptr(v_coeffs),
                                   L.stream_ptr()), "sgn_sh_bwd")
            hooks_after_call_stats["sh"] += 1
            torch.autograd.backward([asked], [v_coeffs.reshape(asked.shape)], retain_graph=True)
            return (None,) * (4 + n_leaves)
        one_dc = src.dc[0].leaf if (len(src.dc) == 1 and src.dc[0].weights is None) else None
        v_dc = _leaf_grad(_arena_leaves(one_dc), 0, (n, 1, 3), f32)        # (a DP bucket member: produced in its slice)
        v_rest = torch.empty(n, ctx.k - 1, 3, **f32)
        L.check(lib.sgn_sh_bwd_multi(n, ctx.k, ctx.degrees_to_use, 1, L.ptr(viewdirs), None, None, None, None,
                                     L.ptr(v_colors), 1.0, L.ptr(v_rest), L.ptr(v_dc), L.stream_ptr()),
                "sgn_sh_bwd_multi")
        # DC: plain parts are row windows of v_dc; the Fourier parts fan out in ONE launch
        grads, row, four = [], 0, []
        for part in src.dc:
            rows = part.leaf.shape[0]
            if part.weights is None:
                grads.append(v_dc if rows == n else v_dc[row:row + rows])
            else:
                out = torch.empty(part.leaf.shape, **f32)
                four.append((row, rows, part.leaf.shape[1], part.weights, out))
                grads.append(out)
            row += rows
        if four:
            m = len(four)
            i32s, ptrs = C.c_int32 * m, C.c_void_p * m
            L.check(lib.sgn_fourier_dc_bwd(m, i32s(*[f[0] for f in four]), i32s(*[f[1] for f in four]),
                                           i32s(*[f[2] for f in four]), ptrs(*[f[3].data_ptr() for f in four]),
                                           ptrs(*[f[4].data_ptr() for f in four]), L.ptr(v_dc), L.stream_ptr()),
                    "sgn_fourier_dc_bwd")
        if len(src.rest) == 1:
            grads.append(v_rest)
        else:
            grads.extend(v_rest.split([r.shape[0] for r in src.rest]))
        return (None, None, None, None) + tuple(grads)


def spherical_harmonics(degrees_to_use: int, viewdirs: torch.Tensor, coeffs: torch.Tensor,
                        method: str = "fast") -> torch.Tensor:
    """gsplat/sh.py spherical_harmonics (sgn_splatfacto.py:939; scene_graph.py:285).

    ``coeffs`` [N, K, 3] with K in {1,4,9,16,25}; gradient flows to ``coeffs`` only.
    ``method`` is accepted for signature parity ("poly" and "fast" are the same
    polynomial; the kernel evaluates the "fast" recurrences)."""
    assert coeffs.shape[-2] >= num_sh_bases(degrees_to_use)
    assert method in ("poly", "fast"), "Invalid method."
    claimed = _sh_exchange is not None and _sh_exchange.claims_coeffs(coeffs, degrees_to_use)
    if _sh_exchange is None and _proofs_on(sh_split_backward) and coeffs.is_contiguous() and (
            coeffs.is_cuda or not proofs.need_device):
        # the scene graph's sub-model passes evaluate the SH twice from the SAME concatenation (scene_graph.py:285, then
        # render_gaussian_attrs :939): the second call reuses the first one's reading of the graph
        global _sh_memo
        if _sh_memo is not None and _sh_memo[0]() is coeffs and coeffs._version == 0 and proofs.unhooked(coeffs):
            src = _sh_memo[1]
        else:
            src = proofs.sh_source(coeffs)
            _sh_memo = (weakref.ref(coeffs), src) if src is not None else None
        if src is not None:
            sh_split_stats["split"] += 1
            _call_state.bypassed = (coeffs,)
            return _SphericalHarmonicsSplit.apply(degrees_to_use, _contig(viewdirs), coeffs.detach(), src,
                                                  *[p.leaf for p in src.dc], *src.rest)
    sh_split_stats["dense"] += 1
    return _SphericalHarmonics.apply(degrees_to_use, _contig(viewdirs), _contig(coeffs), claimed)


# ---------------------------------------------------------------- project
class _ProjectGaussians(Function):
    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width,
                block_width, clip_thresh=0.01):
        outs, saved = _project_forward(ctx, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height,
                                       img_width, block_width, clip_thresh)
        ctx.save_for_backward(*saved)
        ctx.arena_leaves = _arena_leaves(means3d, scales, quats)
        return outs

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_compensation, v_num_tiles_hit, v_cov3d):
        means3d, scales, quats, viewmat, cov3d, radii, conics, compensation = ctx.saved_tensors
        n, dev = means3d.shape[0], means3d.device
        f32 = dict(dtype=torch.float32, device=dev)
        v_xys = _f32c(v_xys) if v_xys is not None else torch.zeros(n, 2, **f32)
        v_depths = _f32c(v_depths) if v_depths is not None else None      # NULL = zeros inside the kernel
        v_conics = _f32c(v_conics) if v_conics is not None else torch.zeros(n, 3, **f32)
        v_comp = _f32c(v_compensation) if v_compensation is not None else None
        al = getattr(ctx, "arena_leaves", None)
        v_mean = _leaf_grad(al, 0, (n, 3), f32)
        v_scale = _leaf_grad(al, 1, (n, 3), f32)
        v_quat = _leaf_grad(al, 2, (n, 4), f32)
        want_viewmat = bool(ctx.needs_input_grad[4])
        v_cov2d = torch.empty(n, 3, **f32) if want_viewmat else None
        L.check(L.load().sgn_project_bwd(
            n, L.ptr(means3d), L.ptr(scales), ctx.glob_scale, L.ptr(quats), L.ptr(viewmat), ctx.fx, ctx.fy,
            L.ptr(cov3d), L.ptr(radii), L.ptr(conics), L.ptr(compensation), L.ptr(v_xys), L.ptr(v_depths),
            L.ptr(v_conics), L.ptr(v_comp), L.ptr(v_cov2d), None, L.ptr(v_mean), L.ptr(v_scale), L.ptr(v_quat),
            ctx.sem, ctx.img_hw[0], ctx.img_hw[1], L.stream_ptr()), "sgn_project_bwd")
        # (means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, block, clip)
        # viewmat gradient: never requested by the reference (camera optimiser "off", sgn_config.py:44: `None`, as
        # upstream returns when viewmat.requires_grad is False); when it IS requested (round 5) it is assembled on the
        # host from the kernel's per-Gaussian outputs — a cold path, a dozen torch ops
        v_viewmat = _viewmat_grad(means3d, viewmat, ctx.fx, ctx.fy, cov3d, radii, v_mean, v_cov2d,
                                  ctx.viewmat_shape) if want_viewmat else None
        return v_mean, v_scale, None, v_quat, v_viewmat, None, None, None, None, None, None, None, None


def _viewmat_grad(means3d, viewmat12, fx, fy, cov3d, radii, v_mean, v_cov2d, shape):
    """dL/d(viewmat) of `project_gaussians` from what the backward kernel already returns (upstream computes it when
    `viewmat.requires_grad`; the reference never asks).  With V = [R | t], p_v = R p + t, T = J(p_v) R and
    cov2d = T Sigma T^T:
        v_R = sum_i v_pv_i (x) p_i + J_i^T v_T_i,    v_t = sum_i v_pv_i,
    where v_pv is the gradient of the view-space point — the kernel's `v_mean = R^T v_pv`, solved back — and
    v_T = 2 G T Sigma with G the symmetric cov2d gradient (`v_cov2d`, off-diagonal halved).  Same conventions as the
    analytic vjp of the means (un-clamped J); culled Gaussians contribute nothing."""
    f64 = torch.float64
    V = viewmat12.reshape(3, 4).to(f64)
    R, t = V[:, :3], V[:, 3]
    live = (radii > 0)
    p = means3d.to(f64)
    vm = torch.where(live[:, None], v_mean.to(f64), torch.zeros_like(p))
    v_pv = torch.linalg.solve(R.T, vm.T).T                                   # v_mean = R^T v_pv
    pv = p @ R.T + t
    rz = 1.0 / torch.where(live, pv[:, 2], torch.ones_like(pv[:, 2]))
    J = torch.zeros(p.shape[0], 2, 3, dtype=f64, device=p.device)
    J[:, 0, 0], J[:, 0, 2] = fx * rz, -fx * pv[:, 0] * rz * rz
    J[:, 1, 1], J[:, 1, 2] = fy * rz, -fy * pv[:, 1] * rz * rz
    T = J @ R
    c = cov3d.to(f64)
    S = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], dim=-1).reshape(-1, 3, 3)
    g = torch.where(live[:, None], v_cov2d.to(f64), torch.zeros(1, 3, dtype=f64, device=p.device))
    G = torch.stack([g[:, 0], 0.5 * g[:, 1], 0.5 * g[:, 1], g[:, 2]], dim=-1).reshape(-1, 2, 2)
    vT = 2.0 * G @ T @ S
    v_R = v_pv.T @ p + (J.transpose(1, 2) @ vT).sum(dim=0)
    out = torch.zeros(shape, dtype=torch.float32, device=p.device)
    out[:3, :3] = v_R.to(torch.float32)
    out[:3, 3] = v_pv.sum(dim=0).to(torch.float32)
    return out


# The reference hands `project_gaussians` its activated parameters — `torch.exp(scales)` (sgn_splatfacto.py:857) and
# `quats / quats.norm(dim=-1, keepdim=True)` (:864) — and autograd then carries the projection's gradient back through
# those expressions: ~10 small kernels and four graph nodes per step (division and norm backward, their reductions, the
# accumulation of the two paths, the exp backward), in the scene graph additionally the row-wise split of the scale
# gradient over the sub-models.  When the graph behind the two arguments PROVES those shapes (proofs.exp_leaves: exp of a
# leaf or of a row-wise concatenation of leaves; proofs.normalised_source: X divided by its own 2-norm over the last dim
# with keepdim; nobody hooked or retained the activated tensors), the projection node takes the log-scale LEAVES and X as
# its autograd inputs and its backward returns their gradients from one kernel (sgn_project_bwd_act: exp and
# normalisation differentiated inside, from the caller's activated scales and from X — nothing is recomputed from the
# leaves' current values).  The forward still runs on the caller's activated values, bit for bit; gradients agree with
# the chain through torch to fp32 rounding.  Option `graph_proofs` (sgn_rast/config.py) switches it off.
activation_proofs = bool(config.value("graph_proofs"))
activation_proof_stats = {"project": 0, "opacity": 0, "colors": 0, "window": 0}


_QUAT_STAMP_BASE = 1 << 28


def _project_forward(ctx, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width,
                     block_width, clip_thresh):
    """Shared by the two projection nodes: launches sgn_project_fwd, returns (outputs, tensors to save)."""
    num_points = means3d.shape[-2]
    if num_points < 1 or means3d.shape[-1] != 3:
        raise ValueError(f"Invalid shape for means3d: {means3d.shape}")
    dev = L.require_device(means3d, scales, quats, viewmat)
    means3d_c, scales_c, quats_c = _f32c(means3d), _f32c(scales), _f32c(quats)
    viewmat_c = _f32c(viewmat).reshape(-1)[:12].contiguous()
    n = num_points
    f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
    cov3d = torch.empty(n, 6, **f32)
    xys = torch.empty(n, 2, **f32)
    depths = torch.empty(n, **f32)
    radii = torch.empty(n, **i32)
    conics = torch.empty(n, 3, **f32)
    compensation = torch.empty(n, **f32)
    num_tiles_hit = torch.empty(n, **i32)
    plan = getattr(_call_state, "project_plan", None)
    _call_state.project_plan = None
    sem = semantics().flags()
    if plan is not None:
        # ONE library call (sgn_project_fwd_all, round 5): the quats check's device pass, the projection and the early
        # depth ranking queued together, the wait for the check's flag last
        lib = L.load()
        S = _S()
        flag, stamp, slot = None, 0, None
        if plan["check"]:
            # the device flag of this (device, stream): zeroed ONCE, stamped by a failing row with a per-call counter,
            # so no call clears it (a launch saved per step); `eager_side`: pinned ring the flag is copied to
            # (stamps start far above the 0 / 1 the call-by-call check copies into the same pinned ring)
            if S.quat_flag is None or S.quat_flag[1] >= 2**31 - 2:
                S.quat_flag = [torch.zeros(1, **i32), _QUAT_STAMP_BASE]
                S.quat_ring = None
            S.quat_flag[1] += 1
            flag, stamp = S.quat_flag
            if S.quat_ring is None:
                S.quat_ring = [torch.zeros(32, dtype=torch.int32).pin_memory(), 0]
            ring = S.quat_ring             # eight slots of four words: [failed stamp, landed stamp, complete stamp, -]
            slot = ring[0][4 * (ring[1] % 8):4 * (ring[1] % 8) + 4]     # (include/sgn_rast.h sgn_project_fwd_all)
            ring[1] += 1
        gid = ws = None
        if plan["rank"]:
            gid = torch.empty(n, **i32)
            ws = L.workspace(lib.sgn_depth_rank_workspace_bytes(n), dev)
        L.check(lib.sgn_project_fwd_all(
            n, L.ptr(means3d_c), L.ptr(scales_c), float(glob_scale), L.ptr(quats_c), L.ptr(viewmat_c),
            float(fx), float(fy), float(cx), float(cy), int(img_height), int(img_width), int(block_width),
            float(clip_thresh), L.ptr(cov3d), L.ptr(xys), L.ptr(depths), L.ptr(radii), L.ptr(conics),
            L.ptr(compensation), L.ptr(num_tiles_hit), 2 if plan["check"] else 0, 1e-6, L.ptr(flag), int(stamp),
            slot.data_ptr() if slot is not None else None, L.ptr(gid), L.ptr(ws), ws.numel() if ws is not None else 0,
            L.sort_rank_mode(), None, sem, L.stream_ptr()), "sgn_project_fwd_all")
        # (check_quats = 2: queued, not waited for — `project_gaussians` waits after autograd has wrapped the outputs:
        # whatever the host does before the wait is off the step's critical path, the device is busy behind the flag)
        plan["done"], plan["wait"] = True, ((slot, int(stamp)) if plan["check"] else None)
        if gid is not None:
            d, r = depths.detach(), radii.detach()
            S.early["entry"] = dict(key=(d.data_ptr(), d._version, r.data_ptr(), r._version, n, L.stream_handle()),
                                    keep=(d, r), gid=gid, done=None)
            early_rank_stats["started"] += 1
    else:
        L.check(L.load().sgn_project_fwd(
            n, L.ptr(means3d_c), L.ptr(scales_c), float(glob_scale), L.ptr(quats_c), L.ptr(viewmat_c),
            float(fx), float(fy), float(cx), float(cy), int(img_height), int(img_width), int(block_width),
            float(clip_thresh), L.ptr(cov3d), L.ptr(xys), L.ptr(depths), L.ptr(radii), L.ptr(conics),
            L.ptr(compensation), L.ptr(num_tiles_hit), sem, L.stream_ptr()), "sgn_project_fwd")
    ctx.glob_scale, ctx.fx, ctx.fy = float(glob_scale), float(fx), float(fy)
    ctx.sem, ctx.img_hw = sem, (int(img_height), int(img_width))      # the backward runs with the call's semantics
    ctx.viewmat_shape = tuple(viewmat.shape)
    ctx.mark_non_differentiable(radii, num_tiles_hit)
    # seven outputs, two or three of which the loss ever reaches: without this autograd materialises a zero tensor (an
    # allocation and a fill kernel each) for every unused one before calling backward, which handles None itself
    ctx.set_materialize_grads(False)
    return (xys, depths, radii, conics, compensation, num_tiles_hit, cov3d), \
        (means3d_c, scales_c, quats_c, viewmat_c, cov3d, radii, conics, compensation)


class _ProjectGaussiansAct(Function):
    """project_gaussians over PROVEN activations: forward on the caller's activated values, backward straight into the
    un-normalised quaternions ``x`` and the log-scale leaves (see the note above)."""
    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width,
                block_width, clip_thresh, x, *log_scale_leaves):
        ctx.bypassed = _bypassed_refs()          # (the caller's exp(...) and normalised-quaternion tensors; taken FIRST:
        #                                           nothing that raises below may leave them behind for another node)
        outs, saved = _project_forward(ctx, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height,
                                       img_width, block_width, clip_thresh)
        means3d_c, scales_c, _q, viewmat_c, cov3d, radii, conics, compensation = saved
        ctx.leaf_rows = [v.shape[0] for v in log_scale_leaves]
        ctx.arena_leaves = _arena_leaves(means3d, log_scale_leaves[0] if len(log_scale_leaves) == 1 else None, x)
        ctx.save_for_backward(means3d_c, scales_c, x, viewmat_c, cov3d, radii, conics, compensation)
        return outs

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_compensation, v_num_tiles_hit, v_cov3d):
        means3d, scales, x, viewmat, cov3d, radii, conics, compensation = ctx.saved_tensors
        n, dev = means3d.shape[0], means3d.device
        f32 = dict(dtype=torch.float32, device=dev)
        v_xys = _f32c(v_xys) if v_xys is not None else torch.zeros(n, 2, **f32)
        v_depths = _f32c(v_depths) if v_depths is not None else None
        v_conics = _f32c(v_conics) if v_conics is not None else torch.zeros(n, 3, **f32)
        v_comp = _f32c(v_compensation) if v_compensation is not None else None
        by = getattr(ctx, "bypassed", ())
        asked = [_asked_for(r) for r in by] if by else []
        if any(t is not None for t in asked):
            return _ProjectGaussiansAct._backward_through_the_callers_graph(
                ctx, asked, (means3d, scales, x, viewmat, cov3d, radii, conics, compensation), v_xys, v_depths, v_conics,
                v_comp)
        al = getattr(ctx, "arena_leaves", None)
        v_mean, v_ls, v_x = _leaf_grad(al, 0, (n, 3), f32), _leaf_grad(al, 1, (n, 3), f32), _leaf_grad(al, 2, (n, 4), f32)
        L.check(L.load().sgn_project_bwd_act(
            n, L.ptr(means3d), L.ptr(scales), ctx.glob_scale, L.ptr(x), L.ptr(viewmat),
            ctx.fx, ctx.fy, L.ptr(cov3d), L.ptr(radii), L.ptr(conics), L.ptr(compensation), L.ptr(v_xys),
            L.ptr(v_depths), L.ptr(v_conics), L.ptr(v_comp), L.ptr(v_mean), L.ptr(v_ls), L.ptr(v_x),
            ctx.sem, ctx.img_hw[0], ctx.img_hw[1], L.stream_ptr()), "sgn_project_bwd_act")
        v_leaves = (v_ls,) if len(ctx.leaf_rows) == 1 else v_ls.split(ctx.leaf_rows)
        return (v_mean, None, None, None, None, None, None, None, None, None, None, None, None, v_x) + tuple(v_leaves)

    @staticmethod
    def _backward_through_the_callers_graph(ctx, asked, saved, v_xys, v_depths, v_conics, v_comp):
        """A hook / retain_grad was placed on `scales` or `quats` AFTER the call (see `_asked_for`): the plain gradients
        with respect to the activated values, from upstream's own backward kernel; a tensor that was asked for receives
        its gradient through autograd (its hooks fire, the leaves behind it accumulate), the other one's activation is
        differentiated here."""
        means3d, scales, x, viewmat, cov3d, radii, conics, compensation = saved
        n, dev = means3d.shape[0], means3d.device
        f32 = dict(dtype=torch.float32, device=dev)
        norm = x.norm(dim=-1, keepdim=True)
        q = asked[1].detach() if asked[1] is not None else x / norm       # the values the forward projected with
        q = _f32c(q)
        v_mean, v_scale, v_quat = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 4, **f32)
        L.check(L.load().sgn_project_bwd(
            n, L.ptr(means3d), L.ptr(scales), ctx.glob_scale, L.ptr(q), L.ptr(viewmat), ctx.fx, ctx.fy,
            L.ptr(cov3d), L.ptr(radii), L.ptr(conics), L.ptr(compensation), L.ptr(v_xys), L.ptr(v_depths),
            L.ptr(v_conics), L.ptr(v_comp), None, None, L.ptr(v_mean), L.ptr(v_scale), L.ptr(v_quat),
            ctx.sem, ctx.img_hw[0], ctx.img_hw[1], L.stream_ptr()), "sgn_project_bwd")
        through, grads = [], []
        if asked[0] is not None:
            through.append(asked[0]); grads.append(v_scale.reshape(asked[0].shape))
            v_leaves = (None,) * len(ctx.leaf_rows)
        else:
            v_ls = v_scale * scales                                        # d exp(l) / d l = exp(l)
            v_leaves = (v_ls,) if len(ctx.leaf_rows) == 1 else v_ls.split(ctx.leaf_rows)
        if asked[1] is not None:
            through.append(asked[1]); grads.append(v_quat.reshape(asked[1].shape))
            v_x = None
        else:
            v_x = (v_quat - q * (q * v_quat).sum(dim=-1, keepdim=True)) / norm       # d (x / |x|) / d x
        hooks_after_call_stats["project"] += 1
        torch.autograd.backward(through, grads, retain_graph=True)
        return (v_mean, None, None, None, None, None, None, None, None, None, None, None, None, v_x) + tuple(v_leaves)


# Upstream asserts `(quats.norm(dim=-1) - 1 < 1e-6).all()` on every project_gaussians call: four small kernels and a
# host sync that drains the queue in the middle of the forward pass.  "eager" (DEFAULT: a drop-in replacement raises
# where upstream raises — from the same project_gaussians call) runs the same one-sided test as ONE device pass
# (sgn_check_unit_quats), queues the projection behind it and only then waits for the flag, so the GPU projects while
# the host wakes up.  "deferred" is an opt-in (SGN_OPTIONS="quat_check=deferred", or `ops.quat_check = "deferred"` as bench.py
# does): the flag is read at the NEXT host sync the path has anyway (the intersection-count read-back inside
# rasterize_gaussians), raising the same AssertionError there — no sync of its own; "off" skips the test.
quat_check = config.value("quat_check")        # (sgn_rast/config.py; SGN_OPTIONS="quat_check=deferred")


def _check_quats(quats: torch.Tensor):
    """Starts the argument check; returns None or a token for :func:`_finish_quat_check` (eager mode)."""
    if quat_check == "off":
        return None
    if quat_check == "eager-upstream" or not quats.is_cuda:      # upstream's literal expression (A/B, CPU tensors)
        assert (quats.norm(dim=-1) - 1 < 1e-6).all(), "quats must be normalized"
        return None
    q = _f32c(quats)
    flag = torch.empty(1, dtype=torch.int32, device=q.device)
    L.check(L.load().sgn_check_unit_quats(q.shape[0], L.ptr(q), 1e-6, L.ptr(flag), L.stream_ptr()),
            "sgn_check_unit_quats")
    S = _S()
    if quat_check == "eager":
        if S.eager_side is None:
            S.eager_side = [torch.zeros(8, dtype=torch.int32).pin_memory(), 0]
        ring = S.eager_side
        slot = ring[0][ring[1] % 8:ring[1] % 8 + 1]
        ring[1] += 1
        slot.copy_(flag, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return slot, done, flag
    if len(S.pending_checks) >= 16:    # projections without a rasterize call in between: settle the backlog now
        raise_pending_checks()
    S.pending_checks.append(flag)
    return None


def _finish_quat_check(token) -> None:
    if token is not None:
        slot, done, _keep = token
        done.synchronize()
        assert int(slot[0]) == 0, "quats must be normalized"


def raise_pending_checks() -> None:
    """Called right after an existing host sync: the flags are already final, reading them costs no stall."""
    failed, pending = False, _S().pending_checks
    while pending:
        failed |= bool(int(pending.pop().item()))
    assert not failed, "quats must be normalized"


# ------------------------------------------------------------ early depth rank
# The depth ranking of the Gaussians (a 4-pass radix sort over N keys, ~80 us) needs depths and radii only.  On the
# drop-in path with its default EAGER argument check (and with the reference's own `radii.sum() == 0` right after the
# projection) the host loses its lead over the device at that sync: the caller's view directions, SH, clamp and
# sigmoid are then launched one by one in front of an idle GPU, and `rasterize_gaussians` starts its binning late.
# `project_gaussians` therefore queues the ranking right behind the projection (same stream, BEFORE it waits for the
# check), and the coming `rasterize_gaussians` on these very depths / radii picks it up (sgn_bin_prepare(rank_ready)).
# Same kernels, same results, started earlier.  Speculative: a projection that no rasterize call follows wastes the
# ranking; after three such misses in a row the speculation pauses for 200 calls.
early_rank = config.value("early_rank")                  # "auto": with the eager check; "on"; "off"
# "main": behind the projection on the caller's stream.  "aux": on the library's second stream, forked behind the
# projection and joined by the rasterize call — a host sync of the CALLER's stream right after the projection (the
# reference's `radii.sum() == 0`, sgn_splatfacto.py:878) then returns while the ranking still runs
early_rank_stream = "main"      # ("aux" is kept for the record: -2 % on the default step, +1 % with the caller's syncs)
early_rank_stats = {"started": 0, "used": 0}


def _early_rank_wanted(n: int, is_cuda: bool) -> bool:
    """Speculation policy of the early depth rank (see above); keeps the miss / pause bookkeeping."""
    if early_rank == "off" or (early_rank == "auto" and quat_check not in ("eager", "eager-upstream")):
        return False
    if n == 0 or not is_cuda or not tile_culling_enabled:
        return False
    _early = _S().early
    if _early["pause"] > 0:
        _early["pause"] -= 1
        return False
    if _early["entry"] is not None:                 # the previous ranking was never used
        _early["misses"] += 1
        if _early["misses"] >= 3:
            _early["misses"], _early["pause"], _early["entry"] = 0, 200, None
            return False
    return True


def _start_early_rank(depths: torch.Tensor, radii: torch.Tensor) -> None:
    n = depths.shape[0]
    if not _early_rank_wanted(n, depths.is_cuda):
        return
    _early = _S().early
    lib = L.load()
    d, r = depths.detach(), radii.detach()
    key = (d.data_ptr(), d._version, r.data_ptr(), r._version, n, L.stream_handle())
    done = None
    if early_rank_stream == "aux":
        main = torch.cuda.current_stream(d.device)
        aux = L.aux_stream(d.device)
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            gid = torch.empty(n, dtype=torch.int32, device=d.device)
            ws = L.workspace(lib.sgn_depth_rank_workspace_bytes(n), d.device)
            L.check(lib.sgn_depth_rank(n, L.ptr(d), L.ptr(r), L.ptr(gid), L.ptr(ws), ws.numel(), L.sort_rank_mode(),
                                       C.c_void_p(aux.cuda_stream)), "sgn_depth_rank")
            done = torch.cuda.Event()
            done.record(aux)
        d.record_stream(aux); r.record_stream(aux)
    else:
        gid = torch.empty(n, dtype=torch.int32, device=d.device)
        ws = L.workspace(lib.sgn_depth_rank_workspace_bytes(n), d.device)
        L.check(lib.sgn_depth_rank(n, L.ptr(d), L.ptr(r), L.ptr(gid), L.ptr(ws), ws.numel(), L.sort_rank_mode(),
                                   L.stream_ptr()), "sgn_depth_rank")
    _early["entry"] = dict(key=key, keep=(d, r), gid=gid, done=done)
    early_rank_stats["started"] += 1


def _take_early_rank(depths: torch.Tensor, radii: torch.Tensor):
    """The ranking `project_gaussians` started for exactly these tensors (float32 depths / int32 radii), or None."""
    _early = _S().early
    e = _early["entry"]
    if e is None or depths.dtype != torch.float32 or radii.dtype != torch.int32:
        return None
    if e["key"] != (depths.data_ptr(), depths._version, radii.data_ptr(), radii._version, depths.shape[0],
                    L.stream_handle()):
        return None
    _early["entry"], _early["misses"] = None, 0
    early_rank_stats["used"] += 1
    if e["done"] is not None:              # ranked on the auxiliary stream: join it here
        main = torch.cuda.current_stream(depths.device)
        main.wait_event(e["done"])
        e["gid"].record_stream(main)
    return e["gid"]


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width,
                      block_width, clip_thresh: float = 0.01):
    """gsplat/project_gaussians.py project_gaussians (sgn_splatfacto.py:860-873).

    Returns ``(xys, depths, radii, conics, compensation, num_tiles_hit, cov3d)``;
    ``viewmat`` is the world->camera matrix ([3,4] or [4,4]; only rows 0-2 are read)."""
    assert block_width > 1 and block_width <= 16, "block_width must be between 2 and 16"
    plan = None
    if (composite_forward and quat_check in ("eager", "off") and quats.is_cuda and means3d.is_cuda
            and early_rank_stream == "main" and means3d.shape[-2] >= 1 and means3d.shape[-1] == 3):
        plan = dict(check=quat_check == "eager", rank=_early_rank_wanted(means3d.shape[-2], True), done=False, wait=None)
    token = _check_quats(quats) if plan is None else None
    _call_state.project_plan = plan
    ls_leaves = x = None
    if _proofs_on(activation_proofs) and scales.is_cuda and not viewmat.requires_grad:   # (viewmat gradient: plain node)
        ls_leaves = proofs.exp_leaves(scales)
        x = proofs.normalised_source(quats) if ls_leaves is not None else None
    if x is not None:
        activation_proof_stats["project"] += 1
        _call_state.bypassed = (scales, quats)
        out = _ProjectGaussiansAct.apply(means3d.contiguous(), scales.detach().contiguous(), glob_scale,
                                         quats.detach().contiguous(), viewmat.contiguous(), fx, fy, cx, cy,
                                         img_height, img_width, block_width, clip_thresh, x, *ls_leaves)
    else:
        out = _ProjectGaussians.apply(means3d.contiguous(), scales.contiguous(), glob_scale, quats.contiguous(),
                                      viewmat.contiguous(), fx, fy, cx, cy, img_height, img_width, block_width,
                                      clip_thresh)
    if plan is not None and plan["done"]:
        if plan.get("wait") is not None:
            slot, stamp = plan["wait"]
            bad = C.c_int32(0)
            L.check(L.load().sgn_project_check_wait(slot.data_ptr(), stamp, C.byref(bad), L.stream_ptr()),
                    "sgn_project_check_wait")
            assert bad.value == 0, "quats must be normalized"        # (upstream raises from this very call)
        return out
    _start_early_rank(out[1], out[2])  # depth ranking queued behind the projection, before the host waits
    _finish_quat_check(token)          # eager mode: the projection is already queued while the host waits here
    return out


# ----------------------------------------------------------------- binning
def compute_cumulative_intersects(num_tiles_hit: torch.Tensor) -> Tuple[int, torch.Tensor]:
    """gsplat/utils.py compute_cumulative_intersects: (num_intersects, cum_tiles_hit int32).
    Reads the total back to the host (the buffers below are sized by it), as upstream does."""
    dev = L.require_device(num_tiles_hit)
    nth = num_tiles_hit.detach().to(torch.int32).contiguous()
    n = nth.numel()
    cum = torch.empty_like(nth)
    if n == 0:
        return 0, cum
    lib = L.load()
    ws = L.workspace(lib.sgn_scan_workspace_bytes(n), dev)
    L.check(lib.sgn_scan_i32(n, L.ptr(nth), L.ptr(cum), L.ptr(ws), ws.numel(), L.stream_ptr()), "sgn_scan_i32")
    total = int(cum[-1].item())
    raise_pending_checks()
    return total, cum


def map_gaussian_to_intersects(num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds,
                               block_width):
    """gsplat/utils.py map_gaussian_to_intersects -> (isect_ids int64 [I], gaussian_ids int32 [I])."""
    dev = L.require_device(xys, depths, radii, cum_tiles_hit)
    # (zero-filled, as upstream's torch.zeros: slots the emission does not reach — a `num_intersects` / `cum_tiles_hit` that
    # does not belong to these xys / radii — then read as tile 0 / id 0 instead of as whatever the allocator left there)
    keys = torch.zeros(num_intersects, dtype=torch.int64, device=dev)
    vals = torch.zeros(num_intersects, dtype=torch.int32, device=dev)
    if num_intersects > 0:
        L.check(L.load().sgn_map_isect(
            num_points, L.ptr(_f32c(xys)), L.ptr(_f32c(depths)), L.ptr(radii.contiguous()),
            L.ptr(cum_tiles_hit.contiguous()), int(tile_bounds[0]), int(tile_bounds[1]), int(block_width),
            L.ptr(keys), L.ptr(vals), semantics().flags(), L.stream_ptr()), "sgn_map_isect")
    return keys, vals


def sort_intersects(isect_ids: torch.Tensor, gaussian_ids: torch.Tensor, n_tiles: int):
    """Stable (tile, depth) sort of the intersection pairs; replaces torch.sort + gather upstream."""
    dev = L.require_device(isect_ids, gaussian_ids)
    n = isect_ids.numel()
    keys_sorted = torch.empty_like(isect_ids)
    vals_sorted = torch.empty_like(gaussian_ids)
    if n > 0:
        lib = L.load()
        ws = L.workspace(lib.sgn_sort_workspace_bytes(n), dev)
        end_bit = 32 + max(1, int(n_tiles - 1).bit_length())
        L.check(lib.sgn_sort_pairs(n, 0, end_bit, L.ptr(isect_ids), L.ptr(gaussian_ids), L.ptr(keys_sorted),
                                   L.ptr(vals_sorted), L.ptr(ws), ws.numel(), L.sort_rank_mode(), L.stream_ptr()),
                "sgn_sort_pairs")
    return keys_sorted, vals_sorted


def get_tile_bin_edges(num_intersects, isect_ids_sorted, tile_bounds) -> torch.Tensor:
    """gsplat/utils.py get_tile_bin_edges -> int32 [n_tiles, 2]."""
    dev = L.require_device(isect_ids_sorted)
    n_tiles = int(tile_bounds[0]) * int(tile_bounds[1])
    bins = torch.empty(n_tiles, 2, dtype=torch.int32, device=dev)
    L.check(L.load().sgn_tile_bins(num_intersects, L.ptr(isect_ids_sorted), n_tiles, L.ptr(bins),
                                   L.stream_ptr()), "sgn_tile_bins")
    return bins


def bin_and_sort_gaussians(num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds,
                           block_width):
    """gsplat/utils.py bin_and_sort_gaussians ->
    (isect_ids_unsorted, gaussian_ids_unsorted, isect_ids_sorted, gaussian_ids_sorted, tile_bins)."""
    isect_ids, gaussian_ids = map_gaussian_to_intersects(
        num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds, block_width)
    n_tiles = int(tile_bounds[0]) * int(tile_bounds[1])
    isect_ids_sorted, gaussian_ids_sorted = sort_intersects(isect_ids, gaussian_ids, n_tiles)
    tile_bins = get_tile_bin_edges(num_intersects, isect_ids_sorted, tile_bounds)
    return isect_ids, gaussian_ids, isect_ids_sorted, gaussian_ids_sorted, tile_bins


tile_culling_enabled = bool(config.value("tile_culling"))   # exact alpha-cutoff tile culling inside rasterize_gaussians (results unchanged)
# Quadrant masks (r03): with the culling on (16x16 tiles) the emission also decides, per (tile, Gaussian) pair, which of
# the tile's four 8x8 quadrants the Gaussian can reach and hands the four bits to the raster kernels in the top of the
# id word (include/sgn_rast.h: sgn_bin_intersect(quadrant_masks)); results unchanged (tests/test_gpu_quadrant_masks.py).
# What they cost and earn (profiles/r03x_*): the emission pays per LISTED pair (+14 us event-timed, +31 us inside the
# step, for the benchmark scene's 8.3 M pairs), the raster kernels earn per WALKED entry (forward -3.5 %, backward -5.8 %).
# Where tiles saturate early (benchmark scene: a tenth of the listed entries is ever walked) that is a loss of ~1 %;
# where they do not (street-like and translucent content: 90-100 % walked) a gain of 2-3 %.  So "auto" (default) turns
# them on when the last backward's tile order reported that at least QMASK_MIN_WALKED_PERMILLE of the listed entries
# were walked (sgn_tile_order's statistic, read back with the next binning's count); "on" / "off" force it.
quadrant_masks = config.value("quadrant_masks")
QMASK_MIN_WALKED_PERMILLE = 200       # break-even measured at ~150 (31 us / 8.3 M listed vs 20.5 us / 0.84 M walked)
QMASK_ID_BITS = 28
quadrant_mask_stats = {"binnings_with_masks": 0, "walked_permille": None}


def _quadrant_masks_wanted() -> bool:
    if quadrant_masks in ("on", "1", True):
        return True
    if quadrant_masks == "auto":
        w = _S().walked_permille
        return w is not None and w >= QMASK_MIN_WALKED_PERMILLE
    return False


def _ids_only(ids: torch.Tensor) -> torch.Tensor:
    """The Gaussian ids of a list, whatever rides on top of them."""
    return (ids & ((1 << QMASK_ID_BITS) - 1)) if getattr(ids, "_sgn_qmask", False) else ids


def bin_gaussians_fused(num_points, xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics=None,
                        opacity=None, opacity_is_logit=False, cull=False):
    """The binning rasterize_gaussians actually runs: rank the Gaussians by depth once, emit the
    intersections in rank order, stable-sort them by tile id only, emit gaussian_ids_sorted + tile_bins.
    With ``cull=False`` the result is bit-identical to compute_cumulative_intersects + bin_and_sort_gaussians
    (see tests) at about a third of the HBM traffic.  With ``cull=True`` (needs ``conics`` and ``opacity``)
    (tile, Gaussian) pairs that cannot reach alpha >= 1/255 on any pixel centre of the tile are dropped: the
    list becomes a sub-sequence of upstream's and the rasterizer's outputs are unchanged.
    Returns (num_intersects, gaussian_ids_sorted, tile_bins)."""
    n_isect, ids, bins = _bin_finish(_bin_prepare_async(num_points, xys, depths, radii, num_tiles_hit, tile_bounds,
                                                        block_width, conics, opacity, opacity_is_logit, cull))
    return n_isect, _ids_only(ids), bins




def _bin_prepare_async(num_points, xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics, opacity,
                       opacity_is_logit, cull):
    """First half of the fused binning: depth rank, kept-tile counts, scan — and the asynchronous read-back of the
    intersection count into pinned memory, so the caller may queue independent work behind it before it calls
    :func:`_bin_finish` (which is where the host waits, for the copy only)."""
    dev = L.require_device(xys, depths, radii, num_tiles_hit, conics, opacity)
    lib = L.load()
    n = int(num_points)
    tx, ty = int(tile_bounds[0]), int(tile_bounds[1])
    i32 = dict(dtype=torch.int32, device=dev)
    st = dict(n=n, tx=tx, ty=ty, block=int(block_width), dev=dev, tile_bins=torch.empty(tx * ty, 2, **i32))
    if n == 0:
        return st
    radii_c = _i32c(radii)
    xys_c = _f32c(xys)
    do_cull = int(bool(cull and conics is not None and opacity is not None))
    st["qmask"] = bool(do_cull and int(block_width) == 16 and n < (1 << QMASK_ID_BITS) and _quadrant_masks_wanted())
    conics_c = _f32c(conics) if do_cull else None
    opac_c = _f32c(opacity).reshape(-1) if do_cull else None
    cum_r = torch.empty(n, **i32)
    early = _take_early_rank(depths, radii)          # started by project_gaussians behind the projection?
    gid_by_rank = early if early is not None else torch.empty(n, **i32)
    bin_recs = torch.empty(n, 8, dtype=torch.float32, device=dev)
    ws = L.workspace(lib.sgn_bin_prepare_workspace_bytes(n), dev)
    L.check(lib.sgn_bin_prepare(n, L.ptr(xys_c), L.ptr(_f32c(depths)), L.ptr(radii_c), L.ptr(conics_c), L.ptr(opac_c),
                                int(bool(opacity_is_logit)), do_cull, tx, ty, int(block_width), L.ptr(cum_r),
                                L.ptr(gid_by_rank), int(early is not None), L.ptr(bin_recs), L.ptr(ws), ws.numel(),
                                L.sort_rank_mode(), semantics().flags(), L.stream_ptr()),
            "sgn_bin_prepare")
    S = _S()
    if S.side is None:
        S.side = [torch.empty(4, 8, dtype=torch.int32).pin_memory(), 0]
    pool = S.side
    pinned = pool[0][pool[1] % 4]          # rotate: a prepare that is still pending keeps its own slot
    pool[1] += 1
    # pending argument checks ride along: their flags reach the host in the same transfer as the count, so the
    # deferred assertion costs no round trip of its own.  The copies are queued on the current stream (a side stream
    # would add an event hop of ~15 us before the copy even starts); work queued afterwards simply follows them.
    flags = [S.pending_checks.pop() for _ in range(min(len(S.pending_checks), 6))]
    pinned[0:1].copy_(cum_r[n - 1:n], non_blocking=True)
    for i, f in enumerate(flags):
        pinned[1 + i:2 + i].copy_(f, non_blocking=True)
    # the last backward's walked / listed statistic rides along too — not every time: each tiny device-to-host copy is a
    # ~5 us item on the queue, and the policy it feeds does not need a fresh value per step
    walk_stat, S.walk_stat = S.walk_stat, None
    if walk_stat is not None and S.walked_permille is not None and S.stat_skipped < 7:
        S.stat_skipped += 1
        walk_stat = None
    if walk_stat is not None:
        S.stat_skipped = 0
        pinned[7:8].copy_(walk_stat, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    st.update(cum_r=cum_r, gid_by_rank=gid_by_rank, bin_recs=bin_recs, ws=ws, done=done, pinned=pinned,
              n_flags=len(flags), has_walk_stat=walk_stat is not None, keep=(xys_c, radii_c, conics_c, opac_c, flags))
    return st


speculative_binning = bool(config.value("speculative_binning"))               # queue emission + tile sort behind
                               # the count copy, sized from the previous call
_SPEC_MARGIN = 1.3             # capacity = recent peak count of the same (n, tile grid) x this


def _bin_finish(st):
    """Second half: wait for the count (the path's one host sync, as upstream), emit, tile sort, bins.

    The count only sizes buffers, so when earlier calls with the same number of Gaussians and the same tile grid
    left their counts behind, emission + tile sort + bins are queued FIRST — into buffers sized 1.3x the recent peak,
    the kernels reading the true count on the device (`sgn_bin_intersect`'s speculative form) — and the host waits
    afterwards: the GPU works through ~0.16 ms of binning instead of idling through the host's wake-up, its
    allocations and launches.  How much that is worth depends on the host: on a box whose launches were slow the
    traced step went from 79 % to 94 % GPU-busy (1.79 -> 1.51 ms), on a fast one nothing changes (profiles/
    r02k_spec_gaps.md).  The result is taken only if the true count fits the capacity; otherwise (the scene or the
    view changed abruptly) the plain form runs with the real count, exactly as without speculation."""
    i32 = dict(dtype=torch.int32, device=st["dev"])
    tile_bins, n = st["tile_bins"], st["n"]
    binning_stats["binnings"] += 1
    if n == 0:
        tile_bins.zero_()
        return 0, torch.zeros(0, **i32), tile_bins
    lib = L.load()
    def run(count_or_cap, count_dev):
        ids = torch.empty(count_or_cap, **i32)
        ws2 = L.workspace(lib.sgn_bin_intersect_workspace_bytes(count_or_cap), st["dev"])
        L.check(lib.sgn_bin_intersect(n, count_or_cap, L.ptr(st["bin_recs"]), L.ptr(st["cum_r"]),
                                      L.ptr(st["gid_by_rank"]), st["tx"], st["ty"], st["block"], L.ptr(ids),
                                      L.ptr(tile_bins), int(st["qmask"]), L.ptr(ws2), ws2.numel(), count_dev,
                                      L.sort_rank_mode(), L.stream_ptr()),
                "sgn_bin_intersect")
        return ids

    def tagged(ids):
        ids._sgn_qmask = st["qmask"]        # what the list carries travels with the list (cache, window passes)
        quadrant_mask_stats["binnings_with_masks"] += int(st["qmask"])
        return ids

    key = (st["dev"], n, st["tx"], st["ty"], st["block"])
    cap, spec_ids = 0, None
    S = _S()
    _last_count = S.last_count
    skip_spec, S.no_speculation_once = S.no_speculation_once, False
    if speculative_binning and _last_count.get(key, 0) > 0 and not skip_spec:
        cap = min(int(_last_count[key] * _SPEC_MARGIN) + 1024, (1 << 31) - 1)
        spec_ids = run(cap, C.c_void_p(st["cum_r"].data_ptr() + 4 * (n - 1)))
    st["done"].synchronize()
    num_intersects = int(st["pinned"][0])
    if st.get("has_walk_stat"):
        S.walked_permille = quadrant_mask_stats["walked_permille"] = int(st["pinned"][7])
    failed = any(int(st["pinned"][1 + i]) for i in range(st["n_flags"]))
    assert not failed, "quats must be normalized"
    if S.pending_checks:                   # checks queued after the prefetch (rare): one more read-back
        raise_pending_checks()
    _last_count[key] = max(num_intersects, int(0.9 * _last_count.get(key, 0)))
    if spec_ids is not None and 0 < num_intersects <= cap:
        binning_stats["speculative_hits"] += 1
        return num_intersects, tagged(spec_ids[:num_intersects]), tile_bins
    if spec_ids is not None:
        binning_stats["speculative_misses"] += 1
    if num_intersects < 1:
        tile_bins.zero_()
        return num_intersects, torch.zeros(0, **i32), tile_bins
    return num_intersects, tagged(run(num_intersects, None)), tile_bins


binning_stats = {"speculative_hits": 0, "speculative_misses": 0, "binnings": 0}


tile_order_enabled = bool(config.value("tile_order"))
tile_order_multiblock = True           # the multi-workgroup form of sgn_tile_order (the single-workgroup one serves callers without scratch)
# lend sgn_raster_bwd a second stream: its short-walk and long-walk kernels overlap ("0": one stream, long walks first)
concurrent_backward = bool(config.value("concurrent_backward"))
small_splat_q16 = 0       # experimental: > 0 sends tiles with < q/16 evaluated (entry, quadrant) pairs per walked entry to the 4-waves kernel (no gain measured: profiles/r02_street_balance.md)


def _fwd_long_thresh(ro, block_width: int = 16) -> int:
    """Forward lists with at least this many entries get four waves in the packed forward (16x16 tiles; 0: no such
    prefix)."""
    return int(ro.adapt_fwd if ro.adapt_fwd > 0 else 1024) if int(block_width) == 16 else 0


def _tile_order(tile_bins: torch.Tensor, tile_kmax: Optional[torch.Tensor] = None, long_thresh: int = 0,
                pairs_known: bool = True):
    """Launch order of the raster workgroups (longest first).  Forward: by depth-list length, one tiny kernel per
    binning, shared by every pass that reuses it.  Backward (``tile_kmax`` from its forward): by reverse-walk length,
    with the count of walks >= ``long_thresh`` behind the permutation (the two-kernel adaptive scheme)."""
    if not tile_order_enabled:
        return None
    oc = _S().order_cache
    ok = (id(tile_bins), int(long_thresh))
    if tile_kmax is None and ok in oc and oc[ok][0] is tile_bins:
        oc.move_to_end(ok)
        return oc[ok][1]
    n_tiles = tile_bins.shape[0]
    order = torch.empty(n_tiles + 2, dtype=torch.int32, device=tile_bins.device)   # permutation, n_long, cursor
    lib = L.load()
    S = _S()
    scratch = S.order_scratch.get(n_tiles) if tile_order_multiblock else None
    if scratch is None and tile_order_multiblock:
        # zero-filled once per (stream, tile count); every launch leaves it zero-filled (include/sgn_rast.h)
        if len(S.order_scratch) > 8:
            S.order_scratch.clear()
        scratch = S.order_scratch[n_tiles] = torch.zeros(int(lib.sgn_tile_order_scratch_bytes(n_tiles)) // 4,
                                                         dtype=torch.int32, device=tile_bins.device)
    L.check(lib.sgn_tile_order(n_tiles, L.ptr(tile_bins), L.ptr(tile_kmax), int(long_thresh),
                               int(small_splat_q16) if (tile_kmax is not None and pairs_known) else 0, L.ptr(order),
                               L.ptr(scratch),
                               4 * scratch.numel() if scratch is not None else 0, L.stream_ptr()), "sgn_tile_order")
    if tile_kmax is None:
        oc[ok] = (tile_bins, order)          # (keeps tile_bins alive: its id cannot be recycled while the entry lives)
        while len(oc) > 4:
            oc.popitem(last=False)
    return order


# One-entry binning cache: the reference renders RGB and depth from the SAME projection in two
# consecutive rasterize_gaussians calls (sgn_splatfacto.py:954-967 then :982-994); the second call
# reuses the first call's sorted list and bins when its geometry inputs are the very same tensors
# (same storage, same autograd version counters), instead of ranking / emitting / sorting again.
# The cache keeps detached aliases of the four geometry tensors, so their storage cannot be freed
# and re-used by the allocator while the entry is alive: equal data_ptr + equal version counter then
# really means "same bytes".
binning_cache_enabled = bool(config.value("binning_cache"))


def _bin_key(tensors, tile_bounds, block_width, flags):
    return tuple((t.data_ptr(), t._version, t.shape, t.stride(), t.dtype) for t in tensors) + (
        tuple(int(b) for b in tile_bounds), int(block_width), flags, L.stream_handle())


def clear_binning_cache() -> None:
    """Forget every cached binning (all devices / streams)."""
    for st in _states.values():
        st.clear_binning()




def _cache_key(xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics, opacity, opacity_is_logit):
    cull = tile_culling_enabled
    tensors = (xys, depths, radii, num_tiles_hit) + ((conics, opacity) if cull else ())
    return _bin_key(tensors, tile_bounds, block_width,
                    (bool(opacity_is_logit), cull, semantics().tile_bbox_add_after_cast)), tensors, cull


def _drop_pending() -> None:
    """A prepared binning that will never be finished (its rasterize call did not come, or a later prefetch replaces
    it) still carries deferred argument-check flags in its read-back slot: hand them back to the backlog so the
    "quats must be normalized" assertion is not lost (advisor finding, round 1)."""
    S = _S()
    _bin_pending = S.bin_pending
    st = _bin_pending["state"]
    if st is not None and st.get("keep") is not None:
        S.pending_checks.extend(st["keep"][-1])
    _bin_pending["key"] = _bin_pending["state"] = _bin_pending["keep"] = None


def prefetch_binning(xys, depths, radii, conics, num_tiles_hit, opacity, img_height, img_width, block_width,
                     opacity_is_logit=False) -> None:
    """Optional hint for callers that have other device work to queue between projection and rasterization (the SH
    evaluation, typically): starts the binning of the coming ``rasterize_gaussians`` call now — depth rank, counts,
    scan, and the asynchronous read-back of the intersection count — so that by the time ``rasterize_gaussians`` needs
    the count on the host the GPU is still busy with the work queued in between, instead of idling while the host
    wakes up and launches the second half.  The rasterize call must receive these very tensors."""
    tile_bounds = ((img_width + block_width - 1) // block_width, (img_height + block_width - 1) // block_width, 1)
    key, tensors, cull = _cache_key(xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics, opacity,
                                    opacity_is_logit)
    S = _S()
    if binning_cache_enabled and S.has_binning(key):
        return
    _drop_pending()
    _bin_pending = S.bin_pending
    _bin_pending["state"] = _bin_prepare_async(xys.size(0), xys, depths, radii, num_tiles_hit, tile_bounds, block_width,
                                               conics, opacity, opacity_is_logit, cull)
    _bin_pending["key"] = key
    _bin_pending["keep"] = tuple(t.detach() for t in tensors)


def _bin_gaussians_cached(num_points, xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics,
                          opacity, opacity_is_logit, pre=None, logit_leaves=()):
    # `pre`: the caller's own (key, tensors, cull) of these very arguments (the key walks six tensors)
    key, tensors, cull = pre if pre is not None else _cache_key(
        xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics, opacity, opacity_is_logit)
    S = _S()
    if binning_cache_enabled:
        val = S.find_binning(key)
        if val is not None:
            return val
    _bin_pending = S.bin_pending
    if _bin_pending["key"] == key:
        state = _bin_pending["state"]
    else:
        _drop_pending()
        state = _bin_prepare_async(num_points, xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics,
                                   opacity, opacity_is_logit, cull)
    _bin_pending["key"] = _bin_pending["state"] = _bin_pending["keep"] = None
    val = _bin_finish(state)
    if binning_cache_enabled:
        S.store_binning(key, tuple(t.detach() for t in tensors), val, _window_info(tensors, cull, logit_leaves))
    return val


# ------------------------------------------------------- window recognition
# The scene graph renders its sub-model passes from torch.cat COPIES of per-model slices of the main projection
# (sgn_splatfacto_scene_graph.py:270-276, passes at :364-366): new tensors every time, so the identity-keyed cache
# above cannot hit, and re-binning each slice cost two more rank/emit/sort rounds per step (121 vs 390 images/s in
# round 1).  Those copies are still BIT-IDENTICAL to a row window of the geometry the cached list was binned for —
# the background is the head of the concatenation, the objects its tail — so a call whose tensors are smaller than
# the cached scene is compared against those two windows on the device (sgn_rows_match: one pass over the window,
# a few MB) and, on a match, rasterized over the CACHED list with an id range: no ranking, no emission, no sort.
# One host read-back of the verdict replaces the intersection-count read-back of the re-binning it avoids.
window_matching_enabled = bool(config.value("window_matching"))
window_stats = {"tried": 0, "hit": 0}


def _window_info(tensors, cull, logit_leaves=()):
    """What the autograd graph says about a binning's geometry tensors (kept with the most recent cache entry): the
    split / concatenation shape of xys, depths, conics (proofs.split_cat) and the logit leaves behind the opacities with
    their version counters — the facts a later sub-model call is matched against WITHOUT touching the device."""
    if not (window_matching_enabled and _proofs_on(activation_proofs)):
        return None
    xys, depths = tensors[0], tensors[1]
    if xys.grad_fn is None:
        return None
    info = dict(xys=proofs.split_cat(xys), depths=proofs.split_cat(depths))
    if cull:
        info["conics"] = proofs.split_cat(tensors[4])
        # the rasterize wrapper has already proven `opacity == sigmoid(cat(logit_leaves))` (and detached it)
        leaves = tuple(logit_leaves) or proofs.sigmoid_leaves(tensors[5])
        info["opacity"] = None if not leaves else (leaves, tuple(v._version for v in leaves))
    return info


def _proven_window(info, xys, depths, conics, opacity, cull, logit_leaves=()):
    """(lo, hi) if the graph PROVES that xys / depths / conics / opacities of this call are copies of rows [lo, hi) of the
    cached scene's tensors (same split of the same projection, a run of the same logit leaves, nothing written since)."""
    if info is None or info.get("xys") is None:
        return None
    win = proofs.window_of_split(proofs.split_cat(xys), info["xys"])
    if win is None or proofs.window_of_split(proofs.split_cat(depths), info["depths"]) != win:
        return None
    if cull:
        if proofs.window_of_split(proofs.split_cat(conics), info.get("conics")) != win or info.get("opacity") is None:
            return None
        full_leaves, versions = info["opacity"]
        mine = tuple(logit_leaves) or proofs.sigmoid_leaves(opacity)
        if not mine or proofs.window_of_leaves(mine, full_leaves) != win:
            return None
        if any(v._version != ver for v, ver in zip(full_leaves, versions)):
            return None
    return win


def _match_window(key_tail, n, xys, depths, radii, num_tiles_hit, conics, opacity, cull, logit_leaves=()):
    """(lo, cached value, n_full) when the call's geometry equals rows [lo, lo + n) of the cached scene, else None.

    Two stages.  The autograd graph usually settles four of the six tensors on the host (`_proven_window`: the
    sub-model's xys / depths / conics are concatenations of parts of the very split the cached tensors were concatenated
    from, its opacities the sigmoid of a run of the same leaves); the two integer tensors (radii, num_tiles_hit) carry no
    graph and are compared on the device — 8 bytes per row at the one proven offset.  Without a proof all six tensors are
    compared at the two candidate offsets (head and tail window), 36 bytes per row."""
    S = _S()
    ck, keep, val = S.bin_cache["key"], S.bin_cache["keep"], S.bin_cache["val"]
    if ck is None or not window_matching_enabled or not binning_cache_enabled or n <= 0:
        return None
    if ck[len(keep):] != key_tail or val[0] < 1:       # tile grid, block, flags (logit / cull), stream
        return None
    n_full = keep[0].shape[0]
    if n_full <= n:
        return None
    mine = (xys, depths, radii, num_tiles_hit) + ((conics, opacity) if cull else ())
    for i, (t, c) in enumerate(zip(mine, keep)):
        if t.dtype != c.dtype or t.shape[1:] != c.shape[1:] or t.device != c.device:
            return None
        # sgn_rows_match compares 32-bit words (row widths {2,1,1,1,3,1}): float32 / int32 tensors only — anything else
        # (int64 radii, float64 xys: the binning accepts and converts them) falls through to re-binning
        if t.dtype not in (torch.float32, torch.int32):
            return None
        # the cached aliases must still hold the bytes the list was binned from (key rows: data_ptr, _version, ...)
        if c._version != ck[i][1] or c.data_ptr() != ck[i][0]:
            return None
    window_stats["tried"] += 1
    proven = _proven_window(S.bin_cache["info"], xys, depths, conics, opacity, cull, logit_leaves)
    if proven is not None and proven[1] - proven[0] != n:
        proven = None
    dev = xys.device
    lib = L.load()
    cands = [proven[0]] if proven is not None else sorted({0, n_full - n})
    lo_host = (C.c_int32 * len(cands))(*cands)
    flags = torch.empty(4, dtype=torch.int32, device=dev)
    w = [t.detach().contiguous() for t in mine] + [None] * (6 - len(mine))
    f = [c.contiguous() for c in keep] + [None] * (6 - len(keep))
    if cull:
        w[5], f[5] = w[5].reshape(-1), f[5].reshape(-1)
    if proven is not None:                       # the graph settled these four: only the integer tensors go to the device
        activation_proof_stats["window"] += 1
        for i in (0, 1, 4, 5):
            w[i] = f[i] = None
    L.check(lib.sgn_rows_match(n, n_full, len(cands), lo_host, *[L.ptr(t) for t in w], *[L.ptr(t) for t in f],
                               L.ptr(flags), L.stream_ptr()), "sgn_rows_match")
    if S.side is None:
        S.side = [torch.empty(4, 8, dtype=torch.int32).pin_memory(), 0]
    pool = S.side
    pinned = pool[0][pool[1] % 4]
    pool[1] += 1
    pinned[0:len(cands)].copy_(flags[0:len(cands)], non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    done.synchronize()
    if S.pending_checks:
        raise_pending_checks()
    for i, lo in enumerate(cands):
        if int(pinned[i]) == 0:
            window_stats["hit"] += 1
            return lo, val, n_full
    return None


# A window that covers a small part of the scene (the objects-only pass: a tenth of the Gaussians) does not walk the
# shared list with everything else made inert — nothing saturates then, every tile walks its whole list, 0.5 ms per step
# at 1 M Gaussians for a pass that draws 100 k — but its own sub-list (sgn_list_window: two reads of the list, no sort,
# same relative order, hence the same image and gradients).  Larger windows (the background: 90 %) keep the shared list.
list_window_enabled = bool(config.value("list_window"))
list_window_max_frac = 0.5
window_stats["sub_lists"] = 0


def _list_window(ids: torch.Tensor, tile_bins: torch.Tensor, lo: int, hi: int, qmask: int):
    lib = L.load()
    n_tiles = tile_bins.shape[0]
    ids_out, bins_out = torch.empty_like(ids), torch.empty_like(tile_bins)
    ws = L.workspace(lib.sgn_list_window_workspace_bytes(n_tiles), ids.device)
    L.check(lib.sgn_list_window(n_tiles, L.ptr(ids), L.ptr(tile_bins), int(lo), int(hi), int(qmask), L.ptr(ids_out),
                                L.ptr(bins_out), L.ptr(ws), ws.numel(), L.stream_ptr()), "sgn_list_window")
    ids_out._sgn_qmask = bool(qmask)
    return ids_out, bins_out


# ------------------------------------------------------------ depth channel
# The reference renders depth with a SECOND full rasterization of the same geometry, the colours being the depths
# themselves (sgn_splatfacto.py:982-994).  The forward kernels can accumulate that image as a fourth channel of the
# FIRST pass (one fma per evaluated pair), and the second call is then answered from it: a device-side comparison proves
# that its colours are `depths.repeat(1, 3)`, the ordinary forward is queued behind a flag that turns its kernels into
# no-ops, and one small kernel writes the image and the per-pixel state the second node's backward needs
# (sgn_depth_reuse) — no host sync, bit-equal to the two-pass result in exact-exp mode.
# "auto" (default): the first pass starts accumulating once a step has been seen to make that second call (a rasterize
# call on the very tensors of the previous one with other colours), and stops again when the calls stop coming;
# "on" / "off" force it.  The fused API asks for the channel explicitly (rasterize_gaussians_fused(depth_channel=True)).
depth_channel = config.value("depth_channel")
depth_stats = {"accumulated": 0, "reused": 0, "proved_on_host": 0}
group_stats = {"passes": 0, "backward_passes": 0}    # forwards that carried the two group accumulations; their backwards


def _depth_wanted() -> bool:
    return depth_channel == "on" or (depth_channel == "auto" and _S().depth_state["want"])


_provably_depths = proofs.repeated_depths     # host-side proof that colours are `depths[:, None].repeat(1, 3)` (:988)


# Optional hook for data-parallel training (sgn_rast.dp.GradAllReducer(sparse=True)): `after_forward(ids, bins, kmax, n,
# qmask)` is told, right after a full (non-window) forward pass, which list entries the pass walked.  None normally.
_touch_sink = None
# A graph proof is decided when the operator is CALLED; a hook or `retain_grad()` the caller places on the bypassed tensor
# AFTERWARDS is found when the node's backward runs (round 6; rounds 3-5 documented that such a hook "sees nothing"):
# the node keeps weak references to the tensors it bypassed (handed over through `_call_state.bypassed`, not as autograd
# inputs: no edge), and if one of them has been asked for its gradient by then, the node computes the PLAIN gradient with
# respect to that tensor and lets autograd carry it from there — `torch.autograd.backward([tensor], [gradient])`, a
# re-entrant pass through the caller's own expression: its hooks fire, `.grad` is retained, the leaves accumulate exactly
# what upstream's graph would have given them.  (`retain_graph=True`: nodes further back may be shared with the outer pass,
# which still visits them — with nothing to add — and must find their saved tensors.)  The fast path pays two attribute
# reads per bypassed tensor.
def _bypassed_refs():
    by = getattr(_call_state, "bypassed", ())
    _call_state.bypassed = ()
    return tuple(None if t is None else weakref.ref(t) for t in by)


def _asked_for(ref) -> Optional[torch.Tensor]:
    """The bypassed tensor, if it is alive and somebody has asked for its own gradient since the call."""
    t = ref() if ref is not None else None
    if t is not None and (t._backward_hooks or t.retains_grad):
        return t
    return None


hooks_after_call_stats = {"project": 0, "opacity": 0, "colors": 0, "sh": 0}


# hook for sgn_rast.dp.GradAllReducer's zero-copy bucket (round 6): `_grad_arena(leaf)` -> the slice of the reducer's flat
# all-reduce buffer this step's gradient of `leaf` belongs in, or None.  A backward node that produces the gradient of an
# input that IS a registered leaf writes it there instead of into a torch.empty tensor; autograd keeps the returned view as
# `.grad` (contiguous, nobody else holds it), so the collective needs no copy in and none back.
_grad_arena = None


def _arena_leaves(*inputs):
    """forward-time: the inputs of a node that are leaves wanting a gradient (what `_leaf_grad` may look up), or None."""
    if _grad_arena is None:
        return None
    return tuple(t if (t is not None and t.is_leaf and t.requires_grad) else None for t in inputs)


def _leaf_grad(leaves, i: int, shape, f32):
    """Gradient buffer of a node's i-th candidate input: its slice of the DP bucket if it is a registered leaf, else a
    fresh tensor.  The view handed back is a NEW tensor object (AccumulateGrad steals a gradient only when it holds the
    last reference)."""
    if leaves is not None and _grad_arena is not None and leaves[i] is not None:
        sl = _grad_arena(leaves[i])
        if sl is not None:
            n = 1
            for d in shape:
                n *= int(d)
            if sl.numel() == n and sl.is_contiguous():
                return sl.view(*shape)
    return torch.empty(*shape, **f32)
# grad mode where the operator was CALLED (inside Function.forward it always reads "off", and ctx.needs_input_grad ignores
# it): the rasterize wrappers note it just before `apply` (same thread: `apply` runs the forward synchronously)
_call_state = threading.local()


# One library call per autograd node (round 5).  `sgn_rasterize_fwd_all` runs the forward's whole
# launch sequence — first half of the binning, count read-back, rows, speculative second half, the wait, launch order,
# forward kernels — behind ONE ctypes call with ONE arena for its temporaries, instead of six calls and a dozen
# allocations driven from here.  It serves the plain case (the full scene, a fresh binning, no depth channel / reuse /
# window / groups, a capacity known from earlier calls of the same shape); everything else, and a capacity miss, takes the
# call-by-call path below, which stays the reference for behaviour.  Same kernels, same order, same results.
# Option `one_call_nodes` (sgn_rast/config.py) switches it off.
composite_forward = bool(config.value("one_call_nodes"))
composite_backward = composite_forward      # the node's backward as one call too (sgn_rasterize_bwd_all, round 6)
composite_stats = {"forwards": 0, "capacity_misses": 0, "windows": 0, "backwards": 0}
_E_CAPACITY = -100


def _forward_composite(S, key, _t, cull, n, xys_c, depths, radii, conics_c, colors_c, opac_c, opacity_is_logit,
                       img_height, img_width, block_width, tile_bounds, bg_c, out_img, final_Ts, final_idx, ro, ro_ptr,
                       logit_leaves, stream_ptr, out_depth=None):
    """(num_intersects, ids, tile_bins, order, tile_kmax, rows) or None (no capacity known yet / the list did not fit:
    the caller takes the call-by-call path)."""
    if not speculative_binning or depths.dtype is not torch.float32 or radii.dtype is not torch.int32:
        return None
    dev = xys_c.device
    tx, ty = int(tile_bounds[0]), int(tile_bounds[1])
    ckey = (dev, n, tx, ty, int(block_width))
    last = S.last_count.get(ckey, 0)
    if last <= 0:
        return None
    cap = min(int(last * _SPEC_MARGIN) + 1024, (1 << 31) - 1)
    lib = L.load()
    i32 = dict(dtype=torch.int32, device=dev)
    n_tiles = tx * ty
    qmask = bool(cull and int(block_width) == 16 and n < (1 << QMASK_ID_BITS) and _quadrant_masks_wanted())
    ids = torch.empty(cap, **i32)
    # bins and tile statistics in ONE buffer, statistics right behind the bins: the emission then clears both and
    # neither costs a clear launch of its own (sgn_rasterize_fwd_all recognises the layout)
    bins_and_stats = torch.empty(2, n_tiles, 2, **i32)
    tile_bins, tile_kmax = bins_and_stats[0], bins_and_stats[1]
    order = torch.empty(n_tiles + 2, **i32)
    rows = L.workspace(lib.sgn_raster_workspace_bytes(n, 0, ro_ptr), dev)
    arena = L.workspace(lib.sgn_rasterize_arena_bytes(n, cap), dev)
    scratch = None
    if tile_order_enabled and tile_order_multiblock:
        scratch = S.order_scratch.get(n_tiles)
        if scratch is None:
            if len(S.order_scratch) > 8:
                S.order_scratch.clear()
            scratch = S.order_scratch[n_tiles] = torch.zeros(int(lib.sgn_tile_order_scratch_bytes(n_tiles)) // 4, **i32)
    early_entry = S.early["entry"]
    early = _take_early_rank(depths, radii)
    if S.side is None:
        S.side = [torch.empty(4, 8, dtype=torch.int32).pin_memory(), 0]
    pool = S.side
    pinned = pool[0][pool[1] % 4]
    pool[1] += 1
    walk_stat, S.walk_stat = S.walk_stat, None          # the last backward's walked / listed statistic rides along
    if walk_stat is not None and S.walked_permille is not None and S.stat_skipped < 7:
        S.stat_skipped += 1
        walk_stat = None
    if walk_stat is not None:
        S.stat_skipped = 0
    n_host = C.c_int64(0)
    rc = lib.sgn_rasterize_fwd_all(
        n, L.ptr(xys_c), L.ptr(_f32c(depths)), L.ptr(_i32c(radii)), L.ptr(conics_c), L.ptr(colors_c), L.ptr(opac_c),
        int(bool(opacity_is_logit)), int(bool(cull)), int(img_height), int(img_width), int(block_width), L.ptr(bg_c),
        L.ptr(early), int(qmask), L.ptr(out_img), L.ptr(final_Ts), L.ptr(final_idx), L.ptr(out_depth), L.ptr(ids), cap,
        L.ptr(tile_bins),
        L.ptr(order), L.ptr(tile_kmax), L.ptr(rows), rows.numel(), L.ptr(scratch),
        4 * scratch.numel() if scratch is not None else 0, L.ptr(arena), arena.numel(), pinned[0:1].data_ptr(),
        L.ptr(walk_stat), pinned[7:8].data_ptr() if walk_stat is not None else None, C.byref(n_host),
        L.sort_rank_mode(), semantics().flags(), ro_ptr, stream_ptr)
    if walk_stat is not None and rc in (0, _E_CAPACITY):
        S.walked_permille = quadrant_mask_stats["walked_permille"] = int(pinned[7])
    count = int(n_host.value)
    if rc == _E_CAPACITY:
        # the view sees more than 1.3x the recent peak: the call-by-call path bins again, WITHOUT speculating (its plain
        # form with the real count, exactly what its own miss does; it also records the new count)
        binning_stats["speculative_misses"] += 1
        composite_stats["capacity_misses"] += 1
        S.no_speculation_once = True
        if early is not None:              # the ranking is still valid for these depths / radii: the fallback takes it
            S.early["entry"] = early_entry
            early_rank_stats["used"] -= 1
        return None
    L.check(rc, "sgn_rasterize_fwd_all")
    binning_stats["binnings"] += 1
    S.last_count[ckey] = max(count, int(0.9 * last))
    binning_stats["speculative_hits"] += int(count >= 1)
    composite_stats["forwards"] += 1
    ids = ids[:max(count, 0)]
    ids._sgn_qmask = qmask
    quadrant_mask_stats["binnings_with_masks"] += int(qmask)
    ro.ids_qmask = int(qmask)
    if count < 1:
        return count, ids, tile_bins, order, tile_kmax, None
    if binning_cache_enabled:
        S.store_binning(key, tuple(t.detach() for t in _t), (count, ids, tile_bins), _window_info(_t, cull, logit_leaves))
    if tile_order_enabled:
        oc = S.order_cache
        oc[(id(tile_bins), _fwd_long_thresh(ro, block_width))] = (tile_bins, order)
        while len(oc) > 4:
            oc.popitem(last=False)
    return count, ids, tile_bins, order, tile_kmax, rows


def _order_scratch(S, lib, n_tiles, dev):
    """The persistent zero-filled scratch of the multi-workgroup tile order (one per stream and tile count)."""
    if not (tile_order_enabled and tile_order_multiblock):
        return None
    scratch = S.order_scratch.get(n_tiles)
    if scratch is None:
        if len(S.order_scratch) > 8:
            S.order_scratch.clear()
        scratch = S.order_scratch[n_tiles] = torch.zeros(int(lib.sgn_tile_order_scratch_bytes(n_tiles)) // 4,
                                                         dtype=torch.int32, device=dev)
    return scratch


def _window_candidate(S, key_tail, n, mine):
    """Host-only pre-check of the window recognition: the most recent binning of this stream if the call's tensors COULD
    be a row window of its scene (same tile grid / block / flags, fewer rows, same dtypes and row widths, the cached
    aliases still holding the bytes the list was binned from) -> (candidate offsets, cached value, kept tensors)."""
    ck, keep, val = S.bin_cache["key"], S.bin_cache["keep"], S.bin_cache["val"]
    if ck is None or not window_matching_enabled or not binning_cache_enabled or n <= 0:
        return None
    if ck[len(keep):] != key_tail or val[0] < 1 or len(mine) != len(keep):
        return None
    n_full = keep[0].shape[0]
    if n_full <= n:
        return None
    for i, (t, c) in enumerate(zip(mine, keep)):
        if t.dtype != c.dtype or t.shape[1:] != c.shape[1:] or t.device != c.device:
            return None
        if t.dtype not in (torch.float32, torch.int32):        # sgn_rows_match compares 32-bit words
            return None
        if c._version != ck[i][1] or c.data_ptr() != ck[i][0]:
            return None
    return sorted({0, n_full - n}), val, keep


def _forward_window_composite(S, cand, n, xys_c, depths, radii, num_tiles_hit, conics_c, colors_c, opac_c,
                              opacity_is_logit, cull, img_height, img_width, block_width, bg_c, out_img, final_Ts,
                              final_idx, ro, ro_ptr, stream_ptr):
    """ONE library call for a sub-model pass over the cached list (sgn_rasterize_window_all, round 6): the device-side
    comparison of all six tensors at the head / tail offsets, the verdict's read-back, the window's rows, its sub-list,
    the launch order and the forward kernels.  -> (lo, n_full, num_intersects, ids, bins, order, tile_kmax, rows) or None
    (not a window of that scene: the caller bins the tensors themselves)."""
    cands, (num_intersects, ids_full, bins_full), keep = cand
    n_full = keep[0].shape[0]
    dev = xys_c.device
    lib = L.load()
    i32 = dict(dtype=torch.int32, device=dev)
    n_tiles = bins_full.shape[0]
    qmask = int(bool(getattr(ids_full, "_sgn_qmask", False)))
    ro.ids_qmask = qmask
    sub = bool(list_window_enabled and n < list_window_max_frac * n_full)
    order_ready = None
    if not sub:
        hit = S.order_cache.get((id(bins_full), _fwd_long_thresh(ro, block_width)))
        if hit is not None and hit[0] is bins_full:
            order_ready = hit[1]
    ids_out = torch.empty_like(ids_full) if sub else None
    bins_out = torch.empty_like(bins_full) if sub else None
    order = torch.empty(n_tiles + 2, **i32) if order_ready is None else None
    tile_kmax = torch.empty(n_tiles, 2, **i32)
    rows = L.workspace(lib.sgn_raster_workspace_bytes(n_full, 0, ro_ptr), dev)
    arena = L.workspace(lib.sgn_rasterize_window_arena_bytes(n_tiles), dev)
    scratch = _order_scratch(S, lib, n_tiles, dev)
    if S.side is None:
        S.side = [torch.empty(4, 8, dtype=torch.int32).pin_memory(), 0]
    pool = S.side
    pinned = pool[0][pool[1] % 4]
    pool[1] += 1
    f = [c if c.is_contiguous() else c.contiguous() for c in keep] + [None] * (6 - len(keep))
    if cull:
        f[5] = f[5].reshape(-1)
    lo_host = (C.c_int32 * len(cands))(*cands)
    matched = C.c_int(-1)
    window_stats["tried"] += 1
    L.check(lib.sgn_rasterize_window_all(
        n, n_full, len(cands), lo_host, L.ptr(xys_c), L.ptr(_f32c(depths)), L.ptr(_i32c(radii)),
        L.ptr(_i32c(num_tiles_hit)), L.ptr(conics_c), L.ptr(colors_c), L.ptr(opac_c), int(bool(opacity_is_logit)),
        L.ptr(f[0]), L.ptr(f[1]), L.ptr(f[2]), L.ptr(f[3]), L.ptr(f[4]), L.ptr(f[5]), int(num_intersects),
        L.ptr(ids_full), L.ptr(bins_full), qmask, int(img_height), int(img_width), int(block_width), L.ptr(bg_c),
        int(sub), L.ptr(order_ready), L.ptr(out_img), L.ptr(final_Ts), L.ptr(final_idx), L.ptr(ids_out),
        L.ptr(bins_out), L.ptr(order), L.ptr(tile_kmax), L.ptr(rows), rows.numel(), L.ptr(scratch),
        4 * scratch.numel() if scratch is not None else 0, L.ptr(arena), arena.numel(), pinned.data_ptr(),
        C.byref(matched), ro_ptr, stream_ptr), "sgn_rasterize_window_all")
    if S.pending_checks:
        raise_pending_checks()             # (a host sync has just happened: deferred flags are final)
    if matched.value < 0:
        return None
    window_stats["hit"] += 1
    composite_stats["windows"] += 1
    if sub:
        window_stats["sub_lists"] += 1
        ids_out._sgn_qmask = bool(qmask)
        ids_use, bins_use = ids_out, bins_out
    else:
        ids_use, bins_use = ids_full, bins_full
        if order_ready is None and tile_order_enabled:
            oc = S.order_cache
            oc[(id(bins_full), _fwd_long_thresh(ro, block_width))] = (bins_full, order)
            while len(oc) > 4:
                oc.popitem(last=False)
    return matched.value, n_full, num_intersects, ids_use, bins_use, order_ready if order_ready is not None else order, tile_kmax, rows


# --------------------------------------------------------------- rasterize
class _RasterizeGaussians(Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                block_width, background=None, return_alpha=False, opacity_is_logit=False, id_range=None,
                want_depth=False, colors_are_depths=False, colors_pre=None, group_split=None, *opacity_logits):
        # opacity_logits / colors_pre (proven by rasterize_gaussians, see proofs.sigmoid_leaves / clamp_pre): `opacity`
        # is sigmoid(cat(opacity_logits, 0)) and `colors` is clamp(colors_pre, min=0); both arrive DETACHED and the
        # gradients go to the extra inputs instead (the activations' backward runs inside sgn_raster_bwd's unpack kernel)
        ctx.bypassed = _bypassed_refs()     # (the caller's sigmoid(...) / clamp(...) tensors, where they were bypassed)
        ctx.grad_to_logits, ctx.grad_to_pre = len(opacity_logits) > 0, colors_pre is not None
        ctx.logit_rows = [v.shape[0] for v in opacity_logits]
        ctx.arena_leaves = _arena_leaves(opacity_logits[0] if len(opacity_logits) == 1 else
                                         (opacity if not opacity_logits else None))
        ctx.alpha_clamp_bwd = semantics().alpha_clamp_bwd      # the backward runs with the CALL's value
        ctx.set_materialize_grads(False)       # an unused alpha / depth output arrives as None, not as a zero image
        dev = L.require_device(xys, depths, radii, conics, num_tiles_hit, colors, opacity, background)
        num_points = xys.size(0)
        tile_bounds = ((img_width + block_width - 1) // block_width,
                       (img_height + block_width - 1) // block_width, 1)
        if colors.shape[-1] != 3:
            raise NotImplementedError(
                "only the 3-channel rasterize path is implemented (the reference never uses N-D colours: "
                "sgn_splatfacto.py:988 repeats depth x3 to stay on it)")
        xys_c, conics_c, colors_c = _f32c(xys), _f32c(conics), _f32c(colors)
        opac_c, bg_c = _f32c(opacity).reshape(-1), _f32c(background)
        f32 = dict(dtype=torch.float32, device=dev)
        lib = L.load()
        ro = L.opts().copy()              # this call's kernel options: the backward runs with the same ones
        ro_ptr = C.byref(ro)
        # everything that does not depend on the intersection count is prepared BEFORE the binning's host sync: the
        # GPU idles from the moment the count is known until the next launch arrives, so that window is kept short
        out_img = torch.empty(img_height, img_width, 3, **f32)
        final_Ts = torch.empty(img_height, img_width, **f32)
        final_idx = torch.empty(img_height, img_width, dtype=torch.int32, device=dev)
        stream_ptr = L.stream_ptr()
        key, _t, cull = _cache_key(xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics, opacity,
                                   opacity_is_logit)
        S = _S()
        _bin_pending, _depth_state = S.bin_pending, S.depth_state
        hit = binning_cache_enabled and S.has_binning(key)
        # a sub-model's copy of a window of the cached scene?  (drop-in scene-graph path; see _match_window)
        n_full, window, win = num_points, 0, None
        wcomp = None
        if id_range is None and not hit and _bin_pending["key"] != key and not want_depth:
            # (an explicit depth request is never served as a window of another scene: the window's tensors are
            # window-local, the channel's depths would be read at full-scene ids — it is binned as a scene of its own)
            cand = None
            if (composite_forward and tile_order_enabled and group_split is None and not want_depth
                    and depths.dtype is torch.float32 and radii.dtype is torch.int32
                    and num_tiles_hit.dtype is torch.int32):
                cand = _window_candidate(S, key[len(_t):], num_points, _t)
            if cand is not None:
                # ONE library call: comparison on the device, verdict, rows, sub-list, order, forward (round 6)
                wcomp = _forward_window_composite(S, cand, num_points, xys_c, depths, radii, num_tiles_hit, conics_c,
                                                  colors_c, opac_c, opacity_is_logit, cull, img_height, img_width,
                                                  block_width, bg_c, out_img, final_Ts, final_idx, ro, ro_ptr, stream_ptr)
            else:
                win = _match_window(key[len(_t):], num_points, xys, depths, radii, num_tiles_hit, conics, opacity, cull,
                                    opacity_logits)
        if wcomp is not None:
            lo, n_full, num_intersects, gaussian_ids_sorted, tile_bins, order, tile_kmax, recs = wcomp
            id_lo, id_hi, window = lo, lo + num_points, 1
            win = (lo, None, n_full)                      # (a window pass for everything below: sink, ctx)
        elif win is not None:
            lo, cached, n_full = win
            id_lo, id_hi, window = lo, lo + num_points, 1
        else:
            id_lo, id_hi = (0, num_points) if id_range is None else (int(id_range[0]), int(id_range[1]))
        # depth channel: is this the reference's depth pass over the geometry of the pass before it (answer it from that
        # pass's fourth channel), or a first pass that should accumulate the channel?
        plain = win is None and id_range is None and num_points > 0 and cull
        dcache = S.depth_caches.get(key)
        reuse = (plain and hit and not want_depth and group_split is None and depth_channel != "off" and dcache is not None
                 and colors_c.shape == (num_points, 3) and dcache["D"].shape == (img_height, img_width))
        if plain and hit and not reuse and depth_channel == "auto" and not want_depth:
            _depth_state["want"] = True        # a second pass over the same geometry: accumulate from the next step on
        # `want_depth` (the fused API's depth_channel=True) is an explicit request: the channel is accumulated whatever the
        # binning's history — a cached list, culling switched off, an id_range pass (round 6: such calls used to get a zero
        # image; a call that asks for depth is not matched as a window of another scene, see above).  The
        # "auto" policy of the drop-in surface accumulates on a plain, freshly binned pass only, and only such a pass's
        # image may answer a later depth pass (`depth_cacheable`).
        explicit = bool(want_depth) and win is None and num_points > 0
        accumulate = (not reuse) and (explicit or (plain and not hit and _depth_wanted()))
        depth_cacheable = accumulate and plain and not hit
        depths_c = _f32c(depths) if (reuse or accumulate) else None
        skip_flag = None
        proved = bool(reuse and colors_are_depths)      # the host knows: no flag, no conditional launches, no copies
        if reuse and not proved:
            skip_flag = torch.empty(1, dtype=torch.int32, device=dev)
            L.check(lib.sgn_colors_match_depths(num_points, L.ptr(colors_c), L.ptr(depths_c), L.ptr(skip_flag),
                                                stream_ptr), "sgn_colors_match_depths")
        out_depth = torch.empty(img_height, img_width, **f32) if accumulate else None
        # ONE library call for the whole node where nothing special is asked (sgn_rasterize_fwd_all, round 5): the full scene,
        # a fresh binning, no depth channel / reuse / groups, a capacity known from earlier calls
        comp = None
        if (wcomp is None and composite_forward and tile_order_enabled and plain and not hit and not reuse
                and group_split is None and _bin_pending["key"] != key and not S.pending_checks):
            _drop_pending()        # a prefetched binning of OTHER tensors will never be finished
            comp = _forward_composite(S, key, _t, cull, num_points, xys_c, depths, radii, conics_c, colors_c, opac_c,
                                      opacity_is_logit, img_height, img_width, block_width, tile_bounds, bg_c, out_img,
                                      final_Ts, final_idx, ro, ro_ptr, opacity_logits, stream_ptr,
                                      out_depth if accumulate else None)
        if wcomp is not None:
            pass                               # everything was queued by the one call above
        elif comp is not None:
            num_intersects, gaussian_ids_sorted, tile_bins, order, tile_kmax, recs = comp
            if num_intersects < 1:
                recs = None
                out_depth = None               # never written (no forward ran): the want_depth branch below returns zeros
                out_img = torch.ones(img_height, img_width, 3, **f32) * bg_c
                final_Ts = torch.ones(img_height, img_width, **f32)
                final_idx = torch.zeros(img_height, img_width, dtype=torch.int32, device=dev)
                gaussian_ids_sorted = torch.zeros(0, dtype=torch.int32, device=dev)
            if accumulate and num_intersects >= 1:      # (the composite serves plain fresh passes only: cacheable)
                depth_stats["accumulated"] += 1
                S.depth_caches[key] = dict(key=key, D=out_depth, T=final_Ts, idx=final_idx, kmax=tile_kmax)
                while len(S.depth_caches) > _State.BIN_ENTRIES:
                    S.depth_caches.popitem(last=False)
                _depth_state["unused"] += 1
                if _depth_state["unused"] > 8 and depth_channel == "auto":   # the depth passes stopped coming
                    _depth_state["want"], _depth_state["unused"] = False, 0
            else:
                S.depth_caches.pop(key, None)  # binned again without the channel: a stale image must not answer later
        else:
            # ... and the per-Gaussian rows are built between the binning's first half and its host sync, so the GPU has
            # work queued while the host wakes up
            recs, rows_built = None, 0
            if num_points > 0 and not proved:
                if win is None and not hit and _bin_pending["key"] != key:
                    _drop_pending()
                    _bin_pending["state"] = _bin_prepare_async(num_points, xys, depths, radii, num_tiles_hit, tile_bounds,
                                                               block_width, conics, opacity, opacity_is_logit, cull)
                    _bin_pending["key"], _bin_pending["keep"] = key, tuple(t.detach() for t in _t)
                recs = L.workspace(lib.sgn_raster_workspace_bytes(n_full, 0, ro_ptr), dev)
                L.check(lib.sgn_raster_build_rows(n_full, L.ptr(xys_c), L.ptr(conics_c), L.ptr(colors_c), L.ptr(opac_c),
                                                  int(bool(opacity_is_logit)), id_lo, id_hi, window, L.ptr(recs),
                                                  recs.numel(), L.ptr(skip_flag), stream_ptr), "sgn_raster_build_rows")
                rows_built = 1
            if win is not None:
                num_intersects, gaussian_ids_sorted, tile_bins = cached
            else:
                num_intersects, gaussian_ids_sorted, tile_bins = _bin_gaussians_cached(
                    num_points, xys, depths, radii, num_tiles_hit, tile_bounds, block_width, conics, opacity,
                    opacity_is_logit, pre=(key, _t, cull), logit_leaves=opacity_logits)
            ro.ids_qmask = int(bool(getattr(gaussian_ids_sorted, "_sgn_qmask", False)))   # the backward runs with `ro` too
            if (num_intersects >= 1 and list_window_enabled and (id_hi - id_lo) < list_window_max_frac * n_full
                    and (id_lo, id_hi) != (0, n_full)):
                # a SMALL window of a shared list (the scene graph's objects-only pass): walk its own entries only
                gaussian_ids_sorted, tile_bins = _list_window(gaussian_ids_sorted, tile_bins, id_lo, id_hi, ro.ids_qmask)
                window_stats["sub_lists"] += 1
            if proved and num_intersects >= 1:
                # the depth pass, proven on the host: its image comes from the first pass's fourth channel, and its node
                # SHARES that pass's per-pixel state and tile statistics (same geometry and opacities: same values)
                final_Ts, final_idx, tile_kmax = dcache["T"], dcache["idx"], dcache["kmax"]
                order = _tile_order(tile_bins, None, _fwd_long_thresh(ro, block_width))
                L.check(lib.sgn_depth_reuse(img_height, img_width, None, L.ptr(dcache["D"]), L.ptr(final_Ts),
                                            L.ptr(final_idx), L.ptr(bg_c), L.ptr(out_img), None, None, 0, None, None,
                                            stream_ptr), "sgn_depth_reuse")
                depth_stats["reused"] += 1
                depth_stats["proved_on_host"] += 1
                _depth_state["unused"] = 0
            elif num_intersects < 1:
                recs = None
                out_depth = None            # never written (no forward ran): the want_depth branch below returns zeros
                out_img = torch.ones(img_height, img_width, 3, **f32) * bg_c
                gaussian_ids_sorted = torch.zeros(0, dtype=torch.int32, device=dev)
                tile_bins = torch.zeros(tile_bounds[0] * tile_bounds[1], 2, dtype=torch.int32, device=dev)
                final_Ts = torch.ones(img_height, img_width, **f32)
                final_idx = torch.zeros(img_height, img_width, dtype=torch.int32, device=dev)
            elif group_split is not None:
                # the main pass with the two group accumulations riding on it (sgn_raster_fwd_groups): head = ids below the
                # split, tail = the others.  The smaller group, if small enough, gets its own compacted list: its backward
                # walks that list, so its indices are recorded in that list's positions (and its forward walk finishes
                # there); the other one walks the shared list (with the first group's rows inert).
                assert win is None and id_range is None and block_width == 16
                if not rows_built:
                    recs = L.workspace(lib.sgn_raster_workspace_bytes(n_full, num_intersects, ro_ptr), dev)
                order = _tile_order(tile_bins, None, _fwd_long_thresh(ro, block_width))
                n_tiles = tile_bins.shape[0]
                tile_kmax = torch.empty(n_tiles, 2, dtype=torch.int32, device=dev)
                split = min(max(int(group_split), 0), n_full)
                sizes = (split, n_full - split)
                small = 0 if sizes[0] <= sizes[1] else 1
                own = small if (list_window_enabled and 0 < sizes[small] < list_window_max_frac * n_full) else -1
                state = torch.empty(4, img_height, img_width, **f32)          # T_head, T_tail, idx_head, idx_tail
                kmax = torch.empty(2, n_tiles, 2, dtype=torch.int32, device=dev)
                groups = []
                for gi, (lo, hi) in enumerate(((0, split), (split, n_full))):
                    g_ids, g_bins = (_list_window(gaussian_ids_sorted, tile_bins, lo, hi, ro.ids_qmask) if gi == own
                                     else (gaussian_ids_sorted, tile_bins))
                    window_stats["sub_lists"] += int(gi == own)
                    groups.append(dict(lo=lo, hi=hi, own=gi == own, ids=g_ids, bins=g_bins, T=state[gi],
                                       idx=state[2 + gi].view(torch.int32), kmax=kmax[gi]))
                og = groups[own] if own >= 0 else None
                L.check(lib.sgn_raster_fwd_groups(
                    img_height, img_width, n_full, num_intersects, L.ptr(gaussian_ids_sorted), L.ptr(tile_bins),
                    L.ptr(xys_c), L.ptr(conics_c), L.ptr(colors_c), L.ptr(opac_c), int(bool(opacity_is_logit)), L.ptr(bg_c),
                    L.ptr(out_img), L.ptr(final_Ts), L.ptr(final_idx), L.ptr(recs), recs.numel(), rows_built, L.ptr(order),
                    L.ptr(tile_kmax), L.ptr(depths_c) if accumulate else None, L.ptr(out_depth), split, own,
                    L.ptr(og["ids"]) if og else None, L.ptr(og["bins"]) if og else None, L.ptr(state), L.ptr(kmax),
                    ro_ptr, stream_ptr), "sgn_raster_fwd_groups")
                group_stats["passes"] += 1
                ctx.groups = groups
                if accumulate:
                    depth_stats["accumulated"] += 1
                if depth_cacheable:
                    S.depth_caches[key] = dict(key=key, D=out_depth, T=final_Ts, idx=final_idx, kmax=tile_kmax)
                    while len(S.depth_caches) > _State.BIN_ENTRIES:
                        S.depth_caches.popitem(last=False)
                elif not hit:
                    S.depth_caches.pop(key, None)
            else:
                if not rows_built:
                    recs = L.workspace(lib.sgn_raster_workspace_bytes(n_full, num_intersects, ro_ptr), dev)
                # packed forward: the leading tiles of the order whose lists reach adapt_fwd entries get four waves
                order = _tile_order(tile_bins, None, _fwd_long_thresh(ro, block_width))
                tile_kmax = torch.empty(tile_bins.shape[0], 2, dtype=torch.int32, device=dev)   # walk depth, pairs
                L.check(lib.sgn_raster_fwd(
                    img_height, img_width, block_width, n_full, num_intersects, L.ptr(gaussian_ids_sorted), L.ptr(tile_bins),
                    L.ptr(xys_c), L.ptr(conics_c), L.ptr(colors_c), L.ptr(opac_c), int(bool(opacity_is_logit)), id_lo, id_hi,
                    window, L.ptr(bg_c), L.ptr(out_img), L.ptr(final_Ts), L.ptr(final_idx), L.ptr(recs), recs.numel(),
                    rows_built, L.ptr(order), L.ptr(tile_kmax), L.ptr(depths_c) if accumulate else None,
                    L.ptr(out_depth), L.ptr(skip_flag), ro_ptr, stream_ptr), "sgn_raster_fwd")
                if reuse:
                    L.check(lib.sgn_depth_reuse(img_height, img_width, L.ptr(skip_flag), L.ptr(dcache["D"]),
                                                L.ptr(dcache["T"]), L.ptr(dcache["idx"]), L.ptr(bg_c), L.ptr(out_img),
                                                L.ptr(final_Ts), L.ptr(final_idx), tile_kmax.numel(),
                                                L.ptr(dcache["kmax"]), L.ptr(tile_kmax), stream_ptr), "sgn_depth_reuse")
                    recs = None                    # the rows were not built if the flag said "reuse": the backward packs them
                    depth_stats["reused"] += 1
                    _depth_state["unused"] = 0
                elif accumulate and not depth_cacheable:
                    depth_stats["accumulated"] += 1            # an explicit request on a cached list / an id range: not kept
                elif accumulate:
                    depth_stats["accumulated"] += 1
                    S.depth_caches[key] = dict(key=key, D=out_depth, T=final_Ts, idx=final_idx, kmax=tile_kmax)
                    while len(S.depth_caches) > _State.BIN_ENTRIES:
                        S.depth_caches.popitem(last=False)
                    _depth_state["unused"] += 1
                    if _depth_state["unused"] > 8 and depth_channel == "auto":   # the depth passes stopped coming
                        _depth_state["want"], _depth_state["unused"] = False, 0
                elif not hit:
                    S.depth_caches.pop(key, None)  # re-binned without the channel: a stale image must not answer later
        # (the sink hears of passes that WILL have a backward only: a forward under no_grad — an evaluation image between
        # two training steps — announces nothing; such an announcement used to stay behind and count as a second view of
        # the next step whenever its list happened to land on another address)
        sink = _touch_sink if (getattr(_call_state, "grad", True) and any(ctx.needs_input_grad)) else None
        if sink is not None and num_intersects >= 1 and win is None and id_range is None and not proved:
            # data-parallel row exchange: which Gaussians this view's backward can touch (the walked entries)
            sink.after_forward(gaussian_ids_sorted, tile_bins, tile_kmax, n_full, ro.ids_qmask)
        if sink is not None and num_intersects >= 1 and (win is not None or id_range is not None
                                                           or group_split is not None):
            # a sub-model pass / group accumulation walks entries the full pass's list of walked rows does not cover
            # (its transmittance falls more slowly): the sink must not take that list for the step's touched rows
            extra = getattr(sink, "extra_pass", None)
            if extra is not None:
                extra()
        ctx.img_width, ctx.img_height, ctx.block_width = img_width, img_height, block_width
        ctx.num_intersects = num_intersects
        ctx.opacity_is_logit = int(bool(opacity_is_logit))
        ctx.id_range = (id_lo, id_hi)
        ctx.window, ctx.n_full, ctx.ro = window, n_full, ro
        ctx.tile_kmax = tile_kmax if num_intersects >= 1 else None
        ctx.opacity_shape = opacity.shape
        ctx.recs = recs
        ctx.save_for_backward(gaussian_ids_sorted, tile_bins, xys_c, conics_c, colors_c, opac_c, bg_c,
                              final_Ts, final_idx, *([_f32c(colors_pre)] if colors_pre is not None else []))
        if group_split is not None:
            if want_depth and out_depth is None:
                out_depth = torch.zeros(img_height, img_width, **f32)
            if out_depth is not None:
                ctx.mark_non_differentiable(out_depth)
            gs = getattr(ctx, "groups", None)
            accs = [(1 - g["T"]) if gs is not None else torch.zeros(img_height, img_width, **f32)
                    for g in (gs or (None, None))]
            return out_img, 1 - final_Ts, out_depth, accs[0], accs[1]
        if want_depth:
            if out_depth is None:                 # nothing visible / a path without the channel: the two-pass image
                out_depth = torch.zeros(img_height, img_width, **f32)
            ctx.mark_non_differentiable(out_depth)
            out_alpha = 1 - final_Ts
            return out_img, out_alpha, out_depth
        if return_alpha:
            out_alpha = 1 - final_Ts
            return out_img, out_alpha
        return out_img

    @staticmethod
    def backward(ctx, v_out_img, v_out_alpha=None, _v_depth=None, v_acc_head=None, v_acc_tail=None):
        (gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity, background, final_Ts,
         final_idx) = ctx.saved_tensors[:9]
        colors_pre = ctx.saved_tensors[9] if ctx.grad_to_pre else None
        # a hook / retain_grad placed on the bypassed sigmoid(...) / clamp(...) tensor AFTER the call (see `_asked_for`):
        # that activation is then differentiated by autograd, from the plain gradient this node computes for it
        by = getattr(ctx, "bypassed", ()) or (None, None)
        ask_o = _asked_for(by[0]) if ctx.grad_to_logits else None
        ask_c = _asked_for(by[1]) if ctx.grad_to_pre else None
        logit_mode = ctx.opacity_is_logit if (ask_o is not None or not ctx.grad_to_logits) else 2
        if ask_c is not None:
            colors_pre = None                # plain colour gradient (the clamp's mask is autograd's business then)
        dev = xys.device
        n = xys.shape[0]                     # rows of the caller's tensors (= the window's rows in window mode)
        H, W = ctx.img_height, ctx.img_width
        f32 = dict(dtype=torch.float32, device=dev)
        groups = getattr(ctx, "groups", None)
        # the two group accumulations (sgn_raster_fwd_groups): each one that reached the loss is one more reverse walk
        # — of the group's own list or of the shared one — with the state its forward recorded; alpha only
        group_work = [(g, v) for g, v in zip(groups or (), (v_acc_head, v_acc_tail)) if v is not None and g["hi"] > g["lo"]]
        main = not (group_work and v_out_img is None and v_out_alpha is None)     # (nothing but group outputs in the loss)
        if v_out_alpha is None:
            v_out_alpha = torch.zeros(H, W, **f32)
        # only the alpha output reached the loss (the scene graph's accumulation passes, scene_graph.py:364-366): the
        # colour gradient is exactly zero — hand autograd None instead, and the clamp / SH / concatenation backward
        # behind the colours (a dense [n,K,3] gradient and its split over the leaves) is skipped altogether
        no_color_grad = v_out_img is None                   # (the kernel takes NULL for "zeros": no fill, no read)
        v_out_img = None if v_out_img is None else _f32c(v_out_img)
        v_out_alpha = _f32c(v_out_alpha)

        def one_pass(ids, bins, Ts, idx, kmax, v_img, v_alpha, id_range, window, recs, pre, out=None, part=None):
            # `out` / `part`: one walk of a sequence that accumulates into one gradient workspace (sgn_raster_bwd_part):
            # out = (v_xy, v_conic, v_colors, v_opacity, workspace) shared by the sequence, part = (first, last)
            lib = L.load()
            if out is None:
                out = (torch.empty(n, 2, **f32), torch.empty(n, 3, **f32), torch.empty(n, 3, **f32),
                       _leaf_grad(getattr(ctx, "arena_leaves", None), 0, (n,), f32),
                       L.workspace(lib.sgn_raster_bwd_workspace_bytes(ctx.n_full), dev))
            v_xy, v_conic, v_colors, v_opacity, gws = out
            ro_ptr = C.byref(ctx.ro)
            packed = 1
            if recs is None:
                recs = L.workspace(lib.sgn_raster_workspace_bytes(ctx.n_full, ctx.num_intersects, ro_ptr), dev)
                packed = 0
            # the backward's own launch order, by REVERSE-WALK length (tried in r03: the forward's order by list length
            # plus per-tile classification, no launch — the one-wave kernel then ran 323 instead of 288 us on the
            # benchmark scene: a late long walk is a lone-wave tail; profiles/r03f_*)
            # (a group walk's statistics hold the walk depth only — sgn_raster_fwd_groups records no per-tile pair count
            # for the head / tail groups — so the small-splat classification, which reads that count, stays off for it)
            pairs_known = kmax is ctx.tile_kmax
            if composite_backward:
                # ONE library call: launch order + reverse walks + unpack (sgn_rasterize_bwd_all, round 6)
                S = _S()
                n_tiles = bins.shape[0]
                order = torch.empty(n_tiles + 2, dtype=torch.int32, device=dev) if tile_order_enabled else None
                scratch = _order_scratch(S, lib, n_tiles, dev) if order is not None else None
                if order is not None and not window and id_range == (0, n):
                    S.walk_stat = order[-1:]         # rides to the host with the next binning's count (mask policy)
                first, last = (1, 1) if part is None else (int(part[0]), int(part[1]))
                L.check(lib.sgn_rasterize_bwd_all(
                    H, W, ctx.block_width, ctx.n_full, ctx.num_intersects, L.ptr(ids), L.ptr(bins), L.ptr(kmax),
                    int(pairs_known), L.ptr(xys), L.ptr(conics), L.ptr(colors), L.ptr(opacity),
                    logit_mode, id_range[0], id_range[1], window,
                    L.ptr(background), L.ptr(Ts), L.ptr(idx), L.ptr(v_img), L.ptr(v_alpha), ctx.alpha_clamp_bwd,
                    L.ptr(v_xy), L.ptr(v_conic), L.ptr(v_colors), L.ptr(v_opacity), L.ptr(recs), recs.numel(), packed,
                    L.ptr(gws), gws.numel(), L.ptr(order), L.ptr(scratch),
                    4 * scratch.numel() if scratch is not None else 0, int(small_splat_q16), L.ptr(pre), ro_ptr,
                    L.stream_ptr(), L.aux_stream_ptr(dev) if concurrent_backward else None, first, last),
                    "sgn_rasterize_bwd_all")
                composite_stats["backwards"] += 1
                return out
            order = _tile_order(bins, kmax, ctx.ro.adapt_bwd, pairs_known=pairs_known)
            if order is not None and not window and id_range == (0, n):
                _S().walk_stat = order[-1:]          # rides to the host with the next binning's count (mask policy)
            args = (H, W, ctx.block_width, ctx.n_full, ctx.num_intersects, L.ptr(ids), L.ptr(bins),
                    L.ptr(xys), L.ptr(conics), L.ptr(colors), L.ptr(opacity),
                    logit_mode, id_range[0],
                    id_range[1], window, L.ptr(background), L.ptr(Ts),
                    L.ptr(idx), L.ptr(v_img), L.ptr(v_alpha), ctx.alpha_clamp_bwd, L.ptr(v_xy),
                    L.ptr(v_conic), L.ptr(v_colors), L.ptr(v_opacity), L.ptr(recs), recs.numel(), packed,
                    L.ptr(gws), gws.numel(), L.ptr(order), L.ptr(pre), ro_ptr, L.stream_ptr(),
                    L.aux_stream_ptr(dev) if concurrent_backward else None)
            if part is None:
                L.check(lib.sgn_raster_bwd(*args), "sgn_raster_bwd")
            else:
                L.check(lib.sgn_raster_bwd_part(*args, int(part[0]), int(part[1])), "sgn_raster_bwd_part")
            return out

        if ctx.num_intersects < 1 or not (main or group_work):
            v_xy, v_conic = torch.zeros(n, 2, **f32), torch.zeros(n, 3, **f32)
            v_colors, v_opacity = torch.zeros(n, 3, **f32), torch.zeros(n, **f32)
        else:
            # the walks of this node: the main pass, then every group accumulation in the loss — its own list holds the
            # group's entries only, so the main pass's rows serve; on the shared list the other group's rows must be
            # inert, so the rows are built again for the id range (recs = None).  More than one walk: all accumulate
            # into one gradient workspace and the last one unpacks (sgn_raster_bwd_part).
            walks = []
            if main:
                walks.append((gaussian_ids_sorted, tile_bins, final_Ts, final_idx, ctx.tile_kmax, v_out_img, v_out_alpha,
                              ctx.id_range, ctx.window, ctx.recs, colors_pre))
            else:
                no_color_grad = True
            for g, v in group_work:
                group_stats["backward_passes"] += 1
                walks.append((g["ids"], g["bins"], g["T"], g["idx"], g["kmax"], None, _f32c(v), (g["lo"], g["hi"]), 0,
                              ctx.recs if g["own"] else None, colors_pre if main else None))
            out = None
            for wi, w in enumerate(walks):
                out = one_pass(*w, out=out, part=None if len(walks) == 1 else (wi == 0, wi == len(walks) - 1))
            v_xy, v_conic, v_colors, v_opacity = out[:4]
        v_opacity = v_opacity.reshape(ctx.opacity_shape)
        if no_color_grad:
            v_colors = None
        v_logits = ()
        if ctx.grad_to_logits:
            v_logits = (v_opacity,) if len(ctx.logit_rows) == 1 else v_opacity.split(ctx.logit_rows)
        if ask_o is not None or ask_c is not None:
            through, grads = [], []
            if ask_o is not None:
                through.append(ask_o); grads.append(v_opacity.reshape(ask_o.shape))
                v_logits = (None,) * len(ctx.logit_rows)
                hooks_after_call_stats["opacity"] += 1
            if ask_c is not None and v_colors is not None:
                through.append(ask_c); grads.append(v_colors.reshape(ask_c.shape))
                v_colors = None
                hooks_after_call_stats["colors"] += 1
            if through:
                torch.autograd.backward(through, grads, retain_graph=True)
        # (xys, depths, radii, conics, num_tiles_hit, colors, opacity, H, W, block, background, return_alpha,
        #  opacity_is_logit, id_range, want_depth, colors_are_depths, colors_pre, group_split, *opacity_logits)
        return (v_xy, None, None, v_conic, None, None if ctx.grad_to_pre else v_colors,
                None if ctx.grad_to_logits else v_opacity) + (None,) * 9 + (
            v_colors if ctx.grad_to_pre else None, None) + tuple(v_logits)


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height: int,
                        img_width: int, block_width: int, background: Optional[torch.Tensor] = None,
                        return_alpha: Optional[bool] = False):
    """gsplat/rasterize.py rasterize_gaussians (sgn_splatfacto.py:954-967, :982-994).

    Returns ``out_img [H,W,3]`` or ``(out_img, out_alpha [H,W])``."""
    assert block_width > 1 and block_width <= 16, "block_width must be between 2 and 16"
    if colors.dtype == torch.uint8:
        colors = colors.float() / 255  # make sure colors are float [0,1]
    if background is not None:
        assert background.shape[0] == colors.shape[-1], \
            f"incorrect shape of background color tensor, expected shape {colors.shape[-1]}"
    else:
        background = torch.ones(colors.shape[-1], dtype=torch.float32, device=colors.device)
    if xys.ndimension() != 2 or xys.size(1) != 2:
        raise ValueError("xys must have dimensions (N, 2)")
    if colors.ndimension() != 2:
        raise ValueError("colors must have dimensions (N, D)")
    if colors.shape[-1] != 3:
        # upstream's N-D path (`_RasterizeGaussians` dispatches D != 3 to nd_rasterize_forward).  The reference never
        # takes it (sgn_splatfacto.py:988 repeats depth x3 to stay on three channels); served here by the 3-channel
        # kernels, three channels per pass: the passes share ONE binning (the cache is keyed on the geometry tensors),
        # the per-pixel walk is the same for every channel, so each output channel equals what an N-D kernel composites.
        d = colors.shape[-1]
        if d < 1:
            raise ValueError("colors must have at least one channel")
        imgs, alpha = [], None
        for c0 in range(0, d, 3):
            w = min(3, d - c0)
            chunk, bg3 = colors[:, c0:c0 + w], background[c0:c0 + w]
            if w < 3:
                chunk = torch.cat([chunk, chunk.new_zeros(chunk.shape[0], 3 - w)], dim=1)
                bg3 = torch.cat([bg3, bg3.new_zeros(3 - w)])
            img, alpha = rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, chunk, opacity, img_height,
                                             img_width, block_width, bg3, True)
            imgs.append(img[..., :w])
        out_img = torch.cat(imgs, dim=-1)
        return (out_img, alpha) if return_alpha else out_img
    logits, pre = (), None
    if _proofs_on(activation_proofs) and opacity.is_cuda and colors.shape[-1] == 3:
        logits, pre = proofs.sigmoid_leaves(opacity) or (), proofs.clamp_pre(colors)
        activation_proof_stats["opacity"] += len(logits) > 0
        activation_proof_stats["colors"] += pre is not None
    _call_state.grad = torch.is_grad_enabled()
    _call_state.bypassed = (opacity if logits else None, colors if pre is not None else None)
    c = _contig
    return _RasterizeGaussians.apply(c(xys), c(depths), c(radii), c(conics), c(num_tiles_hit),
                                     c(colors.detach() if pre is not None else colors),
                                     c(opacity.detach() if logits else opacity), img_height,
                                     img_width, block_width, c(background), return_alpha, False, None, False,
                                     depth_channel != "off" and proofs.enabled() and _provably_depths(colors, depths),
                                     pre, None, *logits)


# -------------------------------------------------------------- _torch_impl
def quat_to_rotmat(quat: torch.Tensor) -> torch.Tensor:
    """gsplat/_torch_impl.py quat_to_rotmat (sgn_splatfacto.py:685; densification only, cold):
    normalises, (w,x,y,z) -> [..., 3, 3].  Plain torch on whatever device the input lives on."""
    assert quat.shape[-1] == 4, quat.shape
    w, x, y, z = torch.unbind(torch.nn.functional.normalize(quat, dim=-1), dim=-1)
    mat = torch.stack(
        [
            1 - 2 * (y**2 + z**2), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x**2 + z**2), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x**2 + y**2),
        ],
        dim=-1,
    )
    return mat.reshape(quat.shape[:-1] + (3, 3))
