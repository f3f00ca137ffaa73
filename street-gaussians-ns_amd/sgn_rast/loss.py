"""Photometric loss of the reference's training step on MI355X: L1 + SSIM, one kernel each way.

SURVEY.md §8f row 3.  The reference computes (``street_gaussians_ns/sgn_splatfacto.py:1084-1087``)

    Ll1 = torch.abs(gt_img - rgb).mean()
    simloss = 1 - self.ssim(gt_img.permute(2, 0, 1)[None, ...], rgb.permute(2, 0, 1)[None, ...])
    loss = (1 - ssim_lambda) * Ll1 + ssim_lambda * simloss

with ``self.ssim = pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)`` (``:330``): ten depthwise
``conv2d`` launches forward plus their autograd graph.  Here both terms come out of ``csrc/loss.hip`` through the C
ABI (``sgn_l1_ssim_fwd/bwd``) directly on the rasterizer's HWC image; fails loudly without the HIP library.

* :func:`l1_ssim` — ``(Ll1, ssim)`` of two [H,W,3] images, gradient to ``pred``.
* :class:`SSIM` — ``pytorch_msssim.SSIM`` call shape (``forward(X, Y)`` on [1,3,H,W]) for the import shim; the
  permuted views the reference passes are recognised and used in place (no NCHW copy).
* :func:`sky_accumulation`, :func:`object_acc_entropy`, :func:`accumulation_losses` — the two accumulation
  regularisers of the reference's loss dictionary (``sgn_splatfacto.py:1090-1093``,
  ``sgn_splatfacto_scene_graph.py:386-389``): ``(sky_mask * accumulation).mean()`` and the binary entropy of the clamped
  object accumulation, one streaming pass each way for both (``sgn_acc_losses_fwd/bwd``) instead of ~10 elementwise
  torch launches forward and as many backward.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _forward(ctx, pred, gt, data_range, clamp_max, ssim_lambda):
    L.require_device(pred, gt)
    if pred.dim() != 3 or pred.shape[-1] != 3 or pred.shape != gt.shape:
        raise ValueError(f"l1_ssim expects two [H,W,3] images, got {tuple(pred.shape)} and {tuple(gt.shape)}")
    h, w = pred.shape[0], pred.shape[1]
    if min(h, w) <= 10:
        raise ValueError("images must be larger than the 11-tap SSIM window")   # pytorch_msssim asserts too
    p, g = pred.contiguous().float(), gt.contiguous().float()
    lib = L.load()
    out3 = torch.empty(3, dtype=torch.float32, device=p.device)
    need_grad = int(bool(ctx.needs_input_grad[0]))
    maps = L.workspace(lib.sgn_l1_ssim_workspace_bytes(h, w, need_grad), p.device)
    cmax = float("inf") if clamp_max is None else float(clamp_max)
    L.check(lib.sgn_l1_ssim_fwd(h, w, L.ptr(p), L.ptr(g), float(data_range), cmax, float(ssim_lambda), L.ptr(out3),
                                need_grad, L.ptr(maps), maps.numel(), L.stream_ptr()), "sgn_l1_ssim_fwd")
    ctx.hw, ctx.cmax, ctx.maps = (h, w), cmax, maps
    ctx.save_for_backward(p, g)
    return out3


def _backward(ctx, gscale):
    p, g = ctx.saved_tensors
    h, w = ctx.hw
    v = torch.empty_like(p)
    L.check(L.load().sgn_l1_ssim_bwd(h, w, L.ptr(p), L.ptr(g), ctx.cmax, L.ptr(ctx.maps), L.ptr(gscale), L.ptr(v),
                                     L.stream_ptr()), "sgn_l1_ssim_bwd")
    return v


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, data_range, clamp_max):
        out3 = _forward(ctx, pred, gt, data_range, clamp_max, 0.0)
        return out3[0], out3[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        gscale = torch.stack([g_l1.reshape(()), g_ssim.reshape(())]).float().contiguous()
        return _backward(ctx, gscale), None, None, None


_LAMBDA_VEC: dict = {}


class _Photometric(torch.autograd.Function):
    """(1 - l) Ll1 + l (1 - ssim) as ONE node: the weighted sum is formed in the reduction kernel, the backward
    turns the upstream scalar into the two weights with a single tiny multiply."""

    @staticmethod
    def forward(ctx, pred, gt, ssim_lambda, clamp_max):
        out3 = _forward(ctx, pred, gt, 1.0, clamp_max, ssim_lambda)
        ctx.lam = float(ssim_lambda)
        return out3[2]

    @staticmethod
    def backward(ctx, g):
        key = (ctx.lam, str(g.device))
        if key not in _LAMBDA_VEC:
            _LAMBDA_VEC[key] = torch.tensor([1.0 - ctx.lam, -ctx.lam], dtype=torch.float32, device=g.device)
        gscale = (_LAMBDA_VEC[key] * g.reshape(())).contiguous()
        return _backward(ctx, gscale), None, None, None


def l1_ssim(pred: torch.Tensor, gt: torch.Tensor, data_range: float = 1.0, clamp_max=None):
    """(mean |gt - pred|, SSIM(gt, pred)) for [H,W,3] images; differentiable w.r.t. ``pred``.
    ``clamp_max`` folds the caller's ``torch.clamp(rgb, max=clamp_max)`` (``sgn_splatfacto.py:969``) into the kernels:
    ``pred`` is read as ``min(pred, clamp_max)`` and the gradient is zero where ``pred > clamp_max``."""
    return _L1SSIM.apply(pred, gt, data_range, clamp_max)


def photometric_loss(pred: torch.Tensor, gt: torch.Tensor, ssim_lambda: float = 0.2, clamp_max=None) -> torch.Tensor:
    """``(1 - l) * Ll1 + l * (1 - ssim)`` — the sum of losses["Ll1"] and losses["simloss"] (``:1086-1087``), one
    forward and one backward kernel plus a single scalar multiply."""
    return _Photometric.apply(pred, gt, ssim_lambda, clamp_max)


def _as_hwc(t: torch.Tensor) -> torch.Tensor:
    """[1,3,H,W] -> [H,W,3] without a copy when ``t`` is the permuted view of an HWC image (the reference's case)."""
    if t.dim() != 4 or t.shape[0] != 1 or t.shape[1] != 3:
        raise ValueError(f"SSIM expects [1,3,H,W] inputs, got {tuple(t.shape)}")
    return t[0].permute(1, 2, 0)


class SSIM(torch.nn.Module):
    """``pytorch_msssim.SSIM`` for the configuration the reference builds (``sgn_splatfacto.py:330``):
    ``data_range`` free, ``size_average=True``, ``channel=3``, 11-tap window, sigma 1.5, 2-D images, batch 1.
    Anything else raises.  SSIM is symmetric, so the gradient goes to whichever argument requires it."""

    def __init__(self, data_range: float = 255, size_average: bool = True, win_size: int = 11,
                 win_sigma: float = 1.5, channel: int = 3, spatial_dims: int = 2, K=(0.01, 0.03),
                 nonnegative_ssim: bool = False):
        super().__init__()
        if not size_average or win_size != 11 or win_sigma != 1.5 or channel != 3 or spatial_dims != 2 \
                or tuple(K) != (0.01, 0.03) or nonnegative_ssim:
            raise NotImplementedError("sgn_rast.loss.SSIM implements the reference's configuration only")
        self.data_range = float(data_range)

    def forward(self, X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
        x, y = _as_hwc(X), _as_hwc(Y)
        if x.requires_grad and y.requires_grad:
            # both sides differentiable: d/dX from f(X, Y.detach()), d/dY from f(Y, X.detach()) (symmetry)
            a = l1_ssim(x, y.detach(), self.data_range)[1]
            b = l1_ssim(y, x.detach(), self.data_range)[1]
            return a + b - a.detach()
        if x.requires_grad:
            return l1_ssim(x, y, self.data_range)[1]
        return l1_ssim(y, x, self.data_range)[1]


# ------------------------------------------------------------------------------------------- accumulation regularisers
SKY = 2   # street_gaussians_ns/data/utils/data_utils.py:26-29  SemanticType.SKY


def _sem_arg(semantic: torch.Tensor, n: int):
    """(contiguous integer tensor, bytes per element) the C ABI accepts: 1 (uint8 / bool), 4 (int32), 8 (int64)."""
    if semantic.numel() != n:
        raise ValueError(f"semantic has {semantic.numel()} elements, the accumulation image {n}")
    t = semantic.detach()
    if t.dtype == torch.bool:
        t = t.contiguous().view(torch.uint8)
    elif t.dtype not in (torch.uint8, torch.int32, torch.int64):
        t = t.to(torch.int64)
    t = t.contiguous()
    return t, t.element_size()


class _AccLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, accumulation, semantic, sky_value, object_acc):
        dev = L.require_device(accumulation, semantic, object_acc)
        if accumulation is None and object_acc is None:
            raise ValueError("accumulation_losses needs an accumulation image, an object accumulation image or both")
        if accumulation is not None and semantic is None:
            raise ValueError("the sky-accumulation term needs the semantic image")
        n = (accumulation if accumulation is not None else object_acc).numel()
        if n < 1 or (accumulation is not None and object_acc is not None and object_acc.numel() != n):
            raise ValueError("accumulation images must be non-empty and of one size")
        acc = None if accumulation is None else accumulation.detach().float().contiguous()
        obj = None if object_acc is None else object_acc.detach().float().contiguous()
        sem, sem_bytes = (None, 0) if accumulation is None else _sem_arg(semantic, n)
        if sem is not None and sem.dtype == torch.uint8 and semantic.dtype == torch.bool:
            sky_value = 1 if sky_value else 0      # a boolean mask: "is sky" is True
        lib = L.load()
        out2 = torch.empty(2, dtype=torch.float32, device=dev)
        ws = L.workspace(lib.sgn_acc_losses_workspace_bytes(n), dev)
        L.check(lib.sgn_acc_losses_fwd(n, L.ptr(acc), L.ptr(sem), sem_bytes, int(sky_value), L.ptr(obj), L.ptr(out2),
                                       L.ptr(ws), ws.numel(), L.stream_ptr()), "sgn_acc_losses_fwd")
        ctx.n, ctx.sem_bytes, ctx.sky_value = n, sem_bytes, int(sky_value)
        ctx.acc_shape = None if accumulation is None else accumulation.shape
        ctx.obj_shape = None if object_acc is None else object_acc.shape
        ctx.has_sem, ctx.has_obj = sem is not None, obj is not None
        ctx.save_for_backward(*[t for t in (sem, obj) if t is not None])   # in-place edits before backward are detected
        return out2[0], out2[1]

    @staticmethod
    def backward(ctx, g_sky, g_ent):
        want_acc = ctx.acc_shape is not None and ctx.needs_input_grad[0]
        want_obj = ctx.obj_shape is not None and ctx.needs_input_grad[3]
        if not (want_acc or want_obj):
            return None, None, None, None
        saved = list(ctx.saved_tensors)
        sem = saved.pop(0) if ctx.has_sem else None
        obj = saved.pop(0) if ctx.has_obj else None
        gscale = torch.stack([g_sky.reshape(()), g_ent.reshape(())]).float().contiguous()
        f32 = dict(dtype=torch.float32, device=gscale.device)
        v_acc = torch.empty(ctx.n, **f32) if want_acc else None
        v_obj = torch.empty(ctx.n, **f32) if want_obj else None
        L.check(L.load().sgn_acc_losses_bwd(ctx.n, L.ptr(sem) if want_acc else None, ctx.sem_bytes, ctx.sky_value,
                                            L.ptr(obj) if want_obj else None, L.ptr(gscale), L.ptr(v_acc),
                                            L.ptr(v_obj), L.stream_ptr()), "sgn_acc_losses_bwd")
        return (v_acc.reshape(ctx.acc_shape) if want_acc else None, None, None,
                v_obj.reshape(ctx.obj_shape) if want_obj else None)


def accumulation_losses(accumulation, semantic, object_acc, sky_value: int = SKY):
    """``((sky_mask * accumulation).mean(), -(o log o + (1 - o) log(1 - o)).mean())`` with ``sky_mask = (semantic ==
    sky_value)`` and ``o = clamp(object_acc, 1e-5, 1 - 1e-5)`` — the reference's two accumulation regularisers BEFORE
    their config multipliers (``sky_acc_loss_mult`` 0.5, ``object_acc_entropy_loss_mult`` 0.001), in one forward and
    one backward pass.  ``accumulation`` / ``object_acc``: [H,W,1] (or [H,W]) float images; ``semantic``: integer (the
    reference's int64 [H,W,1]) or boolean "is sky" image.  Either image may be ``None`` (its term is then 0)."""
    return _AccLosses.apply(accumulation, semantic, sky_value, object_acc)


def sky_accumulation(accumulation, semantic, sky_value: int = SKY) -> torch.Tensor:
    """``(sky_mask * accumulation).mean()`` (``sgn_splatfacto.py:1092-1093``)."""
    return _AccLosses.apply(accumulation, semantic, sky_value, None)[0]


def object_acc_entropy(object_acc) -> torch.Tensor:
    """``-(o log o + (1 - o) log(1 - o)).mean()``, ``o = clamp(object_acc, 1e-5, 1 - 1e-5)``
    (``sgn_splatfacto_scene_graph.py:387-389``)."""
    return _AccLosses.apply(None, None, 0, object_acc)[1]
