"""Sky cube map — the reference's `EnvLight` on MI355X.

Replaces the nvdiffrast call at ``street_gaussians_ns/sgn_splatfacto.py:145`` (``dr.texture(base[None], l,
filter_mode='linear', boundary_mode='cube')``).  Two entry points, both backed by ``csrc/cubemap.hip`` through the
C ABI (``sgn_cube_texture_*`` / ``sgn_sky_*``) and both failing loudly without the HIP library:

* :func:`texture` — drop-in for the one ``dr.texture`` mode the reference uses (same argument names);
* :class:`EnvLight` — the reference module's interface (``forward(camera, train)``), with ray generation, the
  camera rotation, the GL axis swap and the lookup fused into one kernel (no [H,W,3] direction tensor).

Gradients flow to the texture only; directions come from the camera and carry none in the reference.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L


class _CubeTexture(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, dirs):
        L.require_device(tex, dirs)
        tex_c = tex.contiguous().float()
        d = dirs.reshape(-1, 3).contiguous().float()
        n, R, C = d.shape[0], tex.shape[1], tex.shape[3]
        # the [H,W] grid of uv: the backward tiles it 16x16 so neighbouring pixels share texels in LDS
        h, w = (dirs.shape[-3], dirs.shape[-2]) if dirs.dim() >= 3 and dirs.shape[-3] * dirs.shape[-2] == n else (1, n)
        out = torch.empty(n, C, dtype=torch.float32, device=tex.device)
        L.check(L.load().sgn_cube_texture_fwd(h, w, R, C, L.ptr(tex_c), L.ptr(d), L.ptr(out), L.stream_ptr()),
                "sgn_cube_texture_fwd")
        ctx.save_for_backward(d)
        ctx.grid = (h, w)
        ctx.shape = tuple(tex.shape)
        return out.reshape(dirs.shape[:-1] + (C,))

    @staticmethod
    def backward(ctx, v_out):
        (d,) = ctx.saved_tensors
        _, R, _, C = ctx.shape
        v = v_out.reshape(-1, C).contiguous().float()
        v_tex = torch.empty(ctx.shape, dtype=torch.float32, device=v.device)
        L.check(L.load().sgn_cube_texture_bwd(ctx.grid[0], ctx.grid[1], R, C, L.ptr(d), L.ptr(v), L.ptr(v_tex),
                                              L.stream_ptr()), "sgn_cube_texture_bwd")
        return v_tex, None


def texture(tex: torch.Tensor, uv: torch.Tensor, filter_mode: str = "linear",
            boundary_mode: str = "cube") -> torch.Tensor:
    """``nvdiffrast.torch.texture`` for the mode EnvLight uses: ``tex`` [B,6,R,R,C], ``uv`` [B,H,W,3] direction
    vectors, linear filtering, cube boundary.  Anything else raises (the reference uses nothing else)."""
    if filter_mode != "linear" or boundary_mode != "cube":
        raise NotImplementedError("sgn_rast.sky.texture implements filter_mode='linear', boundary_mode='cube' only")
    if tex.dim() != 5 or tex.shape[1] != 6 or tex.shape[2] != tex.shape[3]:
        raise ValueError(f"cube texture must be [B,6,R,R,C], got {tuple(tex.shape)}")
    if uv.shape[-1] != 3 or uv.shape[0] != tex.shape[0]:
        raise ValueError(f"uv must be [B,H,W,3] with the texture's batch, got {tuple(uv.shape)}")
    return torch.stack([_CubeTexture.apply(tex[b], uv[b]) for b in range(tex.shape[0])], 0)


class _Sky(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, h, w, fx, fy, cx, cy, c2w, jitter):
        L.require_device(tex, c2w)
        tex_c = tex.contiguous().float()
        c2w = c2w.float()
        if c2w.stride(-1) != 1:
            c2w = c2w.contiguous()
        jit = None if jitter is None else jitter.contiguous().float()
        R, C = tex.shape[1], tex.shape[3]
        out = torch.empty(h, w, C, dtype=torch.float32, device=tex.device)
        L.check(L.load().sgn_sky_fwd(h, w, fx, fy, cx, cy, L.ptr(c2w), c2w.stride(0), L.ptr(jit), R, C,
                                     L.ptr(tex_c), L.ptr(out), L.stream_ptr()), "sgn_sky_fwd")
        ctx.cam = (h, w, fx, fy, cx, cy)
        ctx.save_for_backward(c2w, jit if jit is not None else torch.empty(0, device=tex.device))
        ctx.shape = tuple(tex.shape)
        return out

    @staticmethod
    def backward(ctx, v_out):
        c2w, jit = ctx.saved_tensors
        jit = jit if jit.numel() else None
        h, w, fx, fy, cx, cy = ctx.cam
        _, R, _, C = ctx.shape
        v = v_out.contiguous().float()
        v_tex = torch.empty(ctx.shape, dtype=torch.float32, device=v.device)
        L.check(L.load().sgn_sky_bwd(h, w, fx, fy, cx, cy, L.ptr(c2w), c2w.stride(0), L.ptr(jit), R, C, L.ptr(v),
                                     L.ptr(v_tex), L.stream_ptr()), "sgn_sky_bwd")
        return (v_tex,) + (None,) * 8


def sky_color(base: torch.Tensor, h: int, w: int, fx: float, fy: float, cx: float, cy: float, c2w: torch.Tensor,
              jitter: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[h,w,C] sky colour for a pinhole camera; ``c2w`` is camera_to_worlds[0] ([3,4] or [4,4], on the device)."""
    return _Sky.apply(base, int(h), int(w), float(fx), float(fy), float(cx), float(cy), c2w, jitter)


class EnvLight(torch.nn.Module):
    """Same parameter (``base`` [6,R,R,3], init 0.5) and call shape as the reference's EnvLight
    (sgn_splatfacto.py:109-150); ``camera`` needs width/height/cx/cy/fx/fy/camera_to_worlds like nerfstudio's
    ``Cameras`` (tensors or numbers)."""

    def __init__(self, resolution: int = 1024, channels: int = 3):
        super().__init__()
        self.base = torch.nn.Parameter(0.5 * torch.ones(6, resolution, resolution, channels))

    @staticmethod
    def _num(v) -> float:
        return float(v.item()) if torch.is_tensor(v) else float(v)

    def forward(self, camera, train: bool = False) -> torch.Tensor:
        W, H = int(self._num(camera.width)), int(self._num(camera.height))
        c2w = camera.camera_to_worlds
        c2w = c2w[0] if c2w.dim() == 3 else c2w
        jitter = torch.rand(2, H, W, device=self.base.device) if train else None
        return sky_color(self.base, H, W, self._num(camera.fx), self._num(camera.fy), self._num(camera.cx),
                         self._num(camera.cy), c2w.to(self.base.device), jitter)


class _SkyBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, rgb, alpha, h, w, fx, fy, cx, cy, c2w, jitter):
        L.require_device(tex, rgb, alpha, c2w)
        if tex.shape[3] != 3:
            raise ValueError("sky_blend needs a 3-channel cube map")
        tex_c, rgb_c, a_c = tex.contiguous().float(), rgb.contiguous().float(), alpha.contiguous().float()
        c2w = c2w.float()
        if c2w.stride(-1) != 1:
            c2w = c2w.contiguous()
        jit = None if jitter is None else jitter.contiguous().float()
        out = torch.empty(h, w, 3, dtype=torch.float32, device=tex.device)
        sky = torch.empty(h, w, 3, dtype=torch.float32, device=tex.device)
        L.check(L.load().sgn_sky_blend_fwd(h, w, fx, fy, cx, cy, L.ptr(c2w), c2w.stride(0), L.ptr(jit), tex.shape[1],
                                           L.ptr(tex_c), L.ptr(rgb_c), L.ptr(a_c), L.ptr(out), L.ptr(sky),
                                           L.stream_ptr()), "sgn_sky_blend_fwd")
        ctx.cam = (h, w, fx, fy, cx, cy)
        ctx.save_for_backward(tex_c, rgb_c, a_c, c2w, jit if jit is not None else torch.empty(0, device=tex.device))
        ctx.mark_non_differentiable(sky)
        return out, sky

    @staticmethod
    def backward(ctx, v_out, _v_sky):
        tex, rgb, alpha, c2w, jit = ctx.saved_tensors
        jit = jit if jit.numel() else None
        h, w, fx, fy, cx, cy = ctx.cam
        v = v_out.contiguous().float()
        v_tex, v_rgb, v_alpha = torch.empty_like(tex), torch.empty_like(rgb), torch.empty_like(alpha)
        L.check(L.load().sgn_sky_blend_bwd(h, w, fx, fy, cx, cy, L.ptr(c2w), c2w.stride(0), L.ptr(jit), tex.shape[1],
                                           L.ptr(tex), L.ptr(rgb), L.ptr(alpha), L.ptr(v), L.ptr(v_rgb),
                                           L.ptr(v_alpha), L.ptr(v_tex), L.stream_ptr()), "sgn_sky_blend_bwd")
        return (v_tex, v_rgb, v_alpha) + (None,) * 8


def sky_blend(base: torch.Tensor, rgb: torch.Tensor, alpha: torch.Tensor, fx: float, fy: float, cx: float,
              cy: float, c2w: torch.Tensor, jitter: Optional[torch.Tensor] = None):
    """``rgb.clamp(max=1) * alpha + sky * (1 - alpha)`` (sgn_splatfacto.py:969-972) with the sky lookup fused in;
    ``rgb`` [H,W,3], ``alpha`` [H,W] or [H,W,1].  Returns (composite [H,W,3], sky [H,W,3] detached)."""
    h, w = rgb.shape[0], rgb.shape[1]
    return _SkyBlend.apply(base, rgb, alpha.reshape(h, w), int(h), int(w), float(fx), float(fy), float(cx),
                           float(cy), c2w, jitter)
