"""Deterministic synthetic scenes shared by bench.py, smoke() and tests.

All tensors are generated on the CPU with a seeded ``torch.Generator`` and moved to
the requested device afterwards, so the CPU oracle and the HIP path see identical
bits.  Conventions are the reference call sites' (sgn_splatfacto.py:825-873):
camera looks down +z (gsplat / OpenCV), quats (w,x,y,z), SH coeffs [N,K,3].
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict

import torch

SH_C0 = 0.28209479177387814


@dataclass
class Camera:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    viewmat: torch.Tensor  # [4,4] world->camera, row-major
    cam_pos: torch.Tensor  # [3] camera centre in world coordinates


def make_camera(width: int, height: int, focal: float, yaw: float = 0.0, device="cpu") -> Camera:
    """Identity camera at the origin looking down +z, optionally yawed about the y axis
    (rank r of the data-parallel harness renders yaw offset r)."""
    c, s = math.cos(yaw), math.sin(yaw)
    R_c2w = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float32)
    viewmat = torch.eye(4, dtype=torch.float32)
    viewmat[:3, :3] = R_c2w.T
    return Camera(width, height, float(focal), float(focal), width / 2.0, height / 2.0,
                  viewmat.to(device), torch.zeros(3, dtype=torch.float32, device=device))


def make_gaussians(n: int, cam: Camera, seed: int = 0, z_range=(2.0, 60.0),
                   scale_range=(0.01, 0.10), sh_degree: int = 3, widen: float = 1.15,
                   device="cpu") -> Dict[str, torch.Tensor]:
    """Raw (pre-activation) parameters in the layout SplatfactoModel keeps them
    (sgn_splatfacto.py:251-268): means, log-scales, raw quats, opacity logits,
    features_dc [N,1,3], features_rest [N,K-1,3]."""
    g = torch.Generator().manual_seed(seed)
    z = torch.rand(n, generator=g) * (z_range[1] - z_range[0]) + z_range[0]
    ux = (torch.rand(n, generator=g) * 2 - 1) * widen * (cam.width / 2.0 / cam.fx)
    uy = (torch.rand(n, generator=g) * 2 - 1) * widen * (cam.height / 2.0 / cam.fy)
    means = torch.stack([ux * z, uy * z, z], dim=-1)
    lo, hi = math.log(scale_range[0]), math.log(scale_range[1])
    log_scales = torch.rand(n, 3, generator=g) * (hi - lo) + lo
    quats = torch.randn(n, 4, generator=g)
    lim = math.log(0.98 / 0.02)
    opac = (torch.randn(n, 1, generator=g) * 1.5).clamp(-lim, lim)
    k = (sh_degree + 1) ** 2
    dc = ((torch.rand(n, 1, 3, generator=g) - 0.5) / SH_C0)
    rest = torch.randn(n, k - 1, 3, generator=g) * 0.05
    out = dict(means=means, log_scales=log_scales, quats=quats, opacity_logits=opac,
               features_dc=dc, features_rest=rest)
    return {k_: v.to(device) for k_, v in out.items()}


SCENES = {
    # name: (N, W, H, focal, z_range)
    "c1": (10_000, 128, 128, 128.0, (1.0, 5.0)),          # BASELINE.json configs[0] (CPU plumbing)
    "c2": (500_000, 1920, 1280, 2000.0, (2.0, 60.0)),     # configs[1]
    "metric": (1_000_000, 1920, 1280, 2000.0, (2.0, 60.0)),  # BASELINE.json "metric"
    "c4": (2_000_000, 1920, 1280, 2000.0, (2.0, 60.0)),   # configs[3], per-rank yawed views
}


def make_scene(name: str, seed: int = 0, yaw: float = 0.0, device="cpu", n_override: int = 0):
    n, w, h, f, zr = SCENES[name]
    if n_override:
        n = n_override
    cam = make_camera(w, h, f, yaw=yaw, device=device)
    cam0 = make_camera(w, h, f, yaw=0.0)
    params = make_gaussians(n, cam0, seed=seed, z_range=zr, device=device)
    return cam, params
