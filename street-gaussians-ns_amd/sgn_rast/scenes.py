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


def make_scene_graph(n_total: int, cam: Camera, n_objects: int = 8, object_frac: float = 0.1, fourier_dim: int = 5,
                     seed: int = 0, z_range=(2.0, 60.0), device="cpu", object_depth=(8.0, 48.0), object_extent=1.0):
    """Background + ``n_objects`` rigid objects (configs[2] shape: scene-graph dynamic objects).  Returns
    (models, poses [M,16], idft [M,F]); object parameters are in the object's local frame, its Gaussians
    spread ~1.5 m around the pose centre; ``features_dc`` of objects carries ``fourier_dim`` coefficients
    (sgn_config.py:66), the background one (sgn_config.py:56)."""
    from .fused import make_pose_table
    g = torch.Generator().manual_seed(seed + 17)
    n_obj = int(n_total * object_frac) // max(1, n_objects)
    n_bg = n_total - n_obj * n_objects
    cam0 = make_camera(cam.width, cam.height, cam.fx)
    models = [make_gaussians(n_bg, cam0, seed=seed, z_range=z_range)]
    Rs, ts, idfts = [torch.eye(3)], [torch.zeros(3)], [torch.cat([torch.ones(1), torch.zeros(fourier_dim - 1)])]
    for k in range(n_objects):
        m = make_gaussians(n_obj, cam0, seed=seed + 1 + k, z_range=z_range)
        m["means"] = torch.randn(n_obj, 3, generator=g) * torch.tensor([1.0, 0.6, 1.5]) * object_extent
        dc = torch.randn(n_obj, fourier_dim, 3, generator=g) * 0.1
        dc[:, 0] += m["features_dc"][:, 0]
        m["features_dc"] = dc
        models.append(m)
        yaw = float(torch.rand(1, generator=g)) * 6.28
        c, s = math.cos(yaw), math.sin(yaw)
        Rs.append(torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]]))
        z = object_depth[0] + (object_depth[1] - object_depth[0]) * float(torch.rand(1, generator=g))
        x = (float(torch.rand(1, generator=g)) * 2 - 1) * 0.8 * z * (cam.width / 2.0 / cam.fx)
        ts.append(torch.tensor([x, 0.4 * z * (cam.height / 2.0 / cam.fy) * 0.5, z]))
        t_norm = float(torch.rand(1, generator=g))
        w = [math.cos(2 * math.pi * t_norm * kk / fourier_dim) if kk % 2 == 0
             else math.sin(2 * math.pi * t_norm * (kk + 1) / fourier_dim) for kk in range(fourier_dim)]
        idfts.append(torch.tensor(w))
    poses = make_pose_table(torch.stack(Rs), torch.stack(ts))
    models = [{k: v.to(device) for k, v in m.items()} for m in models]
    return models, poses.to(device), torch.stack(idfts).to(device)


def make_street_gaussians(n: int, cam: Camera, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """A deliberately NON-uniform scene for load-balance profiling (not a BASELINE config): empty sky in the
    upper part of the image, a ground plane, two facades, and 30 % of the Gaussians packed into 20 small
    semi-transparent "foliage" clusters, so a handful of tiles carry very long depth lists that do not saturate."""
    g = torch.Generator().manual_seed(seed + 99)
    raw = make_gaussians(n, cam, seed=seed)
    n_ground, n_wall = int(0.4 * n), int(0.3 * n)
    n_clu = n - n_ground - n_wall
    u = lambda k, lo, hi: torch.rand(k, generator=g) * (hi - lo) + lo
    ground = torch.stack([u(n_ground, -20, 20), torch.full((n_ground,), 1.6), u(n_ground, 2, 80)], -1)
    side = (torch.rand(n_wall, generator=g) > 0.5).float() * 2 - 1
    wall = torch.stack([side * 12.0, u(n_wall, -6, 1.6), u(n_wall, 4, 80)], -1)
    centres = torch.stack([u(20, -8, 8), u(20, -3, 0.5), u(20, 20, 60)], -1)
    which = torch.randint(0, 20, (n_clu,), generator=g)
    clu = centres[which] + torch.randn(n_clu, 3, generator=g) * 0.6
    raw["means"] = torch.cat([ground, wall, clu])
    raw["log_scales"][: n_ground, 1] -= 2.0                    # flat on the ground
    raw["log_scales"][n_ground + n_wall:] -= 0.7               # small leaves
    raw["opacity_logits"][n_ground + n_wall:] = -2.0           # alpha ~ 0.12: long lists, no early termination
    return {k: v.to(device) for k, v in raw.items()}
