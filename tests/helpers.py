"""Shared builders for the parity tests (CPU tensors; the GPU tests move them over)."""
import torch

from sgn_rast import scenes


def activated(P):
    """The activations the reference applies before calling the ops (sgn_splatfacto.py:857,864,949)."""
    scales = P["log_scales"].exp()
    quats = P["quats"] / P["quats"].norm(dim=-1, keepdim=True)
    opac = torch.sigmoid(P["opacity_logits"])
    coeffs = torch.cat([P["features_dc"], P["features_rest"]], dim=1)
    return scales, quats, opac, coeffs


def small_scene(n=3000, w=128, h=128, focal=128.0, seed=0, z_range=(1.0, 5.0)):
    cam = scenes.make_camera(w, h, focal)
    P = scenes.make_gaussians(n, cam, seed=seed, z_range=z_range)
    return cam, P


def rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
