"""Freezes the ONLY known answers the reference itself holds on this path's conventions:
`RGB2SH` / `SH2RGB` with C0 = 0.28209479177387814 (`street_gaussians_ns/sgn_splatfacto.py:57-70`) — the functions the
reference initialises `features_dc` with (`:259-262`) and reads colours back with.  They pin the degree-0 SH basis
constant and the `+ 0.5` the call sites add (`:940`): `spherical_harmonics(0, dirs, RGB2SH(rgb)[:, None, :]) + 0.5 == rgb`.

Imports the two functions from /root/reference (tests/refhost.py; nothing is copied), evaluates them on seeded inputs and
writes tests/golden/known_rgb2sh.npz so the GPU box (no reference checkout) can run the same check on the HIP kernels.

Run from the repo root in the build container:  python tests/golden/make_known_answers.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]

import refhost  # noqa: E402

OUT = os.path.join(HERE, "known_rgb2sh.npz")


def inputs():
    g = torch.Generator().manual_seed(20240926)
    rgb = torch.rand(4096, 3, generator=g)
    rgb[:8] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0],
                            [0.0, 0.0, 1.0], [0.25, 0.75, 0.125], [1e-3, 0.999, 0.5]])
    sh = torch.randn(4096, 3, generator=g) * 1.5
    dirs = torch.randn(4096, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    return rgb, sh, dirs


def compute():
    ns = refhost.load("oracle")
    rgb, sh, dirs = inputs()
    return dict(rgb=rgb.numpy(), sh=sh.numpy(), dirs=dirs.numpy(), RGB2SH_of_rgb=ns.splat.RGB2SH(rgb).numpy(),
                SH2RGB_of_sh=ns.splat.SH2RGB(sh).numpy(),
                SH2RGB_of_RGB2SH=ns.splat.SH2RGB(ns.splat.RGB2SH(rgb)).numpy())


if __name__ == "__main__":
    np.savez(OUT, **compute())
    print("wrote", OUT)
