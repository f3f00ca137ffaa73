"""Regenerates tests/golden/step_c1small.npz and tests/golden/aux_sky_loss.npz from the CPU oracle.

    python tests/golden/make_golden.py

PARITY UNPINNED: upstream gsplat is neither vendored in /root/reference nor installed here and the
reference ships no fixtures for this path, so these vectors are a pinned snapshot of
the repo's own oracle (C restatement with upstream's analytic backward, cross-checked against fp64
autograd in tests/test_oracle_consistency.py) — a regression anchor, not upstream ground truth.
Scene: 1500 Gaussians, 96x64, fx=fy=96, SH degree 3, seed 11, background (0,0,0), loss of
sgn_rast.step.train_step with loss_weights(seed=7), plus the depth pass of sgn_splatfacto.py:982-996.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]


def build():
    import oracle_ops
    from sgn_rast import scenes, step
    cam = scenes.make_camera(96, 64, 96.0)
    raw = scenes.make_gaussians(1500, cam, seed=11, z_range=(1.0, 5.0))
    raw["opacity_logits"][:60] = 8.0  # alpha above the 0.99 backward clamp
    P = step.leaf_params(raw)
    w_img, w_a = step.loss_weights(cam, seed=7)
    out = step.train_step(P, cam, w_img, w_a, with_depth=True, ops=oracle_ops)
    d = dict(rgb=out.rgb, alpha=out.alpha, depth=out.depth, xys=out.xys, depths=out.depths, radii=out.radii,
             conics=out.conics, num_tiles_hit=out.num_tiles_hit, xys_grad=out.xys.grad, loss=out.loss.reshape(1))
    for k, v in P.items():
        d["grad_" + k] = v.grad
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


def build_aux():
    """Sky cube map + photometric loss: same status — snapshots of the oracle
    restatements of nvdiffrast / pytorch_msssim behaviour, neither of which is available here."""
    from oracle import torch_oracle as O
    g = torch.Generator().manual_seed(21)
    tex = torch.rand(6, 8, 8, 3, generator=g).requires_grad_(True)
    c2w = torch.tensor([[0.36, -0.48, 0.8, 1.0], [0.8, 0.6, 0.0, 2.0], [-0.48, 0.64, 0.6, 3.0]])
    dirs = O.env_light_directions(24, 40, 14.0, 13.0, 20.0, 12.0, c2w)        # wide FOV: several faces
    sky = O.cube_texture(tex, dirs)
    w = torch.rand(24, 40, 3, generator=g)
    (sky * w).sum().backward()
    pred = torch.rand(40, 56, 3, generator=g).requires_grad_(True)
    gt = (pred.detach() + 0.2 * torch.randn(40, 56, 3, generator=g)).clamp(0, 1)
    l1, s = O.l1_ssim_losses(pred, gt)
    (0.8 * l1 + 0.2 * (1 - s)).backward()
    d = dict(sky_tex=tex, sky_c2w=c2w, sky_dirs=dirs, sky_out=sky, sky_w=w, sky_tex_grad=tex.grad,
             loss_pred=pred, loss_gt=gt, loss_l1=l1.reshape(1), loss_ssim=s.reshape(1), loss_pred_grad=pred.grad)
    return {k: v.detach().cpu().numpy() for k, v in d.items()}


if __name__ == "__main__":
    for name, fn in (("step_c1small.npz", build), ("aux_sky_loss.npz", build_aux)):
        d = fn()
        path = os.path.join(HERE, name)
        np.savez_compressed(path, **d)
        print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in d.items()})
