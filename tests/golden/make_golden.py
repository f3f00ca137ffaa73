"""Regenerates tests/golden/step_c1small.npz from the CPU oracle.

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


if __name__ == "__main__":
    d = build()
    path = os.path.join(HERE, "step_c1small.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in d.items()})
