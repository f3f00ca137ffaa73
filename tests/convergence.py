"""Shared driver of the convergence / PSNR parity tests (test infrastructure).

`fit` trains a perturbed copy of a small scene towards a target image with the reference's photometric loss
(`sgn_splatfacto.py:1084-1087`: 0.8 L1 + 0.2 (1 - SSIM)) and its per-group Adam learning rates
(`sgn_config.py:71-108`, eps 1e-15), through the call-site replay `sgn_rast.step.train_step`, and returns the PSNR of
every step.  Run twice — operator namespace under test vs. the oracle's — the two trajectories must stay within the
north star's 0.05 dB."""
import math

import torch

LRS = {"means": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacity_logits": 0.05,
       "log_scales": 0.005, "quats": 0.001}


def make_problem(n=20_000, size=128, seed=0):
    """(camera, ground-truth params, perturbed start, target image rendered by the C oracle)."""
    import oracle_ops
    from sgn_rast import scenes, step
    cam = scenes.make_camera(size, size, float(size))
    truth = scenes.make_gaussians(n, cam, seed=seed, z_range=(1.0, 5.0))
    with torch.no_grad():
        gt = torch.clamp(step.render(step.leaf_params(truth), cam, ops=oracle_ops, caller_syncs=False).rgb, 0.0, 1.0)
    g = torch.Generator().manual_seed(seed + 1)
    start = {k: v.clone() for k, v in truth.items()}
    start["means"] += 0.01 * torch.randn(start["means"].shape, generator=g)
    start["log_scales"] += 0.1 * torch.randn(start["log_scales"].shape, generator=g)
    start["features_dc"] += 0.3 * torch.randn(start["features_dc"].shape, generator=g)
    start["opacity_logits"] += 0.3 * torch.randn(start["opacity_logits"].shape, generator=g)
    return cam, truth, start, gt.detach()


def psnr(img, gt):
    mse = float(((torch.clamp(img, 0.0, 1.0) - gt) ** 2).mean())
    return 10.0 * math.log10(1.0 / mse)


def fit(start, cam, gt, steps, device="cpu", ops=None, loss_fn=None, adam=None):
    """Returns (psnr per step, final params).  ``ops`` None = the product's HIP ops (device must be a GPU);
    ``adam`` = optimiser factory (default torch.optim.Adam on every side: same update rule, the thing under test is
    the rasterizer)."""
    from sgn_rast import scenes, step
    cam_d = scenes.Camera(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, cam.viewmat.to(device),
                          cam.cam_pos.to(device))
    P = step.leaf_params({k: v.to(device) for k, v in start.items()})
    make = adam or (lambda p, lr: torch.optim.Adam([p], lr=lr, eps=1e-15))
    opts = [make(P[k], LRS[k]) for k in P]
    gt_d = gt.to(device)
    w_img = torch.zeros(cam.height, cam.width, 3, device=device)
    w_a = torch.zeros(cam.height, cam.width, device=device)
    kw = {} if ops is None else dict(ops=ops, loss_fn=loss_fn)
    out_psnr = []
    for _ in range(steps):
        out = step.train_step(P, cam_d, w_img, w_a, gt=gt_d, **kw)
        out_psnr.append(psnr(out.rgb.detach(), gt_d))
        for o in opts:
            o.step()
    return out_psnr, {k: v.detach().cpu() for k, v in P.items()}


def oracle_loss(rgb, gt, lam):
    from oracle import torch_oracle as O
    l1, s = O.l1_ssim_losses(rgb, gt)
    return (1 - lam) * l1 + lam * (1 - s)
