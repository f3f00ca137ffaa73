"""End-to-end sanity: the pieces form a training loop that actually fits an image — HIP rasterizer (fused front
ends) + sky sphere + fused L1/SSIM loss + multi-tensor Adam, 60 steps on a small scene."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_training_loop_reduces_photometric_loss():
    from sgn_rast import loss, optim, scenes, step
    dev = "cuda"
    cam, raw_t = scenes.make_scene("c1", seed=11, device=dev, n_override=3000)
    w_img, w_a = step.loss_weights(cam, seed=1, device=dev)
    with torch.no_grad():                                    # target: another random scene's rendering
        target = torch.clamp(step.render_fused(step.leaf_params(raw_t), cam).rgb, max=1.0).detach()
    _, raw = scenes.make_scene("c1", seed=12, device=dev, n_override=3000)
    P = step.leaf_params(raw)
    c2w = torch.eye(4, device=dev)[:3]
    sky = {"base": (0.5 * torch.ones(6, 32, 32, 3, device=dev)).requires_grad_(True), "c2w": c2w, "train": True}
    lrs = {"means": 1.6e-3, "features_dc": 0.01, "features_rest": 0.0005, "opacity_logits": 0.05, "log_scales": 0.005,
           "quats": 0.001}
    opts = [optim.FusedAdam([P[k]], lr=lrs[k], eps=1e-15) for k in P] + [optim.FusedAdam([sky["base"]], lr=0.01)]
    history = []
    for it in range(60):
        out = step.train_step(P, cam, w_img, torch.zeros_like(w_a), 3, 16, fused=True, sky=sky, gt=target)
        optim.step_many(opts)
        history.append(float(out.loss))
    assert all(torch.isfinite(p).all() for p in P.values())
    first, last = sum(history[:5]) / 5, sum(history[-5:]) / 5
    assert last < 0.75 * first, (first, last)
    # and the loss the step reports is the reference composition on the final image
    with torch.no_grad():
        out = step.render_fused(P, cam)
        step.composite_sky(out, cam, sky["base"], c2w, train=False, fused=True)
        val = loss.photometric_loss(out.rgb, target, 0.2)
    assert float(val) < first
