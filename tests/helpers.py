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


class TorchStats:
    """`sgn_rast.densify.Stats` interface with `SplatfactoModel.after_train`'s own arithmetic in plain torch
    (sgn_splatfacto.py:520-541) — lets the CPU tests drive `Densifier` without the HIP statistics kernel."""

    def __init__(self):
        self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None

    def reset(self):
        self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None

    @torch.no_grad()
    def update(self, xys_grad, radii, last_size):
        visible = (radii > 0).flatten()
        grads = xys_grad.detach().norm(dim=-1)
        if self.xys_grad_norm is None:
            self.xys_grad_norm = grads
            self.vis_counts = torch.ones_like(grads)
        else:
            self.vis_counts[visible] = self.vis_counts[visible] + 1
            self.xys_grad_norm[visible] = grads[visible] + self.xys_grad_norm[visible]
        if self.max_2Dsize is None:
            self.max_2Dsize = torch.zeros_like(radii, dtype=torch.float32)
        self.max_2Dsize[visible] = torch.maximum(self.max_2Dsize[visible],
                                                 radii.detach()[visible] / float(max(last_size[0], last_size[1])))

    def sync(self, group=None):
        from sgn_rast.dp import sync_densify_stats
        if self.xys_grad_norm is not None:
            sync_densify_stats(self.xys_grad_norm, self.vis_counts, self.max_2Dsize, group=group)
