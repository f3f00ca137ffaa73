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


from sgn_rast import densify as _densify


class TorchStats(_densify.Stats):
    """`sgn_rast.densify.Stats` with `SplatfactoModel.after_train`'s own arithmetic in plain torch
    (sgn_splatfacto.py:520-541) in place of the HIP statistics kernel — lets the CPU tests drive `Densifier`, and its
    view-parallel bookkeeping (first view of the interval, joining with zeros: the product's code), without a GPU."""

    def _accumulate(self, g, r, max_dim, first):
        visible = (r > 0).flatten()
        grads = g.norm(dim=-1)
        if first:
            self.xys_grad_norm.copy_(grads)
            self.vis_counts.fill_(1.0)
            self.max_2Dsize.zero_()
        else:
            self.vis_counts[visible] = self.vis_counts[visible] + 1
            self.xys_grad_norm[visible] = grads[visible] + self.xys_grad_norm[visible]
        self.max_2Dsize[visible] = torch.maximum(self.max_2Dsize[visible], r[visible] / float(max_dim))
