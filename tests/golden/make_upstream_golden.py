#!/usr/bin/env python
"""Parity-pin kit, part 1: golden vectors of the REAL gsplat for the hot path.

    pip install gsplat==0.1.11          # any box with an NVIDIA GPU; nothing of this repository is needed
    python make_upstream_golden.py --out upstream/            # writes one <scene>.npz per scene (~3 MB in all)
    # copy the directory to  <this repo>/tests/golden/upstream/  and run
    #   python -m pytest tests/test_upstream_golden.py            (CPU: both oracles)
    #   python -m pytest tests/test_upstream_golden.py -m gpu     (MI355X: the HIP kernels)

Why it exists.  The reference reaches this path through the third-party package `gsplat` (call sites
street_gaussians_ns/sgn_splatfacto.py:860-873, 939, 954-967, 982-994; sgn_splatfacto_scene_graph.py:285), which is
neither vendored nor installable where this repository is built, and the reference holds no vectors for it: the oracles
under oracle/ restate gsplat 0.1.x from its published algorithm and parity is "unpinned" (DESIGN.md section 2).  This
file is SELF-CONTAINED (torch + numpy + whichever implementation is asked for): run against the real package it freezes,
per scene, the inputs and EVERYTHING the path produces —
  * all 7 outputs of `project_gaussians`,
  * `gsplat.utils.compute_cumulative_intersects` / `bin_and_sort_gaussians`: unsorted + sorted 64-bit keys, sorted ids,
    `tile_bins`,
  * `spherical_harmonics` colours, the rgb (or N-channel) image and alpha of `rasterize_gaussians`,
  * every gradient of a fixed random-weight loss: means, scales, quats, colours / SH coefficients, opacities, and the
    retained gradients of xys / conics / depths between the two autograd nodes
— for the C1 plumbing scene and eleven edge scenes built to hit what is decided-not-verified here:
  `bbox_left_top`   splats whose 3-sigma box ends within one tile LEFT of / ABOVE the image   ([verify] A.1: `+ 1`
                    before or after the cast in the tile box)
  `ewa_clamp`       centres outside 1.3x the frustum whose boxes still reach the image         ([verify] A.5: does the
                    projection vjp see the forward's clamp?)
  `alpha_clamp`     alphas above 0.99                                                          ([decide] A.4: backward clamp)
  `depth_ties`, `near_plane`, `nd_colors` (D = 5), `odd_size_block8`, `sh_deg4`, `sh_deg0_k16`, `uint8_colors`,
  `empty` (nothing in front of the camera), `one_gaussian`.
`--impl oracle` runs the same script on this repository's C oracle (CPU) and `--impl hip` on its HIP operators: that is
how tests/test_upstream_golden.py proves, without gsplat, that the kit's scenes DISCRIMINATE the decided behaviours (a
file generated under the other reading fails the comparison and passes with the switch flipped).
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np
import torch

C0 = 0.28209479177387814


# ------------------------------------------------------------------------------------------------- scenes
def _camera(W, H, focal, cx=None, cy=None):
    return dict(W=int(W), H=int(H), fx=float(focal), fy=float(focal), cx=float(W / 2 if cx is None else cx),
                cy=float(H / 2 if cy is None else cy), viewmat=np.eye(4, dtype=np.float32))


def _rand_quats(g, n):
    q = torch.randn(n, 4, generator=g)
    return torch.nn.functional.normalize(q, dim=-1)


def _place(cam, u, v, z):
    """World points (identity camera looking down +z) that project to pixel (u, v) at depth z."""
    return torch.stack([(u - cam["cx"]) * z / cam["fx"], (v - cam["cy"]) * z / cam["fy"], z], dim=-1)


def _base(cam, means, scales, quats, opac, g, sh_degree=3, k=16, block=16, background=None, colors=None):
    n = means.shape[0]
    sc = dict(cam, N=n, block=int(block), glob_scale=1.0, clip=0.01,
              means=means.float().numpy(), scales=scales.float().numpy(), quats=quats.float().numpy(),
              opacities=opac.float().reshape(n, 1).numpy())
    if colors is None:
        dc = (torch.rand(n, 1, 3, generator=g) - 0.5) / C0
        rest = torch.randn(n, k - 1, 3, generator=g) * 0.05
        sc["coeffs"] = torch.cat([dc, rest], dim=1).numpy() if k > 1 else dc.numpy()
        sc["sh_degree"] = int(sh_degree)
        d = 3
    else:
        sc["colors"] = colors.numpy()
        d = colors.shape[-1]
    sc["background"] = (np.zeros(d, dtype=np.float32) if background is None else np.asarray(background, dtype=np.float32))
    sc["w_img"] = torch.rand(cam["H"], cam["W"], d, generator=g).numpy()
    sc["w_alpha"] = torch.rand(cam["H"], cam["W"], generator=g).numpy()
    sc["w_depth"] = torch.rand(n, generator=g).numpy()
    sc["w_comp"] = torch.rand(n, generator=g).numpy()
    return sc


def _uniform(cam, g, n, z_range=(1.0, 5.0), scale_range=(0.01, 0.10), widen=1.15):
    z = torch.rand(n, generator=g) * (z_range[1] - z_range[0]) + z_range[0]
    u = cam["cx"] + (torch.rand(n, generator=g) * 2 - 1) * widen * cam["W"] / 2
    v = cam["cy"] + (torch.rand(n, generator=g) * 2 - 1) * widen * cam["H"] / 2
    lo, hi = math.log(scale_range[0]), math.log(scale_range[1])
    scales = torch.exp(torch.rand(n, 3, generator=g) * (hi - lo) + lo)
    lim = math.log(0.98 / 0.02)
    opac = torch.sigmoid((torch.randn(n, generator=g) * 1.5).clamp(-lim, lim))
    return _place(cam, u, v, z), scales, _rand_quats(g, n), opac


def scene_c1(n=10000):
    """10 k Gaussians, 128x128, fx = fy = 128, z ~ U(1, 5)."""
    g = torch.Generator().manual_seed(0)
    cam = _camera(128, 128, 128.0)
    return _base(cam, *_uniform(cam, g, n), g)


def scene_bbox_left_top(n=1200):
    """Centres LEFT of / ABOVE the image whose 3-sigma box ends in (-16, 0): `tile_centre + tile_radius` in (-1, 0).
    CUDA's `(int)(c + r + 1)` gives 0 (empty box: culled), `_torch_impl`'s `(int)(c + r) + 1` gives 1 (listed in tile
    column / row 0).  A third of the splats straddle the border for contrast."""
    g = torch.Generator().manual_seed(1)
    cam = _camera(128, 96, 128.0)
    z = torch.rand(n, generator=g) * 2 + 3
    s = torch.exp(torch.rand(n, 1, generator=g) * 0.6 + math.log(0.06)).expand(n, 3).clone()     # sigma 1.5-4 px
    u = torch.rand(n, generator=g) * (cam["W"] + 30) - 30
    v = torch.rand(n, generator=g) * (cam["H"] + 30) - 30
    left = torch.arange(n) % 3 == 0
    top = torch.arange(n) % 3 == 1
    u[left] = -torch.rand(int(left.sum()), generator=g) * 26 - 2        # centre 2-28 px left of the image
    v[top] = -torch.rand(int(top.sum()), generator=g) * 26 - 2
    opac = torch.rand(n, generator=g) * 0.6 + 0.35                      # above 0.35: alpha reaches 1/255 beyond 3 sigma
    return _base(cam, _place(cam, u, v, z), s, _rand_quats(g, n), opac, g)


def scene_ewa_clamp(n=600):
    """Centres outside 1.3 x the frustum (|x / z| > 1.3 tan(fov / 2)) with scales large enough for the 3-sigma box to
    reach the image: the forward's Jacobian is clamped; upstream's CUDA vjp is believed to use the un-clamped point."""
    g = torch.Generator().manual_seed(2)
    cam = _camera(128, 128, 128.0)
    z = torch.rand(n, generator=g) * 2 + 3
    side = torch.arange(n) % 4
    lim = 1.3 * 0.5 * cam["W"]                       # pixels from the principal point where the clamp starts (83.2)
    off = lim + 4 + torch.rand(n, generator=g) * 30  # 87-117 px from the centre: 23-53 px outside the image
    u = torch.where(side == 0, cam["cx"] - off, torch.where(side == 1, cam["cx"] + off, cam["cx"] + (torch.rand(n, generator=g) - 0.5) * 100))
    v = torch.where(side == 2, cam["cy"] - off, torch.where(side == 3, cam["cy"] + off, cam["cy"] + (torch.rand(n, generator=g) - 0.5) * 100))
    s = torch.exp(torch.rand(n, 3, generator=g) * 0.5 + math.log(0.7))         # sigma ~ 20-40 px at z ~ 4
    opac = torch.rand(n, generator=g) * 0.5 + 0.3
    return _base(cam, _place(cam, u, v, z), s, _rand_quats(g, n), opac, g)


def scene_alpha_clamp(n=800):
    """Opacities in [0.985, 0.9999]: alpha above the backward's 0.99 clamp (and a few above the forward's 0.999)."""
    g = torch.Generator().manual_seed(3)
    cam = _camera(96, 96, 96.0)
    m, s, q, _ = _uniform(cam, g, n, z_range=(2.0, 6.0), scale_range=(0.03, 0.12), widen=0.9)
    opac = 0.985 + torch.rand(n, generator=g) * 0.0149
    return _base(cam, m, s, q, opac, g)


def scene_depth_ties(n=2000):
    """Groups of 8 Gaussians at EXACTLY the same depth (identical 32 depth bits in the sort key): the order inside a
    tile is then the emission order (ascending Gaussian id) iff the sort is stable."""
    g = torch.Generator().manual_seed(4)
    cam = _camera(128, 128, 128.0)
    m, s, q, o = _uniform(cam, g, n)
    zq = (torch.rand(n // 8, generator=g) * 4 + 1).repeat_interleave(8)[:n]
    u = cam["cx"] + m[:, 0] / m[:, 2] * cam["fx"]
    v = cam["cy"] + m[:, 1] / m[:, 2] * cam["fy"]
    return _base(cam, _place(cam, u, v, zq), s, q, o, g)


def scene_near_plane(n=1500):
    """Depths around clip_thresh = 0.01 (z <= 0.01 is culled) and behind the camera."""
    g = torch.Generator().manual_seed(5)
    cam = _camera(128, 128, 128.0)
    m, s, q, o = _uniform(cam, g, n)
    z = torch.rand(n, generator=g) * 0.03 - 0.005            # -0.005 .. 0.025
    z[::5] = torch.rand(len(z[::5]), generator=g) * 4 + 1    # a fifth at ordinary depths
    u = cam["cx"] + (torch.rand(n, generator=g) - 0.5) * cam["W"]
    v = cam["cy"] + (torch.rand(n, generator=g) - 0.5) * cam["H"]
    s = torch.where((z < 0.1)[:, None], s * 0.002, s)        # tiny splats close to the camera stay finite on screen
    return _base(cam, _place(cam, u, v, z), s, q, o, g)


def scene_nd_colors(n=1500):
    """D = 5 colour channels (upstream's N-D rasterize path), non-zero background."""
    g = torch.Generator().manual_seed(6)
    cam = _camera(96, 64, 96.0)
    m, s, q, o = _uniform(cam, g, n)
    return _base(cam, m, s, q, o, g, colors=torch.rand(n, 5, generator=g), background=[0.1, 0.2, 0.3, 0.4, 0.5])


def scene_odd_size_block8(n=2500):
    """100x70 pixels (not a multiple of the tile), block_width 8, off-centre principal point, background (.2, .4, .6),
    SH degree 2 with K = 9."""
    g = torch.Generator().manual_seed(7)
    cam = _camera(100, 70, 90.0, cx=47.5, cy=37.25)
    return _base(cam, *_uniform(cam, g, n), g, sh_degree=2, k=9, block=8, background=[0.2, 0.4, 0.6])


def scene_sh_deg4(n=1500):
    g = torch.Generator().manual_seed(8)
    cam = _camera(96, 96, 96.0)
    return _base(cam, *_uniform(cam, g, n), g, sh_degree=4, k=25)


def scene_sh_deg0_k16(n=1500):
    """The training layout early on: K = 16 coefficients held, degree 0 used (sgn_splatfacto.py:936)."""
    g = torch.Generator().manual_seed(9)
    cam = _camera(96, 96, 96.0)
    return _base(cam, *_uniform(cam, g, n), g, sh_degree=0, k=16)


def scene_uint8_colors(n=1200):
    g = torch.Generator().manual_seed(10)
    cam = _camera(96, 64, 96.0)
    m, s, q, o = _uniform(cam, g, n)
    return _base(cam, m, s, q, o, g, colors=torch.randint(0, 256, (n, 3), generator=g, dtype=torch.uint8))


def scene_empty(n=64):
    """Everything behind the camera: no intersections (rasterize returns the background, zero gradients)."""
    g = torch.Generator().manual_seed(11)
    cam = _camera(64, 48, 64.0)
    m, s, q, o = _uniform(cam, g, n)
    m[:, 2] = -m[:, 2]
    return _base(cam, m, s, q, o, g, background=[0.3, 0.5, 0.7])


def scene_one_gaussian():
    g = torch.Generator().manual_seed(12)
    cam = _camera(64, 64, 64.0)
    return _base(cam, torch.tensor([[0.1, -0.2, 3.0]]), torch.tensor([[0.3, 0.1, 0.2]]), _rand_quats(g, 1),
                 torch.tensor([0.7]), g)


SCENES = {"c1": scene_c1, "bbox_left_top": scene_bbox_left_top, "ewa_clamp": scene_ewa_clamp,
          "alpha_clamp": scene_alpha_clamp, "depth_ties": scene_depth_ties, "near_plane": scene_near_plane,
          "nd_colors": scene_nd_colors, "odd_size_block8": scene_odd_size_block8, "sh_deg4": scene_sh_deg4,
          "sh_deg0_k16": scene_sh_deg0_k16, "uint8_colors": scene_uint8_colors, "empty": scene_empty,
          "one_gaussian": scene_one_gaussian}
# which decided behaviour a scene is there to settle (tests/test_upstream_golden.py names the switch in its message)
SETTLES = {"bbox_left_top": "tile_bbox_add_after_cast", "ewa_clamp": "ewa_vjp_clamped", "alpha_clamp": "alpha_clamp_bwd"}


# ------------------------------------------------------------------------------------------------- implementations
class Impl:
    """The gsplat 0.1.x operator surface of one implementation + the device its tensors live on."""

    def __init__(self, name):
        self.name = name
        if name == "gsplat":
            import gsplat
            from gsplat import utils
            from gsplat.project_gaussians import project_gaussians
            from gsplat.rasterize import rasterize_gaussians
            from gsplat.sh import spherical_harmonics
            assert torch.cuda.is_available(), "gsplat 0.1.x is CUDA only"
            self.device, self.version = torch.device("cuda"), getattr(gsplat, "__version__", "?")
            self.project_gaussians, self.rasterize_gaussians = project_gaussians, rasterize_gaussians
            self.spherical_harmonics = spherical_harmonics
            self.compute_cumulative_intersects = utils.compute_cumulative_intersects
            self.bin_and_sort_gaussians = utils.bin_and_sort_gaussians
        elif name == "oracle":                     # this repository's C oracle behind the same surface (CPU)
            root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            sys.path[:0] = [p for p in (root, os.path.join(root, "tests")) if p not in sys.path]
            import oracle_ops as O
            self.device, self.version = torch.device("cpu"), "sgn C oracle"
            self.project_gaussians, self.rasterize_gaussians = O.project_gaussians, O.rasterize_gaussians
            self.spherical_harmonics = O.spherical_harmonics
            self.compute_cumulative_intersects = O.compute_cumulative_intersects
            self.bin_and_sort_gaussians = O.bin_and_sort_gaussians
        elif name == "hip":                        # this repository's HIP operators through the gsplat import shim
            root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            pkg = os.path.join(root, "street-gaussians-ns_amd")
            sys.path[:0] = [p for p in (pkg,) if p not in sys.path]
            from gsplat import utils
            from gsplat.project_gaussians import project_gaussians
            from gsplat.rasterize import rasterize_gaussians
            from gsplat.sh import spherical_harmonics
            self.device, self.version = torch.device("cuda"), "sgn_rast (HIP)"
            self.project_gaussians, self.rasterize_gaussians = project_gaussians, rasterize_gaussians
            self.spherical_harmonics = spherical_harmonics
            self.compute_cumulative_intersects = utils.compute_cumulative_intersects
            self.bin_and_sort_gaussians = utils.bin_and_sort_gaussians
        else:
            raise ValueError(name)


def run_scene(impl: Impl, sc: dict) -> dict:
    """One forward + backward of the path on `sc`; every tensor the path produces, as numpy arrays."""
    dev = impl.device
    t = lambda a, grad=False: torch.tensor(np.asarray(a), device=dev).requires_grad_(grad)
    H, W, block = sc["H"], sc["W"], sc["block"]
    means, scales, quats = t(sc["means"], True), t(sc["scales"], True), t(sc["quats"], True)
    opac = t(sc["opacities"], True)
    viewmat = t(sc["viewmat"])
    out = {}
    xys, depths, radii, conics, comp, nth, cov3d = impl.project_gaussians(
        means, scales, sc["glob_scale"], quats, viewmat, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, block, sc["clip"])
    for x in (xys, depths, conics):
        if x.requires_grad:
            x.retain_grad()
    out.update(xys=xys, depths=depths, radii=radii, conics=conics, compensation=comp, num_tiles_hit=nth, cov3d=cov3d)
    tile_bounds = ((W + block - 1) // block, (H + block - 1) // block, 1)
    n_isect, cum = impl.compute_cumulative_intersects(nth)
    out["num_intersects"] = np.asarray([n_isect], dtype=np.int64)
    out["cum_tiles_hit"] = cum
    if n_isect > 0:
        ids_u, gids_u, ids_s, gids_s, bins = impl.bin_and_sort_gaussians(
            sc["N"], n_isect, xys.detach(), depths.detach(), radii, cum, tile_bounds, block)
        out.update(isect_ids_unsorted=ids_u, gaussian_ids_unsorted=gids_u, isect_ids_sorted=ids_s,
                   gaussian_ids_sorted=gids_s, tile_bins=bins)
    coeffs = colors_leaf = None
    if "coeffs" in sc:
        coeffs = t(sc["coeffs"], True)
        viewdirs = means.detach() - torch.zeros(3, device=dev)        # camera at the origin
        viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
        sh = impl.spherical_harmonics(sc["sh_degree"], viewdirs, coeffs)
        out["sh_colors"] = sh
        colors = torch.clamp(sh + 0.5, min=0.0)                       # sgn_splatfacto.py:940
    elif np.asarray(sc["colors"]).dtype == np.uint8:
        colors = t(sc["colors"])                                      # uint8: upstream divides by 255, no gradient
    else:
        colors = colors_leaf = t(sc["colors"], True)
    img, alpha = impl.rasterize_gaussians(xys, depths, radii, conics, nth, colors, opac, H, W, block,
                                          background=t(sc["background"]), return_alpha=True)
    out.update(out_img=img, out_alpha=alpha)
    loss = (img * t(sc["w_img"])).sum() + (alpha * t(sc["w_alpha"])).sum() \
        + 1e-3 * (depths * t(sc["w_depth"])).sum() + 1e-2 * (comp * t(sc["w_comp"])).sum()
    out["loss"] = loss.reshape(1)
    if loss.requires_grad:
        loss.backward()
    z = lambda x, like: torch.zeros_like(like) if x is None else x
    out.update(grad_means=z(means.grad, means), grad_scales=z(scales.grad, scales), grad_quats=z(quats.grad, quats),
               grad_opacities=z(opac.grad, opac), grad_xys=z(xys.grad, xys), grad_conics=z(conics.grad, conics),
               grad_depths=z(depths.grad, depths))
    if coeffs is not None:
        out["grad_coeffs"] = z(coeffs.grad, coeffs)
    if colors_leaf is not None:
        out["grad_colors"] = z(colors_leaf.grad, colors_leaf)
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}


def write_scene(impl: Impl, name: str, out_dir: str) -> str:
    sc = SCENES[name]()
    res = run_scene(impl, sc)
    d = {"in_" + k: np.asarray(v) for k, v in sc.items()}
    d.update({"out_" + k: v for k, v in res.items()})
    d["meta"] = np.asarray([impl.name, str(impl.version), torch.__version__,
                            torch.cuda.get_device_name(0) if impl.device.type == "cuda" else "cpu"])
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **d)
    return path


def load_scene(path: str):
    """(scene inputs, frozen outputs, meta) of one golden file."""
    z = np.load(path, allow_pickle=False)
    sc, res = {}, {}
    for k in z.files:
        if k.startswith("in_"):
            v = z[k]
            sc[k[3:]] = v.item() if v.ndim == 0 else v
        elif k.startswith("out_"):
            res[k[4:]] = z[k]
    return sc, res, [str(x) for x in z["meta"]]


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "upstream"))
    ap.add_argument("--impl", default="gsplat", choices=["gsplat", "oracle", "hip"])
    ap.add_argument("--scenes", default=",".join(SCENES))
    a = ap.parse_args()
    impl = Impl(a.impl)
    for nm in a.scenes.split(","):
        p = write_scene(impl, nm, a.out)
        print(f"{nm:18s} -> {p} ({os.path.getsize(p)} bytes)")
    print(f"implementation: {impl.name} {impl.version}; copy {a.out}/ to tests/golden/upstream/ of the repository")
