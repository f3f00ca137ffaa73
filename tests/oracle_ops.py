"""The gsplat operator surface implemented on the C oracle (CPU), with upstream's analytic
backward — the *expected-value generator* for end-to-end parity tests and golden fixtures.
Test infrastructure: lives under tests/, never imported by the product."""
import torch
from torch.autograd import Function

from oracle import c_oracle as CO

ALPHA_CLAMP_BWD = 0.99  # gsplat 0.1.x backward clamp
SEMANTICS = 0           # upstream-variant bits (oracle.c_oracle.SEM_*): 0 = the decided behaviours
# (row_lo, row_hi), a list of such ranges (disjoint bands), or None: restrict the compositing (forward and backward) to
# bands of pixel rows.  Used by the BASELINE-size gradient parity tests together with loss weights that vanish outside the
# bands, so the bands' backward IS the full backward; image rows outside come back as zeros (not compared).
PIXEL_ROWS = None


def _bands(rows):
    if rows is None or isinstance(rows[0], int):
        return [rows]
    return list(rows)


class _Project(Function):
    @staticmethod
    def forward(ctx, means, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, block, clip):
        out = CO.project_fwd(means, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, block, clip, SEMANTICS)
        xys, depths, radii, conics, comp, nth, cov3d = out
        ctx.args = (glob_scale, fx, fy)
        ctx.sem = (SEMANTICS, H, W)
        ctx.save_for_backward(means.detach(), scales.detach(), quats.detach(), viewmat.detach(), cov3d, radii,
                              conics, comp)
        ctx.mark_non_differentiable(radii, nth)
        return xys, depths, radii, conics, comp, nth, cov3d

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_comp, v_nth, v_cov3d):
        means, scales, quats, viewmat, cov3d, radii, conics, comp = ctx.saved_tensors
        gs, fx, fy = ctx.args
        n = means.shape[0]
        z = lambda t, *s: torch.zeros(*s) if t is None else t
        vm, vs, vq, _, _ = CO.project_bwd(means, scales, gs, quats, viewmat, fx, fy, cov3d, radii, conics, comp,
                                          z(v_xys, n, 2), z(v_depths, n), z(v_conics, n, 3), z(v_comp, n), *ctx.sem)
        return (vm, vs, None, vq) + (None,) * 9


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width,
                      block_width, clip_thresh=0.01):
    return _Project.apply(means3d, scales, float(glob_scale), quats, viewmat, fx, fy, cx, cy, img_height,
                          img_width, block_width, clip_thresh)


class _SH(Function):
    @staticmethod
    def forward(ctx, deg, dirs, coeffs):
        ctx.deg, ctx.k = deg, coeffs.shape[1]
        ctx.save_for_backward(dirs.detach())
        return CO.sh_fwd(deg, dirs, coeffs)

    @staticmethod
    def backward(ctx, v_colors):
        (dirs,) = ctx.saved_tensors
        return None, None, CO.sh_bwd(ctx.deg, ctx.k, dirs, v_colors)


def spherical_harmonics(degrees_to_use, viewdirs, coeffs, method="fast"):
    return _SH.apply(degrees_to_use, viewdirs, coeffs)


class _Raster(Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, nth, colors, opacity, H, W, block, background, return_alpha):
        cum, keys, vals, ks, vs, bins = CO.bin_and_sort(xys, depths, radii, nth, H, W, block, SEMANTICS)
        img = fT = fi = None
        for band in _bands(PIXEL_ROWS):           # (outputs are zero outside a band: disjoint bands add up)
            o = CO.raster_fwd(H, W, block, vs, bins, xys, conics, colors, opacity, background, rows=band)
            img, fT, fi = o if img is None else (img + o[0], fT + o[1], fi + o[2])
        ctx.dims = (H, W, block)
        ctx.rows = PIXEL_ROWS
        ctx.oshape = opacity.shape
        ctx.save_for_backward(vs, bins, xys.detach(), conics.detach(), colors.detach(), opacity.detach(),
                              background, fT, fi)
        ctx.aux = (ks, vs, bins, fT, fi)
        if return_alpha:
            return img, 1 - fT
        return img

    @staticmethod
    def backward(ctx, v_img, v_alpha=None):
        vs, bins, xys, conics, colors, opacity, bg, fT, fi = ctx.saved_tensors
        H, W, block = ctx.dims
        if v_alpha is None:
            v_alpha = torch.zeros(H, W)
        acc = None
        for band in _bands(ctx.rows):
            o = CO.raster_bwd(H, W, block, vs, bins, xys, conics, colors, opacity, bg, fT, fi, v_img, v_alpha,
                              ALPHA_CLAMP_BWD, rows=band)
            acc = o if acc is None else tuple(a + b for a, b in zip(acc, o))
        v_xy, v_conic, v_col, v_op = acc
        return (v_xy, None, None, v_conic, None, v_col, v_op.reshape(ctx.oshape)) + (None,) * 5


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                        block_width, background=None, return_alpha=False):
    if colors.dtype == torch.uint8:
        colors = colors.float() / 255            # gsplat/rasterize.py: "make sure colors are float [0,1]"
    d = colors.shape[-1]
    if background is None:
        background = torch.ones(d)
    if d != 3:
        # upstream's N-D path composites every channel with the same per-pixel walk: three channels per pass of the
        # 3-channel oracle (zero-padded) give the same channels
        imgs, alpha = [], None
        for c0 in range(0, d, 3):
            w = min(3, d - c0)
            chunk, bg3 = colors[:, c0:c0 + w], background[c0:c0 + w]
            if w < 3:
                chunk = torch.cat([chunk, chunk.new_zeros(chunk.shape[0], 3 - w)], dim=1)
                bg3 = torch.cat([bg3, bg3.new_zeros(3 - w)])
            img, alpha = _Raster.apply(xys, depths, radii, conics, num_tiles_hit, chunk, opacity, img_height, img_width,
                                       block_width, bg3, True)
            imgs.append(img[..., :w])
        out = torch.cat(imgs, dim=-1)
        return (out, alpha) if return_alpha else out
    return _Raster.apply(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                         block_width, background, return_alpha)


# gsplat/utils.py: the two binning helpers the parity-pin kit freezes (tests/golden/make_upstream_golden.py)
def compute_cumulative_intersects(num_tiles_hit):
    cum = CO.scan_i32(num_tiles_hit)
    return (int(cum[-1]) if cum.numel() else 0), cum


def bin_and_sort_gaussians(num_points, num_intersects, xys, depths, radii, cum_tiles_hit, tile_bounds, block_width):
    keys, vals = CO.map_isect(xys, depths, radii, cum_tiles_hit, int(tile_bounds[0]), int(tile_bounds[1]), block_width,
                              SEMANTICS)
    ks, vs = CO.sort_pairs(keys, vals)
    return keys, vals, ks, vs, CO.tile_bins(ks, int(tile_bounds[0]) * int(tile_bounds[1]))
