"""ctypes front-end for oracle/c/libsgn_oracle.so (the plain-C restatement).

TEST INFRASTRUCTURE ONLY — see the header of oracle/c/sgn_oracle.c ("parity
unpinned").  Takes and returns CPU torch tensors; mirrors the C-ABI of the
product library one-to-one so parity tests can feed both the same buffers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libsgn_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c", "sgn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.sgo_exp_eval.restype = C.c_float
        _lib.sgo_exp_eval.argtypes = [C.c_float]
    return _lib


def _p(t: torch.Tensor):
    assert t.is_contiguous() and t.device.type == "cpu", (t.shape, t.device)
    return C.c_void_p(t.data_ptr())


def _f(t):
    return t.detach().to(torch.float32).contiguous()


def set_exp_mode(mode: int) -> None:
    lib().sgo_set_exp_mode(C.c_int(mode))


def exp_eval(x: float) -> float:
    return float(lib().sgo_exp_eval(C.c_float(x)))


def project_fwd(means, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, block, clip=0.01):
    means, scales, quats = _f(means), _f(scales), _f(quats)
    V = _f(viewmat).reshape(-1)[:12].contiguous()
    N = means.shape[0]
    cov3d = torch.zeros(N, 6); xys = torch.zeros(N, 2); depths = torch.zeros(N)
    radii = torch.zeros(N, dtype=torch.int32); conics = torch.zeros(N, 3)
    comp = torch.zeros(N); nth = torch.zeros(N, dtype=torch.int32)
    lib().sgo_project_fwd(C.c_int(N), _p(means), _p(scales), C.c_float(glob_scale), _p(quats), _p(V),
                          C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_int(H),
                          C.c_int(W), C.c_int(block), C.c_float(clip), _p(cov3d), _p(xys), _p(depths),
                          _p(radii), _p(conics), _p(comp), _p(nth))
    return xys, depths, radii, conics, comp, nth, cov3d


def project_bwd(means, scales, glob_scale, quats, viewmat, fx, fy, cov3d, radii, conics, comp,
                v_xy, v_depth, v_conic, v_comp):
    means, scales, quats = _f(means), _f(scales), _f(quats)
    V = _f(viewmat).reshape(-1)[:12].contiguous()
    N = means.shape[0]
    v_cov2d = torch.zeros(N, 3); v_cov3d = torch.zeros(N, 6)
    v_mean = torch.zeros(N, 3); v_scale = torch.zeros(N, 3); v_quat = torch.zeros(N, 4)
    lib().sgo_project_bwd(C.c_int(N), _p(means), _p(scales), C.c_float(glob_scale), _p(quats), _p(V),
                          C.c_float(fx), C.c_float(fy), _p(_f(cov3d)), _p(radii.contiguous()),
                          _p(_f(conics)), _p(_f(comp)), _p(_f(v_xy)), _p(_f(v_depth)), _p(_f(v_conic)),
                          _p(_f(v_comp)), _p(v_cov2d), _p(v_cov3d), _p(v_mean), _p(v_scale), _p(v_quat))
    return v_mean, v_scale, v_quat, v_cov2d, v_cov3d


def sh_fwd(deg, dirs, coeffs):
    dirs, coeffs = _f(dirs), _f(coeffs)
    N, K = coeffs.shape[0], coeffs.shape[1]
    out = torch.zeros(N, 3)
    lib().sgo_sh_fwd(C.c_int(N), C.c_int(K), C.c_int(deg), _p(dirs), _p(coeffs), _p(out))
    return out


def sh_bwd(deg, K, dirs, v_colors):
    dirs, v_colors = _f(dirs), _f(v_colors)
    N = dirs.shape[0]
    out = torch.zeros(N, K, 3)
    lib().sgo_sh_bwd(C.c_int(N), C.c_int(K), C.c_int(deg), _p(dirs), _p(v_colors), _p(out))
    return out


def scan_i32(x):
    x = x.to(torch.int32).contiguous()
    out = torch.zeros_like(x)
    lib().sgo_scan_i32(C.c_int(x.numel()), _p(x), _p(out))
    return out


def map_isect(xys, depths, radii, cum, tiles_x, tiles_y, block):
    I = int(cum[-1]) if cum.numel() else 0
    keys = torch.zeros(I, dtype=torch.int64); vals = torch.zeros(I, dtype=torch.int32)
    lib().sgo_map_isect(C.c_int(xys.shape[0]), _p(_f(xys)), _p(_f(depths)), _p(radii.contiguous()),
                        _p(cum.contiguous()), C.c_int(tiles_x), C.c_int(tiles_y), C.c_int(block),
                        _p(keys), _p(vals))
    return keys, vals


def sort_pairs(keys, vals):
    ko = torch.zeros_like(keys); vo = torch.zeros_like(vals)
    lib().sgo_sort_pairs(C.c_int64(keys.numel()), _p(keys.contiguous()), _p(vals.contiguous()), _p(ko), _p(vo))
    return ko, vo


def tile_bins(keys_sorted, n_tiles):
    bins = torch.zeros(n_tiles, 2, dtype=torch.int32)
    lib().sgo_tile_bins(C.c_int64(keys_sorted.numel()), _p(keys_sorted.contiguous()), _p(bins))
    return bins


def bin_and_sort(xys, depths, radii, num_tiles_hit, H, W, block):
    tiles_x, tiles_y = (W + block - 1) // block, (H + block - 1) // block
    cum = scan_i32(num_tiles_hit)
    keys, vals = map_isect(xys, depths, radii, cum, tiles_x, tiles_y, block)
    ks, vs = sort_pairs(keys, vals)
    bins = tile_bins(ks, tiles_x * tiles_y)
    return cum, keys, vals, ks, vs, bins


def raster_fwd(H, W, block, ids, bins, xys, conics, colors, opac, bg, rows=None):
    """``rows=(lo, hi)``: pixel rows [lo, hi) only (the rest of the outputs stays zero) — a band of a
    BASELINE-size image is a band of the full result, at a fraction of the CPU time."""
    out = torch.zeros(H, W, 3); fT = torch.zeros(H, W); fi = torch.zeros(H, W, dtype=torch.int32)
    lo, hi = (0, H) if rows is None else rows
    lib().sgo_raster_fwd_rows(C.c_int(H), C.c_int(W), C.c_int(block), _p(ids.contiguous()), _p(bins.contiguous()),
                              _p(_f(xys)), _p(_f(conics)), _p(_f(colors)), _p(_f(opac).reshape(-1)), _p(_f(bg)),
                              _p(out), _p(fT), _p(fi), C.c_int(lo), C.c_int(hi))
    return out, fT, fi


def raster_bwd(H, W, block, ids, bins, xys, conics, colors, opac, bg, final_T, final_idx, v_out,
               v_out_alpha, alpha_clamp_bwd=0.99, rows=None):
    """``rows=(lo, hi)``: contributions of pixel rows [lo, hi) only (= the full backward when ``v_out`` and
    ``v_out_alpha`` vanish outside the band)."""
    N = xys.shape[0]
    v_xy = torch.zeros(N, 2); v_conic = torch.zeros(N, 3); v_col = torch.zeros(N, 3); v_op = torch.zeros(N, 1)
    lo, hi = (0, H) if rows is None else rows
    lib().sgo_raster_bwd_rows(C.c_int(H), C.c_int(W), C.c_int(block), C.c_int(N), _p(ids.contiguous()),
                              _p(bins.contiguous()), _p(_f(xys)), _p(_f(conics)), _p(_f(colors)),
                              _p(_f(opac).reshape(-1)), _p(_f(bg)), _p(_f(final_T)), _p(final_idx.contiguous()),
                              _p(_f(v_out)), _p(_f(v_out_alpha)), C.c_float(alpha_clamp_bwd), _p(v_xy),
                              _p(v_conic), _p(v_col), _p(v_op), C.c_int(lo), C.c_int(hi))
    return v_xy, v_conic, v_col, v_op


def cube_texture(tex, dirs, v_out=None):
    """tex [6,R,R,C], dirs [n,3] -> out [n,C] (and the texture gradient for ``v_out`` [n,C] when given)."""
    tex, d = _f(tex), _f(dirs).reshape(-1, 3).contiguous()
    n, R, Cc = d.shape[0], tex.shape[1], tex.shape[3]
    out = torch.zeros(n, Cc)
    v_tex = torch.zeros_like(tex) if v_out is not None else None
    v = _f(v_out).reshape(n, Cc).contiguous() if v_out is not None else None
    lib().sgo_cube_texture(n, R, Cc, _p(tex), _p(d), _p(out), _p(v) if v is not None else None,
                           _p(v_tex) if v_tex is not None else None)
    return (out, v_tex) if v_out is not None else out


def l1_ssim(x, y, data_range: float = 1.0):
    """(mean |y - x|, mean SSIM) of two [H,W,3] images, double precision inside."""
    x, y = _f(x), _f(y)
    fn = lib().sgo_l1_ssim
    fn.restype = C.c_double
    l1 = C.c_double(0.0)
    s = fn(x.shape[0], x.shape[1], _p(x), _p(y), C.c_double(data_range), C.byref(l1))
    return float(l1.value), float(s)
