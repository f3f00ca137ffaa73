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


# upstream-variant semantics bits (sgn_oracle.c SGO_SEM_*; 0 = the decided behaviours)
SEM_BBOX_ADD_AFTER_CAST, SEM_EWA_VJP_CLAMPED = 1, 2


def project_fwd(means, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, block, clip=0.01, semantics=0):
    means, scales, quats = _f(means), _f(scales), _f(quats)
    V = _f(viewmat).reshape(-1)[:12].contiguous()
    N = means.shape[0]
    cov3d = torch.zeros(N, 6); xys = torch.zeros(N, 2); depths = torch.zeros(N)
    radii = torch.zeros(N, dtype=torch.int32); conics = torch.zeros(N, 3)
    comp = torch.zeros(N); nth = torch.zeros(N, dtype=torch.int32)
    lib().sgo_project_fwd(C.c_int(N), _p(means), _p(scales), C.c_float(glob_scale), _p(quats), _p(V),
                          C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_int(H),
                          C.c_int(W), C.c_int(block), C.c_float(clip), _p(cov3d), _p(xys), _p(depths),
                          _p(radii), _p(conics), _p(comp), _p(nth), C.c_int(semantics))
    return xys, depths, radii, conics, comp, nth, cov3d


def project_bwd(means, scales, glob_scale, quats, viewmat, fx, fy, cov3d, radii, conics, comp,
                v_xy, v_depth, v_conic, v_comp, semantics=0, H=16, W=16):
    means, scales, quats = _f(means), _f(scales), _f(quats)
    V = _f(viewmat).reshape(-1)[:12].contiguous()
    N = means.shape[0]
    v_cov2d = torch.zeros(N, 3); v_cov3d = torch.zeros(N, 6)
    v_mean = torch.zeros(N, 3); v_scale = torch.zeros(N, 3); v_quat = torch.zeros(N, 4)
    lib().sgo_project_bwd(C.c_int(N), _p(means), _p(scales), C.c_float(glob_scale), _p(quats), _p(V),
                          C.c_float(fx), C.c_float(fy), _p(_f(cov3d)), _p(radii.contiguous()),
                          _p(_f(conics)), _p(_f(comp)), _p(_f(v_xy)), _p(_f(v_depth)), _p(_f(v_conic)),
                          _p(_f(v_comp)), _p(v_cov2d), _p(v_cov3d), _p(v_mean), _p(v_scale), _p(v_quat),
                          C.c_int(semantics), C.c_int(H), C.c_int(W))
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


def map_isect(xys, depths, radii, cum, tiles_x, tiles_y, block, semantics=0):
    I = int(cum[-1]) if cum.numel() else 0
    keys = torch.zeros(I, dtype=torch.int64); vals = torch.zeros(I, dtype=torch.int32)
    lib().sgo_map_isect(C.c_int(xys.shape[0]), _p(_f(xys)), _p(_f(depths)), _p(radii.contiguous()),
                        _p(cum.contiguous()), C.c_int(tiles_x), C.c_int(tiles_y), C.c_int(block),
                        _p(keys), _p(vals), C.c_int(semantics))
    return keys, vals


def sort_pairs(keys, vals):
    ko = torch.zeros_like(keys); vo = torch.zeros_like(vals)
    lib().sgo_sort_pairs(C.c_int64(keys.numel()), _p(keys.contiguous()), _p(vals.contiguous()), _p(ko), _p(vo))
    return ko, vo


def tile_bins(keys_sorted, n_tiles):
    bins = torch.zeros(n_tiles, 2, dtype=torch.int32)
    lib().sgo_tile_bins(C.c_int64(keys_sorted.numel()), _p(keys_sorted.contiguous()), _p(bins))
    return bins


def bin_and_sort(xys, depths, radii, num_tiles_hit, H, W, block, semantics=0):
    tiles_x, tiles_y = (W + block - 1) // block, (H + block - 1) // block
    cum = scan_i32(num_tiles_hit)
    keys, vals = map_isect(xys, depths, radii, cum, tiles_x, tiles_y, block, semantics)
    ks, vs = sort_pairs(keys, vals)
    bins = tile_bins(ks, tiles_x * tiles_y)
    return cum, keys, vals, ks, vs, bins


# Host threads for the two compositing calls (default 1: the plain scalar oracle).  Pixel rows are independent, so
# THREADS > 1 hands disjoint row chunks of [lo, hi) to a thread pool (ctypes releases the GIL inside the C call): the
# forward writes disjoint rows of the same buffers (bit-identical to one call), the backward sums the per-chunk
# per-Gaussian results in chunk order (deterministic; differs from the one-call result by float rounding of the
# partial sums only).  Used by the long convergence test to keep its CPU side short; nothing else sets it.
THREADS = 1


def _row_chunks(lo, hi):
    n = max(1, min(int(THREADS), hi - lo))
    edges = [lo + (hi - lo) * i // n for i in range(n + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _run_chunks(fn, chunks):
    if len(chunks) == 1:
        return [fn(*chunks[0])]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(chunks)) as ex:
        return list(ex.map(lambda c: fn(*c), chunks))


def raster_fwd(H, W, block, ids, bins, xys, conics, colors, opac, bg, rows=None):
    """``rows=(lo, hi)``: pixel rows [lo, hi) only (the rest of the outputs stays zero) — a band of a
    BASELINE-size image is a band of the full result, at a fraction of the CPU time."""
    out = torch.zeros(H, W, 3); fT = torch.zeros(H, W); fi = torch.zeros(H, W, dtype=torch.int32)
    lo, hi = (0, H) if rows is None else rows
    a = (ids.contiguous(), bins.contiguous(), _f(xys), _f(conics), _f(colors), _f(opac).reshape(-1), _f(bg))
    L = lib()

    def go(r0, r1):
        L.sgo_raster_fwd_rows(C.c_int(H), C.c_int(W), C.c_int(block), *[_p(t) for t in a], _p(out), _p(fT), _p(fi),
                              C.c_int(r0), C.c_int(r1))
    _run_chunks(go, _row_chunks(lo, hi))
    return out, fT, fi


# relative difference between two correct-to-an-ulp exp implementations on the compositing's range [-5.6, 0]: 1 ulp of the
# result each + the rounding of the argument's scaling by log2(e) (|x| 2^-24 ln 2 <= 2.3e-7); see sgn_oracle.c
EXP_REL_EPS = 1e-6


def raster_threshold_adjacent(H, W, block, ids, bins, xys, conics, opac, eps_exp=EXP_REL_EPS, rows=None, other=None):
    """bool [H, W]: pixels whose forward walk puts the 1/255 skip test or the 1e-4 stop test between the values of two
    implementations — which differ by `eps_exp` (relative) in exp and, with `other = (xys, conics, opac)` of the other
    side, by whatever their inputs differ — the only pixels where the two may differ by more than rounding."""
    out = torch.zeros(H, W, dtype=torch.int32)
    lo, hi = (0, H) if rows is None else rows
    first = (_f(xys), _f(conics), _f(opac).reshape(-1))
    second = first if other is None else (_f(other[0]), _f(other[1]), _f(other[2]).reshape(-1))
    a = (ids.contiguous(), bins.contiguous()) + first + second
    L = lib()

    def go(r0, r1):
        L.sgo_raster_threshold_adjacent_rows(C.c_int(H), C.c_int(W), C.c_int(block), *[_p(t) for t in a],
                                             C.c_float(eps_exp), _p(out), C.c_int(r0), C.c_int(r1))
    _run_chunks(go, _row_chunks(lo, hi))
    return out.bool()


def raster_bwd(H, W, block, ids, bins, xys, conics, colors, opac, bg, final_T, final_idx, v_out,
               v_out_alpha, alpha_clamp_bwd=0.99, rows=None):
    """``rows=(lo, hi)``: contributions of pixel rows [lo, hi) only (= the full backward when ``v_out`` and
    ``v_out_alpha`` vanish outside the band)."""
    N = xys.shape[0]
    lo, hi = (0, H) if rows is None else rows
    a = (ids.contiguous(), bins.contiguous(), _f(xys), _f(conics), _f(colors), _f(opac).reshape(-1), _f(bg),
         _f(final_T), final_idx.contiguous(), _f(v_out), _f(v_out_alpha))
    L = lib()

    def go(r0, r1):
        o = (torch.zeros(N, 2), torch.zeros(N, 3), torch.zeros(N, 3), torch.zeros(N, 1))
        L.sgo_raster_bwd_rows(C.c_int(H), C.c_int(W), C.c_int(block), C.c_int(N), *[_p(t) for t in a],
                              C.c_float(alpha_clamp_bwd), *[_p(t) for t in o], C.c_int(r0), C.c_int(r1))
        return o
    parts = _run_chunks(go, _row_chunks(lo, hi))
    v_xy, v_conic, v_col, v_op = parts[0]
    for o in parts[1:]:
        v_xy += o[0]; v_conic += o[1]; v_col += o[2]; v_op += o[3]
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
