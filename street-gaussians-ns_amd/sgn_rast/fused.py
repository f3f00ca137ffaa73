"""Fused front ends — the caller-side glue of the reference folded into the kernels.

``BASELINE.json:north_star`` asks for "covariance projection with the scene-graph's per-object rigid
transform fused in".  The reference does that glue in Python around the gsplat ops
(``sgn_splatfacto.py:857-864, 934-940, 949``; ``sgn_splatfacto_scene_graph.py:239-247, 355-360,
404-433``): exp / normalise / sigmoid / +0.5 / clamp, view directions, the Fourier DC sum, the
quaternion product and ``means @ R.T + t`` per object, and three N-sized ``torch.cat``s per step.
These ops take the *raw* parameters (local means, log-scales, raw quats, opacity logits, un-concatenated
``features_dc`` / ``features_rest``) plus a per-Gaussian ``object_ids`` row index into small per-object
tables and do all of it in registers, forward and backward.  They are extensions: the drop-in
gsplat-shaped ops in :mod:`sgn_rast.ops` stay the reference-faithful path.
"""
from __future__ import annotations

from typing import Optional

import os

import torch
from torch.autograd import Function

from . import _lib as L
from . import ops as _ops
from .ops import _RasterizeGaussians, _f32c


def make_pose_table(rotations: torch.Tensor, translations: torch.Tensor) -> torch.Tensor:
    """[M,3,3] object->world rotations + [M,3] translations -> [M,16] rows (R row-major, t, q_o2w wxyz).
    Row 0 is conventionally the background (identity).  ``quaternion_from_matrix`` as in
    ``sgn_splatfacto_scene_graph.py:413`` (trace / largest-diagonal branches, real part first)."""
    R = rotations.to(torch.float64)
    M = R.shape[0]
    q = torch.zeros(M, 4, dtype=torch.float64)
    for m in range(M):
        r = R[m]
        tr = float(r[0, 0] + r[1, 1] + r[2, 2])
        if tr > 0:
            s = (tr + 1.0) ** 0.5 * 2
            q[m] = torch.tensor([0.25 * s, (r[2, 1] - r[1, 2]) / s, (r[0, 2] - r[2, 0]) / s, (r[1, 0] - r[0, 1]) / s])
        else:
            i = int(torch.argmax(torch.stack([r[0, 0], r[1, 1], r[2, 2]])))
            j, k = (i + 1) % 3, (i + 2) % 3
            s = float(r[i, i] - r[j, j] - r[k, k] + 1.0) ** 0.5 * 2
            v = [0.0, 0.0, 0.0]
            v[i] = 0.25 * s
            v[j] = float(r[j, i] + r[i, j]) / s
            v[k] = float(r[k, i] + r[i, k]) / s
            q[m] = torch.tensor([float(r[k, j] - r[j, k]) / s, v[0], v[1], v[2]])
    table = torch.cat([R.reshape(M, 9), translations.to(torch.float64).reshape(M, 3), q], dim=1)
    return table.to(torch.float32).contiguous()


_LAYOUTS: dict = {}


def object_ids_for(counts, device) -> torch.Tensor:
    """int32 [sum(counts)]: row i of the aggregated scene belongs to sub-model ``object_ids[i]`` (0 = background).
    Depends on the model sizes only (they change at densification, not per step), so it is built on the host once per
    layout — ``repeat_interleave`` with device counts would sync every step."""
    key = (str(device), tuple(int(c) for c in counts))
    if key not in _LAYOUTS:
        if len(_LAYOUTS) > 64:
            _LAYOUTS.clear()
        _LAYOUTS[key] = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32),
                                                torch.tensor([int(c) for c in counts])).to(device)
    return _LAYOUTS[key]


def cat_features_dc(parts) -> torch.Tensor:
    """``torch.cat`` of the sub-models' ``features_dc`` [n_i, F_i, 3] with every part zero-padded to the widest Fourier
    dimension (background: F = 1, objects: ``fourier_features_dim``, sgn_config.py:56,66); the padded coefficients meet
    zero weights in :func:`scene_graph_tables`' idft rows."""
    fmax = max(int(p.shape[1]) for p in parts)
    if all(int(p.shape[1]) == fmax for p in parts):
        return torch.cat(list(parts), dim=0)
    padded = [p if int(p.shape[1]) == fmax else torch.cat([p, p.new_zeros(p.shape[0], fmax - int(p.shape[1]), 3)], dim=1)
              for p in parts]
    return torch.cat(padded, dim=0)


_MEMO: dict = {}


def memo(fn, *args):
    """``fn(*args)`` remembered by value of its arguments (numpy arrays by their bytes).  For the PURE per-object host
    functions of the scene graph's aggregation — ``quaternion_from_matrix(anno.rot)`` (a numpy eigen-decomposition,
    ``sgn_splatfacto_scene_graph.py:412``) and ``IDFT(t, dim)`` (a dozen small CPU tensor ops, ``:420-433``): the training
    loop comes back to the same frames every epoch, and these cost 0.6 ms of host time per step at eight visible objects
    (profiles/r04q_*).  Same function, same arguments, same result; bounded to 65536 entries."""
    key = (fn,) + tuple(a.tobytes() if hasattr(a, "tobytes") else a for a in args)
    hit = _MEMO.get(key)
    if hit is None:
        if len(_MEMO) >= 65536:
            _MEMO.clear()
        hit = _MEMO[key] = fn(*args)
    return hit


def scene_graph_tables(counts, object_poses, object_idft, device) -> dict:
    """The small per-step tables of the fused scene-graph front end (``sgn_splatfacto_scene_graph.py:332-360`` folded into
    the kernels): ``counts`` Gaussians per visible sub-model, background first; ``object_poses`` one
    ``(rot [3,3], center [3], quat_o2w [4] wxyz)`` per visible object (host arrays: what ``anno.rot``, ``anno.center``
    and ``quaternion_from_matrix(anno.rot)`` are at ``:410-413``); ``object_idft`` per object the [F] inverse-DFT weights
    of the frame (``IDFT``, ``:420-433``) or None for a model without Fourier DC.  Returns
    ``dict(object_ids=int32 [N], poses=float32 [M,16], idft=float32 [M,Fmax])`` on ``device``; row 0 is the background
    (identity pose, weight 1 on coefficient 0)."""
    import numpy as np
    m = 1 + len(object_poses)
    assert len(counts) == m and len(object_idft) == len(object_poses)
    table = np.zeros((m, 16), dtype=np.float32)
    table[0, [0, 4, 8]] = 1.0
    table[0, 12] = 1.0
    fmax = max([1] + [int(len(w)) for w in object_idft if w is not None])
    idft = np.zeros((m, fmax), dtype=np.float32)
    idft[0, 0] = 1.0
    for k, (rot, center, quat) in enumerate(object_poses, start=1):
        table[k, :9] = np.asarray(rot, dtype=np.float64).reshape(9)
        table[k, 9:12] = np.asarray(center, dtype=np.float64).reshape(3)
        table[k, 12:16] = np.asarray(quat, dtype=np.float64).reshape(4)
        w = object_idft[k - 1]
        if w is None:
            idft[k, 0] = 1.0
        else:
            w = w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
            idft[k, :len(w)] = w.reshape(-1)
    both = torch.from_numpy(np.concatenate([table, idft], axis=1)).to(device)        # one host-to-device copy
    return dict(object_ids=object_ids_for(counts, device), poses=both[:, :16].contiguous(), idft=both[:, 16:].contiguous())


class _ProjectFused(Function):
    @staticmethod
    def forward(ctx, means, log_scales, quats_raw, object_ids, poses, viewmat, fx, fy, cx, cy, img_height,
                img_width, block_width, clip_thresh, glob_scale):
        dev = L.require_device(means, log_scales, quats_raw, viewmat, object_ids, poses)
        n = means.shape[0]
        if n < 1 or means.shape[-1] != 3:
            raise ValueError(f"Invalid shape for means3d: {means.shape}")
        means_c, ls_c, q_c = _f32c(means), _f32c(log_scales), _f32c(quats_raw)
        vm = _f32c(viewmat).reshape(-1)[:12].contiguous()
        oid = None if object_ids is None else object_ids.detach().to(torch.int32).contiguous()
        pos = None if poses is None else _f32c(poses)
        f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
        cov3d, xys, depths = torch.empty(n, 6, **f32), torch.empty(n, 2, **f32), torch.empty(n, **f32)
        radii, conics = torch.empty(n, **i32), torch.empty(n, 3, **f32)
        comp, nth = torch.empty(n, **f32), torch.empty(n, **i32)
        sem = _ops.semantics().flags()          # upstream-variant semantics (ops.upstream_variant), default 0
        L.check(L.load().sgn_project_fwd_fused(
            n, L.ptr(means_c), L.ptr(ls_c), float(glob_scale), L.ptr(q_c), L.ptr(oid), L.ptr(pos), L.ptr(vm),
            float(fx), float(fy), float(cx), float(cy), int(img_height), int(img_width), int(block_width),
            float(clip_thresh), L.ptr(cov3d), L.ptr(xys), L.ptr(depths), L.ptr(radii), L.ptr(conics), L.ptr(comp),
            L.ptr(nth), sem, L.stream_ptr()), "sgn_project_fwd_fused")
        ctx.consts = (float(glob_scale), float(fx), float(fy))
        ctx.sem, ctx.img_hw = sem, (int(img_height), int(img_width))     # the backward runs with the call's semantics
        ctx.has_obj = oid is not None
        ctx.arena_leaves = _ops._arena_leaves(means, log_scales, quats_raw)
        saved = [means_c, ls_c, q_c, vm, cov3d, radii, conics, comp]
        if ctx.has_obj:
            saved += [oid, pos]
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(radii, nth)
        ctx.set_materialize_grads(False)     # unused outputs arrive as None (ops._project_forward has the note)
        return xys, depths, radii, conics, comp, nth, cov3d

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_comp, v_nth, v_cov3d):
        saved = ctx.saved_tensors
        means, ls, q, vm, cov3d, radii, conics, comp = saved[:8]
        oid, pos = (saved[8], saved[9]) if ctx.has_obj else (None, None)
        gs, fx, fy = ctx.consts
        n, dev = means.shape[0], means.device
        f32 = dict(dtype=torch.float32, device=dev)
        v_xys = _f32c(v_xys) if v_xys is not None else torch.zeros(n, 2, **f32)
        v_depths = _f32c(v_depths) if v_depths is not None else None      # NULL = zeros inside the kernel
        v_conics = _f32c(v_conics) if v_conics is not None else torch.zeros(n, 3, **f32)
        v_comp = _f32c(v_comp) if v_comp is not None else None
        al = getattr(ctx, "arena_leaves", None)       # (the DP bucket's slices, ops._grad_arena)
        v_m, v_s, v_q = (_ops._leaf_grad(al, 0, (n, 3), f32), _ops._leaf_grad(al, 1, (n, 3), f32),
                         _ops._leaf_grad(al, 2, (n, 4), f32))
        L.check(L.load().sgn_project_bwd_fused(
            n, L.ptr(means), L.ptr(ls), gs, L.ptr(q), L.ptr(oid), L.ptr(pos), L.ptr(vm), fx, fy, L.ptr(cov3d),
            L.ptr(radii), L.ptr(conics), L.ptr(comp), L.ptr(v_xys), L.ptr(v_depths), L.ptr(v_conics), L.ptr(v_comp),
            L.ptr(v_m), L.ptr(v_s), L.ptr(v_q), ctx.sem, ctx.img_hw[0], ctx.img_hw[1], L.stream_ptr()),
            "sgn_project_bwd_fused")
        return (v_m, v_s, v_q) + (None,) * 12


def project_gaussians_fused(means, log_scales, quats_raw, viewmat, fx, fy, cx, cy, img_height, img_width,
                            block_width, object_ids: Optional[torch.Tensor] = None,
                            poses: Optional[torch.Tensor] = None, clip_thresh: float = 0.01,
                            glob_scale: float = 1.0):
    """``project_gaussians(R_o m + t_o, exp(log_scales), glob_scale, normalize(q_o2w (x) q_raw), ...)`` in one
    kernel; gradients w.r.t. the local means, log-scales and raw quaternions.  Same 7-tuple as upstream."""
    assert block_width > 1 and block_width <= 16, "block_width must be between 2 and 16"
    assert (object_ids is None) == (poses is None), "object_ids and poses go together"
    return _ProjectFused.apply(means.contiguous(), log_scales.contiguous(), quats_raw.contiguous(), object_ids, poses,
                               viewmat.contiguous(), fx, fy, cx, cy, img_height, img_width, block_width, clip_thresh,
                               glob_scale)


# hook for sgn_rast.dp.SHGradExchange (see ops._sh_exchange): `claims_leaves` at forward time, `tap_fused` (means,
# cam_pos, effective colour gradient, degree, K) at backward time
_sh_exchange = None


_ONES: dict = {}


def _ones_row(F: int, dev) -> torch.Tensor:
    """[1,F] ones (the idft row of a model without Fourier DC), made once per device instead of a fill per call."""
    key = (F, str(dev))
    if key not in _ONES:
        _ONES[key] = torch.ones(1, F, dtype=torch.float32, device=dev)
    return _ONES[key]


class _SHFused(Function):
    @staticmethod
    def forward(ctx, degree, means, cam_pos, features_dc, features_rest, object_ids, idft, poses, post, claimed=False):
        dev = L.require_device(means, cam_pos, features_dc, features_rest, object_ids, idft, poses)
        ctx.claimed = bool(claimed)
        n, F = features_dc.shape[0], features_dc.shape[1]
        k = 1 + (features_rest.shape[1] if features_rest is not None else 0)
        means_c, cam_c, dc_c = _f32c(means), _f32c(cam_pos).reshape(-1)[:3].contiguous(), _f32c(features_dc)
        rest_c = _f32c(features_rest) if features_rest is not None else None
        oid = None if object_ids is None else object_ids.detach().to(torch.int32).contiguous()
        idft_c = _f32c(idft).reshape(-1, F) if idft is not None else _ones_row(F, dev)
        pos = _f32c(poses) if (poses is not None and oid is not None) else None
        colors = torch.empty(n, 3, dtype=torch.float32, device=dev)
        L.check(L.load().sgn_sh_fwd_fused(n, k, int(degree), L.ptr(means_c), L.ptr(cam_c), L.ptr(dc_c), F,
                                          L.ptr(rest_c), L.ptr(oid), L.ptr(idft_c), L.ptr(pos), int(bool(post)), L.ptr(colors),
                                          L.stream_ptr()), "sgn_sh_fwd_fused")
        ctx.meta = (int(degree), k, F, int(bool(post)), features_rest is not None)
        ctx.has_obj, ctx.has_pose = oid is not None, pos is not None
        ctx.arena_leaves = _ops._arena_leaves(features_dc)
        saved = [means_c, cam_c, idft_c, colors]
        if ctx.has_obj:
            saved.append(oid)
        if ctx.has_pose:
            saved.append(pos)
        ctx.save_for_backward(*saved)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        saved = ctx.saved_tensors
        means, cam, idft, colors = saved[:4]
        oid = saved[4] if ctx.has_obj else None
        pos = saved[5] if ctx.has_pose else None
        degree, k, F, post, has_rest = ctx.meta
        n, dev = means.shape[0], means.device
        if _sh_exchange is not None:
            v_eff = _f32c(v_colors)
            if post and ctx.claimed:
                v_eff = v_eff * (colors > 0)
            if _sh_exchange.tap_fused(means, cam, v_eff, degree, k, ctx.claimed):
                return (None,) * 10      # the data-parallel exchange rebuilds the (summed) gradient itself
        v_dc = _ops._leaf_grad(getattr(ctx, "arena_leaves", None), 0, (n, F, 3), dict(dtype=torch.float32, device=dev))
        v_rest = torch.empty(n, k - 1, 3, dtype=torch.float32, device=dev) if has_rest else None
        L.check(L.load().sgn_sh_bwd_fused(n, k, degree, L.ptr(means), L.ptr(cam), F, L.ptr(oid), L.ptr(idft), L.ptr(pos), post,
                                          L.ptr(colors), L.ptr(_f32c(v_colors)), L.ptr(v_dc), L.ptr(v_rest),
                                          L.stream_ptr()), "sgn_sh_bwd_fused")
        return None, None, None, v_dc, v_rest, None, None, None, None, None


SH_MAX_PARTS = 32


class _SHFusedParts(Function):
    """spherical_harmonics_fused over UN-concatenated sub-models: ``parts`` = dc_0..dc_{m-1}, rest_0..rest_{m-1}."""
    @staticmethod
    def forward(ctx, degree, means, cam_pos, idft, poses, post, m, *parts):
        import ctypes as C
        dcs, rests = parts[:m], parts[m:]
        dev = L.require_device(means, cam_pos, idft, poses, *parts)
        k = 1 + rests[0].shape[1]
        rows = [int(d.shape[0]) for d in dcs]
        Fs = [int(d.shape[1]) for d in dcs]
        n = sum(rows)
        assert means.shape[0] == n and all(r.shape[0] == c and r.shape[1] == k - 1 for r, c in zip(rests, rows))
        means_c, cam_c = _f32c(means), _f32c(cam_pos).reshape(-1)[:3].contiguous()
        idft_c = _f32c(idft)
        assert idft_c.dim() == 2 and idft_c.shape[0] == m and idft_c.shape[1] >= max(Fs)
        pos = _f32c(poses) if poses is not None else None
        dcs_c, rests_c = [_f32c(d) for d in dcs], [_f32c(r) for r in rests]
        colors = torch.empty(n, 3, dtype=torch.float32, device=dev)
        i32s, ptrs = C.c_int32 * m, C.c_void_p * m
        ctx.rows_a, ctx.F_a = i32s(*rows), i32s(*Fs)
        L.check(L.load().sgn_sh_fwd_parts(m, ctx.rows_a, ctx.F_a, ptrs(*[t.data_ptr() for t in dcs_c]),
                                          ptrs(*[t.data_ptr() for t in rests_c]), k, int(degree), L.ptr(means_c),
                                          L.ptr(cam_c), L.ptr(idft_c), int(idft_c.shape[1]), L.ptr(pos), int(bool(post)),
                                          L.ptr(colors), L.stream_ptr()), "sgn_sh_fwd_parts")
        ctx.meta = (int(degree), k, m, int(bool(post)), [tuple(d.shape) for d in dcs], [tuple(r.shape) for r in rests])
        ctx.has_pose = pos is not None
        ctx.save_for_backward(means_c, cam_c, idft_c, colors, *([pos] if pos is not None else []))
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        import ctypes as C
        degree, k, m, post, dc_shapes, rest_shapes = ctx.meta
        if v_colors is None:
            return (None,) * (7 + 2 * m)
        saved = ctx.saved_tensors
        means, cam, idft, colors = saved[:4]
        pos = saved[4] if ctx.has_pose else None
        f32 = dict(dtype=torch.float32, device=means.device)
        v_dc = [torch.empty(sh, **f32) for sh in dc_shapes]
        v_rest = [torch.empty(sh, **f32) for sh in rest_shapes]
        ptrs = C.c_void_p * m
        L.check(L.load().sgn_sh_bwd_parts(m, ctx.rows_a, ctx.F_a, k, degree, L.ptr(means), L.ptr(cam), L.ptr(idft),
                                          int(idft.shape[1]), L.ptr(pos), post, L.ptr(colors), L.ptr(_f32c(v_colors)),
                                          ptrs(*[t.data_ptr() for t in v_dc]), ptrs(*[t.data_ptr() for t in v_rest]),
                                          L.stream_ptr()), "sgn_sh_bwd_parts")
        return (None,) * 7 + tuple(v_dc) + tuple(v_rest)


def spherical_harmonics_fused(degrees_to_use: int, means, cam_pos, features_dc, features_rest,
                              object_ids: Optional[torch.Tensor] = None, idft: Optional[torch.Tensor] = None,
                              poses: Optional[torch.Tensor] = None, post_half_clamp: bool = True) -> torch.Tensor:
    """``clamp(spherical_harmonics(n, normalize(means - cam_pos), cat(dc_eff, rest)) + 0.5, min=0)`` with
    ``dc_eff = sum_f features_dc[:, f] * idft[object, f]`` — view directions, Fourier DC, concat, SH and
    the colour post-processing in one pass; with ``poses`` the means are LOCAL and moved to the world frame
    in-kernel (no gradient to ``means``, as in the reference: ``.detach()``
    at sgn_splatfacto.py:934)."""
    if isinstance(features_dc, (list, tuple)):
        # un-concatenated sub-models (round 4): one features_dc [n_p, F_p, 3] / features_rest [n_p, K-1, 3] pair per
        # sub-model in aggregated order; part p uses pose row p and idft row p (`fused.scene_graph_tables`); no torch.cat
        m = len(features_dc)
        assert isinstance(features_rest, (list, tuple)) and len(features_rest) == m and 1 <= m
        assert idft is not None, "the per-part form needs the idft table (one row per sub-model)"
        k = 1 + features_rest[0].shape[1]
        assert k >= (degrees_to_use + 1) ** 2 and k > 1
        if _sh_exchange is not None:
            _sh_exchange._note_forward(False, degrees_to_use, k, True)      # per-rank tables: never claimed (see dp.py)
        if m <= SH_MAX_PARTS:
            return _SHFusedParts.apply(degrees_to_use, means.detach(), cam_pos, idft, poses, post_half_clamp, m,
                                       *[d.contiguous() for d in features_dc], *[r.contiguous() for r in features_rest])
        # more sub-models than the kernel's table holds: the concatenated form
        counts = [d.shape[0] for d in features_dc]
        object_ids = object_ids if object_ids is not None else object_ids_for(counts, means.device)
        features_dc, features_rest = cat_features_dc(features_dc), torch.cat(list(features_rest), dim=0)
    k = 1 + (features_rest.shape[1] if features_rest is not None else 0)
    assert k >= (degrees_to_use + 1) ** 2
    claimed = _sh_exchange is not None and _sh_exchange.claims_leaves(features_dc, features_rest, object_ids, poses,
                                                                      idft, degrees_to_use)
    return _SHFused.apply(degrees_to_use, means.detach(), cam_pos, features_dc.contiguous(),
                          None if features_rest is None else features_rest.contiguous(), object_ids, idft, poses,
                          post_half_clamp, claimed)


def rasterize_gaussians_fused(xys, depths, radii, conics, num_tiles_hit, colors, opacity_logits, img_height,
                              img_width, block_width, background=None, return_alpha=False, id_range=None,
                              depth_channel=False, group_split=None):
    """``rasterize_gaussians(..., torch.sigmoid(opacity_logits), ...)`` with the sigmoid (and its backward)
    folded into the record build / gradient unpack kernels.

    ``id_range=(lo, hi)`` renders only Gaussians ``lo <= id < hi`` of the SAME tensors — the scene graph's
    objects-only / background-only accumulation passes (``sgn_splatfacto_scene_graph.py:364-366``), which the
    reference renders from re-concatenated slices with a fresh sort each.  Passing the full geometry tensors
    keeps the one-entry binning cache hot, so the pass costs a row build and a walk of the shared depth list;
    result and gradients equal the sliced call's (the sub-list keeps its relative order).

    ``depth_channel=True`` returns ``(img, alpha, depth)`` with ``depth[H,W] = sum_g depths_g alpha_g T_g`` accumulated
    by the SAME pass (one fma per evaluated pair) — the image the reference pays a second rasterization of
    ``depths.repeat(1, 3)`` for (``sgn_splatfacto.py:982-994``); it carries no gradient (the reference's depth output
    enters no loss).

    ``group_split=s`` returns ``(img, alpha, depth or None, acc_head, acc_tail)``: besides the pass itself, the
    accumulation images of the two passes that would render only the Gaussians ``id < s`` / only those ``id >= s`` —
    what two more calls with ``id_range=(0, s)`` / ``(s, N)`` return as their alpha (bit-equal), i.e. the scene graph's
    ``background_acc`` / ``object_acc`` (``sgn_splatfacto_scene_graph.py:364-366``) — accumulated by the SAME walk
    (``sgn_raster_fwd_groups``: each entry belongs to one group and its alpha is evaluated once); each of the two that
    reaches the loss costs one alpha-only reverse walk in the backward.  Kernel options the combined walk does not
    cover fall back to the three calls."""
    assert block_width > 1 and block_width <= 16, "block_width must be between 2 and 16"
    if background is None:
        background = torch.ones(3, dtype=torch.float32, device=colors.device)
    args = (xys.contiguous(), depths.contiguous(), radii.contiguous(), conics.contiguous(), num_tiles_hit.contiguous(),
            colors.contiguous(), opacity_logits.contiguous(), img_height, img_width, block_width, background.contiguous())
    _ops._call_state.grad = torch.is_grad_enabled()
    if group_split is not None:
        assert id_range is None, "group_split splits the whole scene"
        ro, n = L.opts(), xys.shape[0]
        s = min(max(int(group_split), 0), n)
        if s == 0 or s == n:
            # one group is empty (a frame without a visible object): the other one's pass IS the main pass — its
            # accumulation is the alpha image itself, the empty group's is zero.  (Carried through the group kernel, the
            # empty group would never finish and keep every tile's walk alive to the end of its list.)
            main = _RasterizeGaussians.apply(*args, True, True, None, bool(depth_channel))
            zero = torch.zeros(img_height, img_width, dtype=torch.float32, device=xys.device)
            accs = (zero, main[1]) if s == 0 else (main[1], zero)
            return main[0], main[1], (main[2] if depth_channel else None), accs[0], accs[1]
        if group_accumulation_enabled and block_width == 16:
            out = _RasterizeGaussians.apply(*args, True, True, None, bool(depth_channel), False, None, s)
            # (the drop-in surface's own policy — option depth_channel=on — may have accumulated the channel unasked)
            return out[0], out[1], (out[2] if depth_channel else None), out[3], out[4]
        main = _RasterizeGaussians.apply(*args, True, True, None, bool(depth_channel))
        accs = [_RasterizeGaussians.apply(*args, True, True, (lo, hi), False)[1] if hi > lo
                else torch.zeros(img_height, img_width, dtype=torch.float32, device=xys.device)
                for lo, hi in ((0, s), (s, n))]
        return main[0], main[1], (main[2] if depth_channel else None), accs[0], accs[1]
    return _RasterizeGaussians.apply(*args, return_alpha, True, id_range, bool(depth_channel))


from . import config as _config
group_accumulation_enabled = bool(_config.value("group_accumulations"))   # False = the three separate passes
