"""Fused quaternion product for the scene graph's object->world transform (host shims).

``pytorch3d.transforms.quaternion_multiply`` as the reference calls it in ``object2world_gs``
(``street_gaussians_ns/sgn_splatfacto_scene_graph.py:416``): ``quaternion_multiply(quat_o2w, quats)`` with ``quat_o2w``
a CPU float64 4-vector (``torch.from_numpy(quaternion_from_matrix(rot))``, ``:412``) and ``quats`` the [N,4] device
parameter of an object model.  pytorch3d composes it from ~30 elementwise torch kernels forward and ~60 backward per
visible object, which left the drop-in scene-graph step launch-bound; here it is ONE kernel each way
(``csrc/quat.hip``).  Shapes the kernel does not cover (both operands on the CPU, as in the reference's bbox
optimiser; other broadcast patterns) take pytorch3d's own formulation in plain torch — that IS the upstream
implementation of this helper, not a stand-in for a kernel.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.quaternion_raw_multiply (plain torch; any broadcastable shapes / devices)."""
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return torch.stack((ow, ox, oy, oz), -1)


def standardize_quaternion(quaternions: torch.Tensor) -> torch.Tensor:
    return torch.where(quaternions[..., 0:1] < 0, -quaternions, quaternions)


class _QuatMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        # (quaternion_multiply has checked: `b` float32 [N,4] on the device; eight calls per scene-graph step, so the
        # conversions that would be no-ops are not made)
        b_c = b if b.is_contiguous() else b.contiguous()
        n = b_c.shape[0]
        out = torch.empty(n, 4, dtype=torch.float32, device=b.device)
        if a.numel() == 4:                                  # one quaternion for all rows, handed over by value
            a4 = (C.c_float * 4)(*a.reshape(4).tolist())
            a_rows = None
        else:
            a4, a_rows = None, a.detach().to(torch.float32).contiguous()
        L.check(L.load().sgn_quat_mul_fwd(n, a4, L.ptr(a_rows), L.ptr(b_c), L.ptr(out), L.stream_ptr()),
                "sgn_quat_mul_fwd")
        ctx.a4, ctx.a_shape = a4, a.shape
        ctx.save_for_backward(b_c, *([a_rows] if a_rows is not None else []))
        return out

    @staticmethod
    def backward(ctx, v_out):
        saved = ctx.saved_tensors
        b_c, a_rows = saved[0], (saved[1] if len(saved) > 1 else None)
        n = b_c.shape[0]
        v = v_out if (v_out.dtype is torch.float32 and v_out.is_contiguous()) else v_out.to(torch.float32).contiguous()
        need_a = ctx.needs_input_grad[0] and a_rows is not None
        v_b = torch.empty_like(b_c) if ctx.needs_input_grad[1] else None
        v_a = torch.empty_like(a_rows) if need_a else None
        if v_b is not None or v_a is not None:
            L.check(L.load().sgn_quat_mul_bwd(n, ctx.a4, L.ptr(a_rows), L.ptr(b_c), L.ptr(v), L.ptr(v_b), L.ptr(v_a),
                                              L.stream_ptr()), "sgn_quat_mul_bwd")
        return (v_a.reshape(ctx.a_shape) if v_a is not None else None), v_b


def quaternion_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``pytorch3d.transforms.quaternion_multiply``: Hamilton product of rotations, real part first, result with a
    non-negative real part."""
    # The kernel path covers exactly the reference's call shape: `b` a float32 [N,4] device tensor, `a`
    # either ONE quaternion that lives on the host (read by value: no device read-back; pytorch3d unbinds it into
    # 0-dim components, which do not promote the float32 rows of `b`, so float32 is upstream's result dtype too) or a
    # float32 device tensor of b's shape.  Everything else — float64 `b`, a single quaternion on the device (reading
    # it would be a host sync), other broadcasts — takes pytorch3d's own formulation in plain torch.
    one_host_quat = a.numel() == 4 and not a.is_cuda and not a.requires_grad
    same_shape = a.is_cuda and a.shape == b.shape and a.dtype == torch.float32
    kernel_ok = (b.is_cuda and b.dtype == torch.float32 and b.dim() == 2 and b.shape[-1] == 4 and a.shape[-1] == 4
                 and (one_host_quat or same_shape))
    if not kernel_ok:
        return standardize_quaternion(quaternion_raw_multiply(a, b))
    return _QuatMul.apply(a, b)
