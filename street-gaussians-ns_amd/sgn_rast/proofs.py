"""Graph proofs: reading the autograd graph behind an operator's arguments (host logic, no device work).

The reference never hands the gsplat operators its parameters: it hands them EXPRESSIONS over its parameters —
``torch.exp(scales)`` (``sgn_splatfacto.py:857``), ``quats / quats.norm(dim=-1, keepdim=True)`` (``:864``),
``torch.sigmoid(opacities)`` (``:949``), ``torch.cat((features_dc, features_rest), dim=1)`` (``:858``),
``torch.clamp(rgbs + 0.5, min=0.0)`` (``:940``) — and, in the scene graph it ships (``sgn_config.py:42``), the
"parameters" are themselves ``torch.cat`` aggregates of the sub-models' parameters
(``sgn_splatfacto_scene_graph.py:355-360``), the objects' quaternions products with the box rotation (``:416``) and their
DC colour a Fourier sum (``:239-247``).  Autograd then carries every operator gradient back through those expressions:
tens of small kernels and ~100 graph nodes per step on a path that is launch-bound.

Each function here answers ONE question about a tensor's ``grad_fn`` chain — "is this provably ``exp`` of a row-wise
concatenation of float32 leaves?" — and returns what the operator needs to differentiate through the expression itself
(the leaves, the un-normalised tensor, the Fourier weights), or ``None``.  ``None`` always means "take the plain autograd
node": every unprovable shape stays exactly as correct as before, just slower.

What a proof skips: the intermediate autograd nodes between the operator and the returned tensors.  A hook or
``retain_grad()`` on the ARGUMENT itself refuses the proof (checked at call time); hooks on intermediates the caller
never holds (the concatenations inside the scene graph's getters) are not visible from the graph and would not fire —
the reference places none.

The node types and ``_saved_*`` attributes read here belong to the installed PyTorch, not to a documented API, so the
proofs are gated: :func:`enabled` accepts the PyTorch series they were tested on outright and otherwise runs a CPU
self-test of every matcher once (``selftest``) and switches all proofs off, with ONE logged notice, if any expression of
the reference is no longer recognised.  ``tests/test_host_logic.py`` pins the matching on the installed version.
"""
from __future__ import annotations

import logging
import os
from typing import NamedTuple, Optional, Sequence, Tuple

import torch

TESTED_TORCH_SERIES = ("2.10",)          # major.minor the matchers were developed and tested on
need_device = True                       # tests clear it to run the matchers on CPU tensors
_log = logging.getLogger("sgn_rast.proofs")
_enabled: Optional[bool] = None
_MINUS_ONE_DIMS = (1, -1, (1 << 64) - 1)      # how autograd saves dim=-1 of a 2-D tensor


def _name(node) -> str:
    return type(node).__name__ if node is not None else ""


def unhooked(t: torch.Tensor) -> bool:
    """Nobody asked for this tensor's own gradient (hook / retain_grad) or wrote into it since it was made."""
    return not t._backward_hooks and not t.retains_grad and t._version == 0


def _leaf_ok(v: Optional[torch.Tensor]) -> bool:
    return (v is not None and (v.is_cuda or not need_device) and v.dtype is torch.float32 and v.is_contiguous()
            and v.requires_grad)


def leaf_of(node) -> Optional[torch.Tensor]:
    """The float32 contiguous leaf an AccumulateGrad node belongs to."""
    v = getattr(node, "variable", None)
    return v if _leaf_ok(v) else None


def leaf_rows(node, tail: Sequence[int]) -> Optional[Tuple[torch.Tensor, ...]]:
    """The leaves ``(L_0, ..., L_m)``, each ``[n_i, *tail]``, if ``node`` is the grad_fn of ``L_0`` itself
    (AccumulateGrad) or of ``torch.cat((L_0, ..., L_m), dim=0)`` (the scene graph's ``get_aggreated_variable``,
    ``sgn_splatfacto_scene_graph.py:138-146``)."""
    tail = tuple(tail)
    one = leaf_of(node)
    if one is not None:
        return (one,) if tuple(one.shape[1:]) == tail else None
    if _name(node) != "CatBackward0" or getattr(node, "_saved_dim", None) != 0:
        return None
    leaves = []
    for nxt, _nr in node.next_functions:
        v = leaf_of(nxt)
        if v is None or v.dim() != len(tail) + 1 or tuple(v.shape[1:]) != tail:
            return None
        leaves.append(v)
    return tuple(leaves) if leaves else None


def _rows(leaves: Sequence[torch.Tensor]) -> int:
    return sum(int(v.shape[0]) for v in leaves)


def exp_leaves(scales: torch.Tensor) -> Optional[Tuple[torch.Tensor, ...]]:
    """Leaves ``L`` with ``scales == torch.exp(cat(L, 0))`` (``sgn_splatfacto.py:857``)."""
    fn = scales.grad_fn
    if _name(fn) != "ExpBackward0" or not unhooked(scales) or scales.dim() != 2:
        return None
    leaves = leaf_rows(fn.next_functions[0][0], scales.shape[1:])
    return leaves if leaves is not None and _rows(leaves) == scales.shape[0] else None


def sigmoid_leaves(opacity: torch.Tensor) -> Optional[Tuple[torch.Tensor, ...]]:
    """Leaves ``L`` with ``opacity == torch.sigmoid(cat(L, 0))`` (``sgn_splatfacto.py:949``)."""
    fn = opacity.grad_fn
    if _name(fn) != "SigmoidBackward0" or not unhooked(opacity) or opacity.dim() < 1:
        return None
    leaves = leaf_rows(fn.next_functions[0][0], opacity.shape[1:])
    return leaves if leaves is not None and _rows(leaves) == opacity.shape[0] else None


def normalised_source(quats: torch.Tensor) -> Optional[torch.Tensor]:
    """``X`` with ``quats == X / X.norm(dim=-1, keepdim=True)`` (``sgn_splatfacto.py:864``): float32 contiguous [N,4],
    a leaf or not (the scene graph's ``X`` is a concatenation of quaternion products)."""
    fq = quats.grad_fn
    if _name(fq) != "DivBackward0" or not unhooked(quats) or quats.dim() != 2 or quats.shape[1] != 4:
        return None
    nq = fq.next_functions
    if len(nq) != 2 or nq[0][0] is None or _name(nq[1][0]) != "LinalgVectorNormBackward0":
        return None
    norm = nq[1][0]
    dims = tuple(getattr(norm, "_saved_dim", ()) or ())
    if getattr(norm, "_saved_ord", None) != 2 or not getattr(norm, "_saved_keepdim", False) or len(dims) != 1 \
            or int(dims[0]) not in _MINUS_ONE_DIMS:
        return None
    if norm.next_functions[0] != nq[0]:            # the norm of the very tensor that is divided (same node, same output)
        return None
    try:
        x = fq._saved_self                        # raises if X was modified in place after the division
    except RuntimeError:
        return None
    if x is None or not _leaf_ok(x) or x.shape != quats.shape:
        return None
    # the handle must BE the divided tensor: a non-leaf by its (grad_fn, output_nr), a leaf by its AccumulateGrad node
    if x.grad_fn is not None:
        if (x.grad_fn, x.output_nr) != (nq[0][0], nq[0][1]):
            return None
    elif getattr(nq[0][0], "variable", None) is not x:
        return None
    return x


def clamp_pre(colors: torch.Tensor) -> Optional[torch.Tensor]:
    """``pre`` with ``colors == torch.clamp(pre, min=0.0)``, no upper bound (``sgn_splatfacto.py:940``)."""
    fn = colors.grad_fn
    if _name(fn) != "ClampBackward1" or not unhooked(colors):
        return None
    if getattr(fn, "_saved_max", 0) is not None or getattr(fn, "_saved_min", None) != 0:
        return None
    try:
        pre = fn._saved_self
    except RuntimeError:
        return None
    ok = (pre.shape == colors.shape and pre.dtype is torch.float32 and (pre.is_cuda or not need_device)
          and pre.is_contiguous() and pre.requires_grad)
    return pre if ok else None


class DcPart(NamedTuple):
    leaf: torch.Tensor                    # [n, 1, 3] plain, or [n, F, 3] Fourier
    weights: Optional[torch.Tensor]       # None, or the [F] idft row the Fourier sum multiplied with (no gradient)


class ShSource(NamedTuple):
    """``coeffs == cat((DC, REST), dim=1)`` with ``REST = cat(rest, 0)`` and ``DC = cat(parts, 0)``, a part being a leaf
    ``[n,1,3]`` or a Fourier sum ``sum(leaf[n,F,3] * w[F,1], dim=1, keepdim=True)``."""
    dc: Tuple[DcPart, ...]
    rest: Tuple[torch.Tensor, ...]


def _fourier_part(node) -> Optional[DcPart]:
    """``sum(leaf * w[..., None], dim=1, keepdim=True)`` (``get_fourier_features``, scene_graph.py:239-247)."""
    if _name(node) != "SumBackward1" or not getattr(node, "_saved_keepdim", False):
        return None
    dims = tuple(getattr(node, "_saved_dim", ()) or ())
    if len(dims) != 1 or int(dims[0]) not in (1, -2, (1 << 64) - 2):
        return None
    mul = node.next_functions[0][0]
    if _name(mul) != "MulBackward0" or len(mul.next_functions) != 2:
        return None
    (a, _na), (b, _nb) = mul.next_functions
    leaf = leaf_of(a)
    if leaf is None or b is not None or leaf.dim() != 3 or leaf.shape[2] != 3:      # the weights carry no gradient
        return None
    try:
        w = mul._saved_other
    except RuntimeError:
        return None
    if w is None or w.requires_grad or w.dtype is not torch.float32 or w.device != leaf.device:
        return None
    f = leaf.shape[1]
    if tuple(w.shape) not in ((f, 1), (1, f, 1)) or not w.is_contiguous():
        return None
    return DcPart(leaf, w.reshape(f))


def _dc_parts(node) -> Optional[Tuple[DcPart, ...]]:
    one = leaf_of(node)
    if one is not None:
        return (DcPart(one, None),) if tuple(one.shape[1:]) == (1, 3) else None
    single = _fourier_part(node)
    if single is not None:
        return (single,)
    if _name(node) != "CatBackward0" or getattr(node, "_saved_dim", None) != 0:
        return None
    parts = []
    for nxt, _nr in node.next_functions:
        v = leaf_of(nxt)
        if v is not None:
            if v.dim() != 3 or tuple(v.shape[1:]) != (1, 3):
                return None
            parts.append(DcPart(v, None))
            continue
        fp = _fourier_part(nxt)
        if fp is None:
            return None
        parts.append(fp)
    return tuple(parts) if parts else None


def sh_source(coeffs: torch.Tensor) -> Optional[ShSource]:
    """The leaves behind ``coeffs`` if it is provably ``torch.cat((DC, REST), dim=1)`` (``sgn_splatfacto.py:858``,
    ``sgn_splatfacto_scene_graph.py:280``) of the shapes :class:`ShSource` describes."""
    fn = coeffs.grad_fn
    if _name(fn) != "CatBackward0" or getattr(fn, "_saved_dim", None) != 1 or coeffs.dim() != 3:
        return None
    if not unhooked(coeffs) or coeffs.dtype is not torch.float32 or coeffs.shape[2] != 3:
        return None
    nxt = fn.next_functions
    n, k = coeffs.shape[0], coeffs.shape[1]
    if len(nxt) != 2 or k < 2:
        return None
    dc = _dc_parts(nxt[0][0])
    rest = leaf_rows(nxt[1][0], (k - 1, 3))
    if dc is None or rest is None or _rows(rest) != n or sum(int(p.leaf.shape[0]) for p in dc) != n:
        return None
    return ShSource(dc, rest)


def repeated_depths(colors: torch.Tensor, depths: torch.Tensor) -> bool:
    """``colors`` is literally ``depths[:, None].repeat(1, 3)`` of this very ``depths`` (``sgn_splatfacto.py:988``)."""
    fn, src = colors.grad_fn, depths.grad_fn
    if src is None or _name(fn) != "RepeatBackward0":
        return False
    if tuple(getattr(fn, "_saved_repeats", ())) != (1, 3):
        return False
    inner = fn.next_functions[0][0]
    if _name(inner) != "UnsqueezeBackward0" or getattr(inner, "_saved_dim", None) not in (1, -1):
        return False
    node, nr = inner.next_functions[0]
    return node is src and nr == depths.output_nr


class SplitCat(NamedTuple):
    """A tensor that is ``torch.cat`` (dim 0) of ``count`` consecutive parts, starting at part ``first``, of ONE
    ``torch.split`` (dim 0) with part sizes ``sizes``."""
    node: object
    first: int
    count: int
    sizes: Tuple[int, ...]

    @property
    def whole(self) -> bool:
        return self.first == 0 and self.count == len(self.sizes)

    @property
    def rows(self) -> Tuple[int, int]:
        lo = sum(self.sizes[:self.first])
        return lo, lo + sum(self.sizes[self.first:self.first + self.count])


def split_cat(t: torch.Tensor) -> Optional[SplitCat]:
    """The scene graph's setters split every projection output per sub-model and its getters concatenate the parts again
    (``sgn_splatfacto_scene_graph.py:153-215``): all the parts for the main pass, a run of consecutive ones for a
    sub-model pass (``aggregate_submodel_var``, ``:249-253``).  Either way the tensor the rasterizer receives is a COPY
    of rows of the projection's output, and its graph says which."""
    fn = t.grad_fn
    if _name(fn) != "CatBackward0" or getattr(fn, "_saved_dim", None) != 0 or t._version != 0:
        return None
    edges = fn.next_functions
    if not edges:
        return None
    node, first = edges[0]
    if _name(node) != "SplitWithSizesBackward0" or getattr(node, "_saved_dim", None) != 0:
        return None
    sizes = tuple(int(x) for x in node._saved_split_sizes)
    if first + len(edges) > len(sizes) or tuple(edges) != tuple((node, first + i) for i in range(len(edges))):
        return None
    sc = SplitCat(node, first, len(edges), sizes)
    lo, hi = sc.rows
    return sc if hi - lo == t.shape[0] else None


def window_of_split(sub: Optional[SplitCat], full: Optional[SplitCat]) -> Optional[Tuple[int, int]]:
    """Row window ``(lo, hi)`` of ``full`` (all parts of a split) that ``sub`` (some consecutive parts of the SAME split)
    is a copy of."""
    if sub is None or full is None or not full.whole or sub.node is not full.node:
        return None
    return sub.rows


def window_of_leaves(sub_leaves: Sequence[torch.Tensor], full_leaves: Sequence[torch.Tensor]) -> Optional[Tuple[int, int]]:
    """Row window of ``cat(full_leaves)`` that ``cat(sub_leaves)`` equals: ``sub_leaves`` a run of consecutive entries."""
    if not sub_leaves or len(sub_leaves) > len(full_leaves):
        return None
    for first in range(len(full_leaves) - len(sub_leaves) + 1):
        if all(a is b for a, b in zip(sub_leaves, full_leaves[first:first + len(sub_leaves)])):
            lo = _rows(full_leaves[:first])
            return lo, lo + _rows(sub_leaves)
    return None


# ----------------------------------------------------------------------------------------------------------- gating
def selftest() -> bool:
    """Every expression of the reference, built on small CPU tensors, must be recognised — and near misses refused."""
    global need_device
    old, need_device = need_device, False
    try:
        with torch.enable_grad():
            n, k, f = 6, 4, 5
            mk = lambda *s: torch.randn(*s).requires_grad_(True)
            ls, ls2, rq, lo, lo2 = mk(n, 3), mk(2, 3), mk(n, 4), mk(n, 1), mk(2, 1)
            dc, rest, dcf, rest2 = mk(n, 1, 3), mk(n, k - 1, 3), mk(2, f, 3), mk(2, k - 1, 3)
            w = torch.randn(1, f)
            ok = exp_leaves(torch.exp(ls)) == (ls,)
            got = exp_leaves(torch.exp(torch.cat([ls, ls2], 0)))
            ok &= got is not None and got[0] is ls and got[1] is ls2
            ok &= exp_leaves(torch.exp(ls) * 1.0) is None and exp_leaves(torch.exp(ls * 1.0)) is None
            got = sigmoid_leaves(torch.sigmoid(torch.concat([lo, lo2], dim=0)))
            ok &= got is not None and got[0] is lo and got[1] is lo2
            ok &= sigmoid_leaves(torch.sigmoid(lo * 1.0)) is None
            ok &= normalised_source(rq / rq.norm(dim=-1, keepdim=True)) is rq
            x = torch.cat([rq, rq * 2.0], 0)
            ok &= normalised_source(x / x.norm(dim=-1, keepdim=True)) is x
            ok &= normalised_source(rq / rq.norm(dim=-1)[:, None]) is None
            ok &= normalised_source(rq / (rq * 1.0).norm(dim=-1, keepdim=True)) is None
            ok &= normalised_source(rq / rq.norm(p=1, dim=-1, keepdim=True)) is None
            sh = mk(n, 3) * 1.0
            ok &= clamp_pre(torch.clamp(sh + 0.5, min=0.0)) is not None
            ok &= clamp_pre(torch.clamp(sh + 0.5, min=0.0, max=1.0)) is None
            src = sh_source(torch.cat((dc, rest), dim=1))
            ok &= src is not None and src.dc[0].leaf is dc and src.dc[0].weights is None and src.rest == (rest,)
            four = torch.sum(dcf * w[0][..., None], dim=1, keepdim=True)
            src = sh_source(torch.cat((torch.cat([dc, four], 0), torch.cat([rest, rest2], 0)), dim=1))
            ok &= (src is not None and len(src.dc) == 2 and src.dc[1].leaf is dcf
                   and src.dc[1].weights is not None and torch.equal(src.dc[1].weights, w[0]))
            ok &= sh_source(torch.cat((dc * 1.0, rest), dim=1)) is None
            d = mk(n)
            ok &= repeated_depths((d * 1.0)[:, None].repeat(1, 3), d * 1.0) is False
            dd = d * 1.0
            ok &= repeated_depths(dd[:, None].repeat(1, 3), dd)
            xy = mk(n + 2, 2) * 1.0
            parts = torch.split(xy, [n, 1, 1])
            full = split_cat(torch.concat(parts, dim=0))
            ok &= full is not None and full.whole
            ok &= window_of_split(split_cat(torch.cat(parts[1:], 0)), full) == (n, n + 2)
            ok &= window_of_split(split_cat(torch.cat([parts[0]], 0)), full) == (0, n)
            ok &= split_cat(torch.cat([parts[0], parts[2]], 0)) is None
            return bool(ok)
    except Exception:                                        # an attribute vanished: same answer as "not recognised"
        return False
    finally:
        need_device = old


def enabled() -> bool:
    """Whether the graph proofs may be used with the installed PyTorch (decided once per process)."""
    global _enabled
    if _enabled is None:
        series = ".".join(torch.__version__.split(".")[:2])
        if series in TESTED_TORCH_SERIES or selftest():
            _enabled = True
        else:
            _enabled = False
            _log.warning("sgn_rast: the autograd graph of PyTorch %s is not recognised by the graph proofs (tested on %s); "
                         "the operators take their plain autograd path (same results, more small kernels).",
                         torch.__version__, ", ".join(TESTED_TORCH_SERIES))
    return _enabled
