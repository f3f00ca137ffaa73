"""Call-trace recorder for the gsplat operator surface (test infrastructure).

Wraps `project_gaussians` / `spherical_harmonics` / `rasterize_gaussians` of an operator namespace and records, per
call, what crosses the boundary: argument shapes, dtypes, contiguity, `requires_grad`, scalar values, and the
PROVENANCE of every tensor argument relative to earlier outputs of traced calls — the very tensor (`alias`), a view
into its storage (`view`, with the element offset) or unrelated memory (`fresh` + a first-seen
number, e.g. the `torch.cat` copies the scene graph makes of its per-model splits; the same copy handed to two calls
carries the same number).  Two runs with equal traces hand the library the same sequence of calls with the
same aliasing structure, which is what the binning cache and the drop-in fast paths key on.

Used three ways: (1) on the reference's own model files run literally on the CPU oracle (`tests/refhost.py`),
(2) on the call-site replay `sgn_rast.step` on the CPU oracle — (1) == (2) is asserted in
`tests/test_reference_literal.py` and frozen in `tests/golden/calltrace_*.json`, (3) on the replay running on the HIP
ops on the GPU box (where /root/reference does not exist), against the frozen trace.
"""
from __future__ import annotations

import json
from types import SimpleNamespace

import torch

OPS = ("project_gaussians", "spherical_harmonics", "rasterize_gaussians")


class Tracer:
    def __init__(self):
        self.calls = []
        self._outs = []     # (call index, out index, tensor) — kept alive so addresses are never recycled
        self._fresh = []    # caller-made tensors in first-seen order: the same one passed twice gets the same number

    def _prov(self, t: torch.Tensor):
        for ci, oi, o in self._outs:
            if t is o or (t.data_ptr() == o.data_ptr() and t.shape == o.shape and t.stride() == o.stride()
                          and t.dtype == o.dtype):
                return ["alias", ci, oi]
        for ci, oi, o in self._outs:
            lo = o.data_ptr()
            hi = lo + o.numel() * o.element_size()
            if o.numel() and lo <= t.data_ptr() < hi and t.dtype == o.dtype:
                return ["view", ci, oi, (t.data_ptr() - lo) // o.element_size()]
        for k, o in enumerate(self._fresh):
            if t is o or (t.data_ptr() == o.data_ptr() and t.shape == o.shape and t.stride() == o.stride()
                          and t.dtype == o.dtype and t.numel() > 0):
                return ["fresh", k]
        self._fresh.append(t)
        return ["fresh", len(self._fresh) - 1]

    def _desc(self, a):
        if torch.is_tensor(a):
            return {"shape": list(a.shape), "dtype": str(a.dtype).replace("torch.", ""),
                    "contiguous": bool(a.is_contiguous()), "requires_grad": bool(a.requires_grad),
                    "prov": self._prov(a)}
        if isinstance(a, bool) or a is None:
            return a
        if isinstance(a, int):
            return int(a)
        if isinstance(a, float):
            return round(float(a), 6)
        return repr(a)

    def wrap(self, name, fn):
        def traced(*args, **kwargs):
            rec = {"op": name, "args": [self._desc(a) for a in args],
                   "kwargs": {k: self._desc(v) for k, v in sorted(kwargs.items())}}
            ci = len(self.calls)
            self.calls.append(rec)
            out = fn(*args, **kwargs)
            outs = out if isinstance(out, tuple) else (out,)
            for oi, o in enumerate(outs):
                if torch.is_tensor(o):
                    self._outs.append((ci, oi, o))
            rec["n_out"] = len(outs)
            return out
        return traced

    def namespace(self, ops) -> SimpleNamespace:
        """An operator namespace like ``ops`` whose three hot-path entry points are traced."""
        ns = SimpleNamespace(**{k: getattr(ops, k) for k in dir(ops) if not k.startswith("__")})
        for name in OPS:
            setattr(ns, name, self.wrap(name, getattr(ops, name)))
        return ns

    def patch_module(self, mod) -> None:
        """Trace the names a reference module bound at import time (`from gsplat.x import y`)."""
        for name in OPS:
            if hasattr(mod, name):
                setattr(mod, name, self.wrap(name, getattr(mod, name)))

    def dumps(self) -> str:
        return json.dumps(self.calls, indent=1, sort_keys=True)


def canonical(calls) -> str:
    return json.dumps(json.loads(json.dumps(calls)), indent=1, sort_keys=True)
