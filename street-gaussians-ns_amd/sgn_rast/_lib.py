"""ctypes binding of libsgnrast.so (the C ABI declared in include/sgn_rast.h).

There is NO fallback: if the HIP library is missing or a tensor is not on a ROCm
device the call raises.  (The CPU oracle lives in ``oracle/`` and is test
infrastructure only; nothing here imports it.)
"""
from __future__ import annotations

import contextlib
import contextvars
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGN_RAST_LIB") or os.path.join(_HERE, "libsgnrast.so")  # override: debugging builds

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/sgn_rast.h line by line
SIGNATURES = {
    "sgn_version": (_i, []),
    "sgn_last_error": (C.c_char_p, []),
    "sgn_raster_default_opts": (None, [_vp]),
    "sgn_timing_enable": (None, [_i]),
    "sgn_timing_get": (_i, [_i, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "sgn_timing_host_wait_us": (C.c_double, [_i, _vp]),
    "sgn_project_fwd": (_i, [_i, _vp, _vp, _f, _vp, _vp, _f, _f, _f, _f, _i, _i, _i, _f,
                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sgn_project_bwd": (_i, [_i, _vp, _vp, _f, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                             _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sgn_project_fwd_fused": (_i, [_i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _i, _i, _i, _f,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sgn_project_bwd_fused": (_i, [_i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sgn_project_bwd_act": (_i, [_i, _vp, _vp, _f, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sgn_fourier_dc_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgn_sh_fwd_fused": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "sgn_sh_bwd_fused": (_i, [_i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "sgn_sh_fwd_parts": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "sgn_sh_bwd_parts": (_i, [_i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "sgn_cube_texture_fwd": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sgn_cube_texture_bwd": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sgn_sky_fwd": (_i, [_i, _i, _f, _f, _f, _f, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "sgn_sky_bwd": (_i, [_i, _i, _f, _f, _f, _f, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "sgn_sky_blend_fwd": (_i, [_i, _i, _f, _f, _f, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgn_sky_blend_bwd": (_i, [_i, _i, _f, _f, _f, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgn_l1_ssim_workspace_bytes": (_sz, [_i, _i, _i]),
    "sgn_l1_ssim_fwd": (_i, [_i, _i, _vp, _vp, _f, _f, _f, _vp, _i, _vp, _sz, _vp]),
    "sgn_l1_ssim_bwd": (_i, [_i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "sgn_acc_losses_workspace_bytes": (_sz, [_i64]),
    "sgn_acc_losses_fwd": (_i, [_i64, _vp, _vp, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "sgn_acc_losses_bwd": (_i, [_i64, _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sgn_adam_step": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgn_densify_stats": (_i, [_i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    "sgn_check_unit_quats": (_i, [_i, _vp, _f, _vp, _vp]),
    "sgn_quat_mul_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "sgn_quat_mul_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgn_sh_bwd_multi": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "sgn_sh_fwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "sgn_sh_bwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "sgn_scan_workspace_bytes": (_sz, [_i]),
    "sgn_scan_i32": (_i, [_i, _vp, _vp, _vp, _sz, _vp]),
    "sgn_map_isect": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "sgn_sort_workspace_bytes": (_sz, [_i64]),
    "sgn_sort_pairs": (_i, [_i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "sgn_sort_selftest_workspace_bytes": (_sz, []),
    "sgn_sort_selftest": (_i, [_vp, _sz, _i, _vp, _vp]),
    "sgn_tile_bins": (_i, [_i64, _vp, _i, _vp, _vp]),
    "sgn_bin_prepare_workspace_bytes": (_sz, [_i]),
    "sgn_bin_prepare": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _sz, _i, _i, _vp]),
    "sgn_depth_rank_workspace_bytes": (_sz, [_i]),
    "sgn_depth_rank": (_i, [_i, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "sgn_bin_intersect_workspace_bytes": (_sz, [_i64]),
    "sgn_bin_intersect": (_i, [_i, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _sz, _vp, _i, _vp]),
    "sgn_list_window_workspace_bytes": (_sz, [_i]),
    "sgn_list_window": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "sgn_mark_walked": (_i, [_i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sgn_rows_pack": (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "sgn_rows_scatter": (_i, [_i, _vp, _i, _i, _vp, _vp, _f, _i, _vp, _vp]),
    "sgn_rows_outside": (_i, [_i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "sgn_rows_match": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgn_raster_workspace_bytes": (_sz, [_i, _i64, _vp]),
    "sgn_tile_order_scratch_bytes": (_sz, [_i]),
    "sgn_tile_order": (_i, [_i, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "sgn_raster_fwd": (_i, [_i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                            _sz, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgn_raster_fwd_groups": (_i, [_i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i,
                                   _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sgn_project_fwd_all": (_i, [_i, _vp, _vp, _f, _vp, _vp, _f, _f, _f, _f, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _i, _f, _vp, _i, _vp, _vp, _vp, _sz, _i, _vp, _i, _vp]),
    "sgn_project_check_wait": (_i, [_vp, _i, _vp, _vp]),
    "sgn_rasterize_arena_bytes": (_sz, [_i, _i64]),
    "sgn_rasterize_window_arena_bytes": (_sz, [_i]),
    "sgn_rasterize_window_all": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _i64, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "sgn_rasterize_bwd_all": (_i, [_i, _i, _i, _i, _i64, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp,
                                   _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp, _sz, _vp, _vp, _sz, _i, _vp,
                                   _vp, _vp, _vp, _i, _i]),
    "sgn_rasterize_fwd_all": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                   _i64, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "sgn_raster_build_rows": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "sgn_colors_match_depths": (_i, [_i, _vp, _vp, _vp, _vp]),
    "sgn_depth_reuse": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "sgn_raster_bwd_workspace_bytes": (_sz, [_i]),
    "sgn_raster_bwd": (_i, [_i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                            _f, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "sgn_raster_bwd_part": (_i, [_i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                                 _vp, _f, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i, _i]),
}

_lib = None


class RasterOpts(C.Structure):
    """`sgn_raster_opts` of include/sgn_rast.h: kernel-selection options handed to the raster entry points per call."""
    _fields_ = [(k, C.c_int) for k in ("exact_exp", "reduce_mode", "adapt_fwd", "adapt_bwd", "batch_fwd", "batch_bwd",
                                       "debug_flags", "ids_qmask")]

    def copy(self) -> "RasterOpts":
        out = RasterOpts()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(RasterOpts))
        return out


# The library itself is stateless; the HOST keeps the options: one process-wide default object (library defaults,
# overridable through SGN_OPTIONS, sgn_rast/config.py) and, inside `with options(...)`, a private copy that
# only the current thread / context sees (contextvars), so concurrent callers cannot race on a switch.
_OPT_FIELDS = ("reduce_mode", "batch_fwd", "batch_bwd", "adapt_fwd", "adapt_bwd", "exact_exp", "debug_flags")
_process_opts = None
_ctx_opts: contextvars.ContextVar = contextvars.ContextVar("sgn_raster_opts", default=None)


def _library_defaults() -> RasterOpts:
    from . import config
    o = RasterOpts()
    load().sgn_raster_default_opts(C.byref(o))
    for field in _OPT_FIELDS:                   # SGN_OPTIONS="batch_fwd=64,..." (sgn_rast/config.py)
        if field in config._from_env:
            setattr(o, field, int(config._from_env[field]))
    return o


def opts() -> RasterOpts:
    """The options the next raster call of this context will carry."""
    global _process_opts
    cur = _ctx_opts.get()
    if cur is not None:
        return cur
    if _process_opts is None:
        _process_opts = _library_defaults()
    return _process_opts


def opts_ptr():
    return C.byref(opts())


def set_options(**kw) -> None:
    """Change fields of the current options object (the process default, or the private copy inside `options`)."""
    o = opts()
    for k, v in kw.items():
        if k not in dict(RasterOpts._fields_):
            raise AttributeError(k)
        setattr(o, k, int(v))


def reset_options() -> None:
    """Back to the library defaults (+ SGN_* environment overrides)."""
    global _process_opts
    _process_opts = None
    _ctx_opts.set(None)


@contextlib.contextmanager
def options(**kw):
    """`with options(exact_exp=1): ...` — a private copy of the options for this thread / context only."""
    o = opts().copy()
    for k, v in kw.items():
        if k not in dict(RasterOpts._fields_):
            raise AttributeError(k)
        setattr(o, k, int(v))
    token = _ctx_opts.set(o)
    try:
        yield o
    finally:
        _ctx_opts.reset(token)


class SgnRastError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libsgnrast.so and declare every prototype; raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SgnRastError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C street-gaussians-ns_amd/csrc`). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


# ---------------------------------------------------------------------------------------------- sort ranking
# Which in-wave ranking the radix sorts use is an ARGUMENT of every sorting entry point (include/sgn_rast.h,
# `sort_rank_mode`); the library keeps no state about it.  The host's policy, PER DEVICE:
#   default              0 = the ballot-match ranking: documented ISA semantics only (north_star: bit-exact sort keys);
#   sort_rank=atomic     1 = one returning LDS atomic per key (-7 us per binning) — but only on a device that has passed
#                        `sort_selftest_under_load` (>= 1000 probe sorts while a second instance and a GEMM run on other
#                        streams); a device that fails stays on 0, with a warning;
#   sort_rank=atomic-unchecked       1 without the probe (A/B runs).   (option `sort_rank` of sgn_rast/config.py)
# `force_sort_rank` overrides it inside a `with` block (tests, bench.py's A/B line).
_SORT_RANK: dict = {}          # device index -> {"mode": 0 | 1, "probe": str}
_sort_rank_forced = None


def sort_selftest_under_load(rounds: int = 64, device=None) -> int:
    """Mismatching output pairs (0 = the atomic ranking reproduced the documented one) over `rounds` x 16 probe sorts per
    ranking on the current stream of `device`, while a second instance (own workspace, another stream) and a chain of
    GEMMs (a third stream) keep the LDS and the CUs busy."""
    lib = load()
    dev = torch.device("cuda", current_device() if device is None else device)
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        side, load_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        nbytes = int(lib.sgn_sort_selftest_workspace_bytes())
        ws = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        bad = torch.zeros(2, dtype=torch.int32, device=dev)
        a = torch.randn(2048, 2048, device=dev)
        side.wait_stream(main); load_s.wait_stream(main)
        with torch.cuda.stream(load_s):
            for _ in range(max(8, rounds // 2)):
                a = torch.mm(a, a).clamp_(-1.0, 1.0)
        for i, st in enumerate((main, side)):
            check(lib.sgn_sort_selftest(C.c_void_p(ws[i].data_ptr()), nbytes, int(rounds),
                                        C.c_void_p(bad[i:i + 1].data_ptr()), C.c_void_p(st.cuda_stream)),
                  "sgn_sort_selftest")
        main.wait_stream(side); main.wait_stream(load_s)
        for t in ws + [bad, a]:
            t.record_stream(side); t.record_stream(load_s)
        return int(bad.sum().item())


_sort_rank_request = None      # None: what sgn_rast.config says ("ballot" unless SGN_OPTIONS="sort_rank=atomic")


def _decide_sort_ranking(dev_index: int) -> dict:
    from . import config
    want = (_sort_rank_request or config.value("sort_rank")).lower()
    if want == "atomic-unchecked":
        return {"mode": 1, "probe": "forced by sort_rank=atomic-unchecked (no probe)"}
    if want != "atomic" or not torch.cuda.is_available():
        return {"mode": 0, "probe": "not needed: the documented ballot ranking is the default"}
    rounds = 64
    bad = sort_selftest_under_load(rounds, dev_index)
    if bad == 0:
        return {"mode": 1, "probe": f"passed under load on device {dev_index}: {2 * 16 * rounds} probe sorts per ranking, "
                                    "0 mismatching pairs"}
    import warnings
    warnings.warn(f"sort_rank=atomic: the probe counted {bad} mismatching pairs on device {dev_index}; staying on "
                  "the documented ballot ranking there")
    return {"mode": 0, "probe": f"FAILED ({bad} mismatching pairs) on device {dev_index}: ballot ranking"}


def sort_rank_mode() -> int:
    """The `sort_rank_mode` argument for a sort on the CURRENT device (0 ballot / 1 atomic)."""
    if _sort_rank_forced is not None:
        return _sort_rank_forced
    d = current_device() if torch.cuda.is_available() else -1
    e = _SORT_RANK.get(d)
    if e is None:
        e = _SORT_RANK[d] = _decide_sort_ranking(d)
    return e["mode"]


def sort_ranking_report() -> dict:
    """{"mode": "ballot" | "atomic", "probe": ...} of the current device (bench.py prints it)."""
    mode = sort_rank_mode()
    d = current_device() if torch.cuda.is_available() else -1
    e = _SORT_RANK.get(d) or {"probe": "forced"}
    return {"mode": "atomic" if mode else "ballot", "probe": e["probe"] if _sort_rank_forced is None else "forced (A/B)"}


class force_sort_rank:
    """`with force_sort_rank("atomic"):` — every sort issued inside uses that ranking (tests, A/B measurements)."""

    def __init__(self, mode):
        self.mode = 1 if mode in (1, "atomic") else 0

    def __enter__(self):
        global _sort_rank_forced
        self.before, _sort_rank_forced = _sort_rank_forced, self.mode
        return self

    def __exit__(self, *exc):
        global _sort_rank_forced
        _sort_rank_forced = self.before
        return False


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sgn_last_error()
        raise SgnRastError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def require_device(*tensors: torch.Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise SgnRastError(
                "sgn_rast operates on ROCm device tensors only (got a CPU tensor); there is no CPU fallback")
        dev = dev or t.device
        if t.device != dev:
            raise SgnRastError("all tensors must live on the same device")
    if dev is not None and dev.index is not None and dev.index != current_device():
        # kernels are queued on the CURRENT device's current stream (one process per GPU: bench.py / dp.py set it)
        raise SgnRastError(f"tensors live on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                           "wrap the call in torch.cuda.device(...)")
    return dev


def ptr(t):
    """Device address of a tensor for a `void *` parameter (a plain int: ctypes converts it; building a c_void_p object
    per argument was ~60 small allocations per train step), NULL for None."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def current_device() -> int:
    """`torch.cuda.current_device()` without its lazy-init bookkeeping (the library is only ever called with device
    tensors in hand, so the runtime is up): ~40 calls per train step."""
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def stream_handle() -> int:
    """The current HIP stream of the current device as an integer.  `torch.cuda.current_stream()` builds a Python
    Stream object every time (~9 us; ten calls per train step showed up as 90 us of host time under cProfile)."""
    if _raw_stream is not None:
        return _raw_stream(current_device())
    return int(torch.cuda.current_stream().cuda_stream)


def stream_ptr() -> int:
    """The current stream for a `sgn_stream_t` parameter (an int, see `ptr`)."""
    return stream_handle()


_aux_streams: dict = {}


def aux_stream(device) -> "torch.cuda.Stream":
    """A second stream of ``device`` the library may fan independent kernels out to (sgn_raster_bwd runs the two halves
    of its adaptive scheme concurrently); made once per device."""
    dev = torch.device(device)
    if dev not in _aux_streams:
        _aux_streams[dev] = torch.cuda.Stream(device=dev)
    return _aux_streams[dev]


_aux_handles: dict = {}


def aux_stream_ptr(device) -> int:
    h = _aux_handles.get(device)
    if h is None:
        h = _aux_handles[device] = int(aux_stream(device).cuda_stream)
    return h


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


TIMING_SLOTS = ["project_fwd", "project_bwd", "sh_fwd", "sh_bwd", "scan", "map_isect", "sort", "tile_bins",
                "pack_records", "raster_fwd", "raster_bwd", "unpack_grads", "sky_fwd", "sky_bwd", "loss_fwd", "loss_bwd", "adam"]


def timing_enable(on: bool) -> None:
    load().sgn_timing_enable(1 if on else 0)


def timing_report() -> dict:
    """{slot: (launches, total_ms)} from the library's HIP-event spans (synchronises them)."""
    out = {}
    for i, name in enumerate(TIMING_SLOTS):
        c, t = C.c_int(0), C.c_float(0.0)
        check(load().sgn_timing_get(i, C.byref(c), C.byref(t)), "sgn_timing_get")
        out[name] = (c.value, t.value)
    return out
