"""CPU: the C-ABI shared library loads, exports every symbol include/sgn_rast.h declares, the
ctypes prototype table matches the header, and the product path refuses to run without a GPU
(no CPU fallback, no oracle import)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sgn_rast.h")
PKG = os.path.join(ROOT, "street-gaussians-ns_amd")


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|void|double|const char \*)\s*\*?\s*(sgn_\w+)\s*\(([^;{]*)\)\s*;", src):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[name] = n
    return out


@pytest.fixture(scope="module")
def built_lib():
    from sgn_rast import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j8"])
    return _lib


def test_header_declares_expected_entry_points():
    fns = _header_functions()
    for required in ["sgn_project_fwd", "sgn_project_bwd", "sgn_sh_fwd", "sgn_sh_bwd", "sgn_scan_i32",
                     "sgn_map_isect", "sgn_sort_pairs", "sgn_tile_bins", "sgn_raster_fwd", "sgn_raster_bwd",
                     "sgn_bin_prepare", "sgn_bin_intersect",
                     "sgn_last_error", "sgn_version"]:
        assert required in fns


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib.LIB_PATH)
    for name in _header_functions():
        assert hasattr(lib, name), f"{name} declared in sgn_rast.h but not exported"
    assert lib.sgn_version() >= 100


def test_ctypes_table_matches_header(built_lib):
    fns = _header_functions()
    assert set(fns) == set(built_lib.SIGNATURES), set(fns) ^ set(built_lib.SIGNATURES)
    for name, nargs in fns.items():
        assert len(built_lib.SIGNATURES[name][1]) == nargs, name
    built_lib.load()  # sets restype/argtypes on every symbol


def test_no_torch_types_in_abi():
    src = open(HEADER).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code
    assert "#include <torch" not in src and "#include <ATen" not in src


def test_argument_errors_are_reported_without_a_gpu(built_lib):
    lib = built_lib.load()
    # block width out of range -> negative rc + message, before any launch
    rc = lib.sgn_project_fwd(4, None, None, 1.0, None, None, 1.0, 1.0, 0.0, 0.0, 8, 8, 32, 0.01,
                             None, None, None, None, None, None, None, 0, None)
    assert rc < 0 and b"block_width" in lib.sgn_last_error()
    rc = lib.sgn_sh_fwd(4, 7, 3, None, None, None, None)
    assert rc < 0
    assert lib.sgn_sort_workspace_bytes(1 << 20) > (1 << 20) * 12
    import ctypes
    o = built_lib.RasterOpts()
    lib.sgn_raster_default_opts(ctypes.byref(o))                  # options travel with each call: no setters
    assert (o.exact_exp, o.reduce_mode, o.adapt_fwd, o.adapt_bwd, o.batch_fwd, o.batch_bwd, o.debug_flags,
            o.ids_qmask) == (0, 1, 1024, 256, 256, 128, 0, 0)
    assert ctypes.sizeof(built_lib.RasterOpts) == 8 * 4           # the struct of include/sgn_rast.h, field for field
    assert lib.sgn_raster_workspace_bytes(5, 10, None) == 5 * 48  # one 48-byte row per Gaussian
    assert lib.sgn_raster_workspace_bytes(5, 10, ctypes.byref(o)) == 5 * 48
    assert not [n for n in built_lib.SIGNATURES if n.startswith("sgn_set_")]   # the library has no global switches
    with built_lib.options(exact_exp=1) as priv:                  # host-side options are context-local
        assert built_lib.opts() is priv and priv.exact_exp == 1
    assert built_lib.opts().exact_exp == 0
    assert lib.sgn_scan_workspace_bytes(5000) >= 12
    # the one-call projection: check modes are 0 / 1 / 2, the cleared flag needs a device word, the split wait a pinned one
    pf = lambda mode, stamp: lib.sgn_project_fwd_all(4, None, None, 1.0, None, None, 1.0, 1.0, 0.0, 0.0, 8, 8, 16, 0.01,
                                                      None, None, None, None, None, None, None, mode, 1e-6, None, stamp,
                                                      None, None, None, 0, 0, None, 0, None)
    assert pf(3, 0) == -3 and b"check_quats" in lib.sgn_last_error()
    assert pf(1, 0) == -1 and pf(2, 7) == -1
    assert lib.sgn_project_check_wait(None, 1, None, None) == -1 and b"NULL" in lib.sgn_last_error()
    # upstream-variant semantics are ARGUMENTS (round 6): the clamped EWA vjp needs the image size
    pb = lambda sem, h, w: lib.sgn_project_bwd(4, 1, 1, 1.0, 1, 1, 1.0, 1.0, 1, 1, 1, None, 1, None, 1, None, None, None,
                                               1, 1, 1, sem, h, w, None)
    assert pb(2, 0, 0) == -7 and b"img_h" in lib.sgn_last_error()
    from sgn_rast import ops
    assert ops.semantics().flags() == 0 and ops.semantics().alpha_clamp_bwd == 0.99       # the decided defaults
    with ops.upstream_variant(tile_bbox_add_after_cast=True, ewa_vjp_clamped=True, alpha_clamp_bwd=0.999) as v:
        assert ops.semantics() is v and v.flags() == 3 and v.alpha_clamp_bwd == 0.999
    assert ops.semantics().flags() == 0
    with pytest.raises(AttributeError):
        with ops.upstream_variant(no_such_switch=True):
            pass


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_on_cpu_tensors(built_lib):
    from sgn_rast import ops
    n = 8
    q = torch.zeros(n, 4); q[:, 0] = 1
    with pytest.raises(built_lib.SgnRastError):
        ops.project_gaussians(torch.rand(n, 3), torch.rand(n, 3), 1, q, torch.eye(4)[:3], 10., 10., 4., 4., 8, 8, 16)
    with pytest.raises(built_lib.SgnRastError):
        ops.spherical_harmonics(0, torch.rand(n, 3), torch.rand(n, 1, 3))
    with pytest.raises(built_lib.SgnRastError):
        ops.rasterize_gaussians(torch.rand(n, 2), torch.rand(n), torch.ones(n, dtype=torch.int32),
                                torch.rand(n, 3), torch.ones(n, dtype=torch.int32), torch.rand(n, 3),
                                torch.rand(n, 1), 8, 8, 16)


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), fn
                assert "libsgn_oracle" not in txt and "sgn_oracle.c\"" not in txt.replace("oracle/c/sgn_oracle.c", ""), fn


def test_gsplat_shim_import_surface():
    code = ("import sys; sys.path.insert(0, %r);"
            "from gsplat._torch_impl import quat_to_rotmat;"
            "from gsplat.project_gaussians import project_gaussians;"
            "from gsplat.rasterize import rasterize_gaussians;"
            "from gsplat.sh import num_sh_bases, spherical_harmonics;"
            "import inspect;"
            "print(list(inspect.signature(project_gaussians).parameters));"
            "print(list(inspect.signature(rasterize_gaussians).parameters))") % PKG
    out = subprocess.check_output([sys.executable, "-c", code], text=True).splitlines()
    assert out[0] == str(["means3d", "scales", "glob_scale", "quats", "viewmat", "fx", "fy", "cx", "cy",
                          "img_height", "img_width", "block_width", "clip_thresh"])
    assert out[1] == str(["xys", "depths", "radii", "conics", "num_tiles_hit", "colors", "opacity",
                          "img_height", "img_width", "block_width", "background", "return_alpha"])
