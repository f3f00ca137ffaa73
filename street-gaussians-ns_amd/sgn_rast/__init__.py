"""sgn_rast — MI355X-native differentiable Gaussian rasterizer (host side).

Only what the hot path needs: the ctypes binding of libsgnrast.so (``_lib``), the gsplat-compatible operator
surface (``ops``), the fused front ends (``fused``: activations / rigid transform / Fourier DC / sigmoid folded into
the kernels), the callers either side of the path that SURVEY.md §8f ranks next (``sky``: nvdiffrast cube-map lookup,
``loss``: L1 + SSIM, ``optim``: multi-tensor Adam, ``densify``: per-step statistics), the data-parallel helpers
(``dp``), the call-site replay used by bench/smoke/tests (``step``) and the deterministic synthetic scenes
(``scenes``).  Sub-modules are imported on demand; none of them has a CPU fallback.
"""
from .ops import (  # noqa: F401
    bin_and_sort_gaussians,
    compute_cumulative_intersects,
    get_tile_bin_edges,
    map_gaussian_to_intersects,
    num_sh_bases,
    project_gaussians,
    quat_to_rotmat,
    rasterize_gaussians,
    set_alpha_clamp_bwd,
    spherical_harmonics,
)

__version__ = "0.1.0"
