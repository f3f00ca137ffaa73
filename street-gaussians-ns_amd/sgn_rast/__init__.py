"""sgn_rast — MI355X-native differentiable Gaussian rasterizer (host side).

Only what the hot path needs: the ctypes binding of libsgnrast.so (``_lib``), the
gsplat-compatible operator surface (``ops``), the data-parallel helpers (``dp``)
and the deterministic synthetic scenes used by bench/smoke/tests (``scenes``).
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
