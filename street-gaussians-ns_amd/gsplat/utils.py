"""``gsplat.utils`` binning helpers (called inside rasterize_gaussians upstream)."""
from sgn_rast.ops import (  # noqa: F401
    bin_and_sort_gaussians,
    compute_cumulative_intersects,
    get_tile_bin_edges,
    map_gaussian_to_intersects,
)
