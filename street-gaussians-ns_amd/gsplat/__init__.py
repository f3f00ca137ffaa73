"""Drop-in ``gsplat`` 0.1.x import surface backed by libsgnrast.so (MI355X / gfx950).

Put ``street-gaussians-ns_amd/`` on ``sys.path`` and the reference's imports
(``sgn_splatfacto.py:11-14``, ``sgn_splatfacto_scene_graph.py:8``) resolve here
unchanged:

    from gsplat._torch_impl import quat_to_rotmat
    from gsplat.project_gaussians import project_gaussians
    from gsplat.rasterize import rasterize_gaussians
    from gsplat.sh import num_sh_bases, spherical_harmonics
"""
from sgn_rast.ops import (  # noqa: F401
    bin_and_sort_gaussians,
    compute_cumulative_intersects,
    get_tile_bin_edges,
    map_gaussian_to_intersects,
    num_sh_bases,
    project_gaussians,
    rasterize_gaussians,
    spherical_harmonics,
)

__version__ = "0.1.11+sgnrast"
