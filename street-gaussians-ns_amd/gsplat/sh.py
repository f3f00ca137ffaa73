"""``gsplat.sh`` (reference imports: sgn_splatfacto.py:14, sgn_splatfacto_scene_graph.py:8)."""
from sgn_rast.ops import _SphericalHarmonics, deg_from_sh, num_sh_bases, spherical_harmonics  # noqa: F401
