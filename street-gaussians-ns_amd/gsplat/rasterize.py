"""``gsplat.rasterize`` (reference import: sgn_splatfacto.py:13)."""
from sgn_rast.ops import _RasterizeGaussians, rasterize_gaussians  # noqa: F401
