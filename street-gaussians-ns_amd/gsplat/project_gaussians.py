"""``gsplat.project_gaussians`` (reference import: sgn_splatfacto.py:12)."""
from sgn_rast.ops import _ProjectGaussians, project_gaussians  # noqa: F401
