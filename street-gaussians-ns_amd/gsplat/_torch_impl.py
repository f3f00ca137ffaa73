"""``gsplat._torch_impl`` — only ``quat_to_rotmat`` is used by the reference (sgn_splatfacto.py:11,685)."""
from sgn_rast.ops import quat_to_rotmat  # noqa: F401
