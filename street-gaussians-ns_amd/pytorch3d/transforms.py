"""``pytorch3d.transforms`` surface used by the reference: ``quaternion_multiply`` (+ the two helpers it is made of)."""
from sgn_rast.quat import quaternion_multiply, quaternion_raw_multiply, standardize_quaternion  # noqa: F401

__all__ = ["quaternion_multiply", "quaternion_raw_multiply", "standardize_quaternion"]
