"""``nvdiffrast.torch`` surface used by the reference's EnvLight: ``texture`` in linear / cube mode only."""
from sgn_rast.sky import texture  # noqa: F401

__all__ = ["texture"]
