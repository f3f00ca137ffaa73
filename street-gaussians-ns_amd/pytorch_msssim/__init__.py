"""Import shim: ``from pytorch_msssim import SSIM`` (``street_gaussians_ns/sgn_splatfacto.py:15``) resolves to the
fused HIP SSIM of :mod:`sgn_rast.loss` when ``street-gaussians-ns_amd`` is on ``PYTHONPATH`` (same mechanism as the
``gsplat`` shim).  Only the configuration the reference builds (``SSIM(data_range=1.0, size_average=True,
channel=3)``, ``:330``) is implemented; anything else raises ``NotImplementedError``."""
from sgn_rast.loss import SSIM  # noqa: F401

__all__ = ["SSIM"]
