"""Import shim for the one nvdiffrast op the reference uses (``import nvdiffrast.torch as dr`` at
``street_gaussians_ns/sgn_splatfacto.py:8``; ``dr.texture(..., filter_mode='linear', boundary_mode='cube')`` at
``:145``).  See :mod:`sgn_rast.sky`."""
