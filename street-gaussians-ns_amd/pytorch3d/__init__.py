"""Import shim: ``from pytorch3d.transforms import quaternion_multiply``
(``street_gaussians_ns/sgn_splatfacto_scene_graph.py:9``, ``data/utils/bbox_optimizers.py:21``) resolves to the fused
HIP product of :mod:`sgn_rast.quat` when ``street-gaussians-ns_amd`` is on ``PYTHONPATH`` (same mechanism as the
``gsplat`` shim).  Only the names the reference imports exist."""
