"""The only known answers the REFERENCE itself holds on this path: `RGB2SH` / `SH2RGB`
(`street_gaussians_ns/sgn_splatfacto.py:57-70`, C0 = 0.28209479177387814) — what `features_dc` is initialised with
(`:259-262`) — frozen into tests/golden/known_rgb2sh.npz by tests/golden/make_known_answers.py.

    spherical_harmonics(0, dirs, RGB2SH(rgb)[:, None, :]) + 0.5  ==  SH2RGB(RGB2SH(rgb))  (bit for bit)  ~=  rgb

pins the degree-0 basis constant and the `+ 0.5` of the call sites (`:940`) to the reference's own arithmetic: CPU here
(both oracles, and the frozen file against the reference where the checkout exists), HIP kernels under `-m gpu`."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
G = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(HERE, "golden", "known_rgb2sh.npz")).items()}


def _check(sh_fn, dev="cpu"):
    rgb, dirs = G["rgb"].to(dev), G["dirs"].to(dev)
    dc = G["RGB2SH_of_rgb"].to(dev)[:, None, :]
    # degree 0, K = 1: the reference's own round trip, bit for bit, and rgb to fp32 rounding
    out = sh_fn(0, dirs, dc) + 0.5
    assert torch.equal(out.cpu(), G["SH2RGB_of_RGB2SH"])
    assert float((out.cpu() - G["rgb"]).abs().max()) <= 2.4e-7
    # any sh: SH2RGB(sh) == spherical_harmonics(0, ., sh) + 0.5
    assert torch.equal((sh_fn(0, dirs, G["sh"].to(dev)[:, None, :]) + 0.5).cpu(), G["SH2RGB_of_sh"])
    # the training layout (K = 16, bands 1-3 zero) at every degree: still the same colours, whatever the direction
    coeffs = torch.cat([dc, torch.zeros(dc.shape[0], 15, 3, device=dev)], dim=1)
    for deg in (0, 1, 2, 3):
        assert torch.equal((sh_fn(deg, dirs, coeffs) + 0.5).cpu(), G["SH2RGB_of_RGB2SH"]), deg


def test_frozen_file_is_what_the_reference_computes():
    import refhost
    if not refhost.available():
        pytest.skip("no reference checkout on this machine (the frozen file stands in for it)")
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_known_answers as M
    fresh = M.compute()
    assert set(fresh) == set(G)
    for k, v in fresh.items():
        assert np.array_equal(v, G[k].numpy()), k
    # ... and the constant itself, as the scene builders of this repo use it
    from sgn_rast import scenes
    assert scenes.SH_C0 == 0.28209479177387814


def test_degree0_round_trip_torch_oracle(torch_oracle):
    _check(torch_oracle.spherical_harmonics)


def test_degree0_round_trip_c_oracle():
    import oracle_ops
    _check(oracle_ops.spherical_harmonics)


@pytest.mark.gpu
def test_degree0_round_trip_hip_kernels():
    from sgn_rast import ops
    with torch.no_grad():
        _check(ops.spherical_harmonics, dev="cuda")
    # fused front end: un-concatenated leaves, world view directions computed inside, `+ 0.5, clamp(min=0)` fused in
    from sgn_rast import fused
    means = (G["dirs"] * 3.0).cuda()
    dc = G["RGB2SH_of_rgb"].cuda()[:, None, :].contiguous()
    rest = torch.zeros(dc.shape[0], 15, 3, device="cuda")
    with torch.no_grad():
        rgbs = fused.spherical_harmonics_fused(3, means, torch.zeros(3, device="cuda"), dc, rest)
    assert torch.equal(rgbs.cpu(), torch.clamp(G["SH2RGB_of_RGB2SH"], min=0.0))
