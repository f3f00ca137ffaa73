"""GPU (-m gpu): data-parallel pieces that can be exercised on one MI355X — the multi-view SH backward
kernel, and the full exchange path over a world-size-1 RCCL group (real NCCL calls, hooks, streams)."""
import os
import socket

import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("k,deg,R", [(16, 3, 8), (16, 1, 2), (4, 1, 3), (25, 4, 2), (1, 0, 4)])
def test_sh_bwd_multi_equals_sum_of_single_view_backwards(k, deg, R):
    from sgn_rast import _lib as L, dp
    n = 3001
    g = torch.Generator().manual_seed(k * 10 + R)
    means = (torch.randn(n, 3, generator=g) * 3).to(DEV)
    cams = (torch.randn(R, 3, generator=g)).to(DEV)
    v_all = torch.randn(R, n, 3, generator=g).to(DEV)
    dirs_all = torch.stack([means - cams[r] for r in range(R)])
    lib = L.load()
    ref = torch.zeros(n, k, 3, device=DEV)
    for r in range(R):
        one = torch.empty(n, k, 3, device=DEV)
        L.check(lib.sgn_sh_bwd(n, k, deg, L.ptr(dirs_all[r].contiguous()), L.ptr(v_all[r].contiguous()), L.ptr(one),
                               L.stream_ptr()), "sh_bwd")
        ref += one
    # the product returns the two leaves of the reference (band 0, bands 1..) separately
    a = torch.cat(dp._sh_multi_hip(deg, k, dirs_all.contiguous(), None, None, None, None, v_all, 0.5), dim=1)
    b = torch.cat(dp._sh_multi_hip(deg, k, None, means, cams.contiguous(), None, None, v_all, 0.5), dim=1)
    assert a.shape == (n, k, 3)
    assert rel_l2(a, 0.5 * ref) < 1e-6 and rel_l2(b, 0.5 * ref) < 1e-6
    assert float(a[:, (deg + 1) ** 2:, :].abs().max() if (deg + 1) ** 2 < k else 0.0) == 0.0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("fused", [False, True])
def test_exchange_and_reducer_over_single_rank_rccl_group(fused):
    """world_size = 1 RCCL group: the collectives are trivial, but every call the 8-GPU run makes is made here
    (async all_gather from the autograd thread, flat-bucket all_reduce, hook-driven all_reduce)."""
    import torch.distributed as dist
    from sgn_rast import dp, scenes, step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cam, raw = scenes.make_scene("c1", n_override=5000)
        cam.viewmat, cam.cam_pos = cam.viewmat.to(DEV), cam.cam_pos.to(DEV)
        w_img, w_a = step.loss_weights(cam, seed=7, device=DEV)
        Pa = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        step.train_step(Pa, cam, w_img, w_a, fused=fused)
        Pb = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        ex = dp.SHGradExchange(Pb["features_dc"], Pb["features_rest"], force=True).install()
        red = dp.GradAllReducer(list(Pb.values()), big=[Pb["features_rest"]], sh_exchange=ex, force=True)
        # pretend world = 2 with the host-side averaging (what gloo uses; on RCCL the average is ReduceOp.AVG inside the
        # collective, which a 1-rank group cannot show): SUM over 1 rank, then the flat bucket is divided by 2
        red.world, red._avg_in_collective, red._op = 2, False, dist.ReduceOp.SUM
        try:
            step.train_step(Pb, cam, w_img, w_a, fused=fused, reducer=red)
        finally:
            ex.remove()
            red.remove()
        torch.cuda.synchronize()
        for k in Pa:
            scale = 1.0 if k in ("features_dc", "features_rest") else 0.5   # exchange averages over world=1
            assert rel_l2(Pb[k].grad, scale * Pa[k].grad) < 1e-5, k
        # ... and the production form on RCCL: the average taken by the collective itself (1 rank: identity)
        Pc = step.leaf_params({k: v.to(DEV) for k, v in raw.items()})
        red = dp.GradAllReducer(list(Pc.values()), big=[Pc["features_rest"]], force=True, overlap=True)
        assert red._avg_in_collective
        step.train_step(Pc, cam, w_img, w_a, fused=fused, reducer=red)
        red.remove()
        torch.cuda.synchronize()
        assert red.stats["bucket_early"] == 1
        for k in Pa:
            assert rel_l2(Pc[k].grad, Pa[k].grad) < 1e-5, k
    finally:
        dist.destroy_process_group()
