"""CPU: the convergence-parity harness itself (tests/convergence.py) on the two INDEPENDENT oracles — C scalar
(analytic backward with upstream's quirks) vs. pure-PyTorch autograd — so the 0.05 dB criterion of the GPU test
(tests/test_gpu_convergence.py, VERDICT r01 row g) is known to be meaningful: two correct implementations that differ
in rounding, summation order and backward clamp stay well inside it, and training really improves the image."""
import pytest

import convergence as C


@pytest.mark.timeout(600)
def test_two_oracles_train_to_the_same_psnr():
    import oracle_ops
    from oracle import torch_oracle as TO
    cam, truth, start, gt = C.make_problem(n=4000, size=64)
    steps = 25
    a, Pa = C.fit(start, cam, gt, steps, ops=oracle_ops, loss_fn=C.oracle_loss)
    b, Pb = C.fit(start, cam, gt, steps, ops=TO, loss_fn=C.oracle_loss)
    assert a[-1] > a[0] + 1.0, (a[0], a[-1])                      # the fit makes progress (dB)
    worst = max(abs(x - y) for x, y in zip(a, b))
    assert worst <= 0.05, worst


@pytest.mark.timeout(900)
def test_schedule_yardstick_two_runs_of_the_same_oracle():
    """The schedule-shaped fit of tests/test_gpu_convergence_schedule.py (scaled to 12 k Gaussians, through the first
    densification) run twice on the SAME C oracle, the two runs differing only in the float summation order of the
    compositing backward (3 vs 8 host threads over the pixel rows): what two correct implementations can be expected
    to agree to.  Eight-view evaluation PSNR: within 0.05 dB at every checkpoint; single-view training PSNR: within
    0.05 dB before the densification; afterwards the fresh all-zero Adam state of the new Gaussians (eps 1e-15: first
    updates are lr x sign(noise)) lets single views drift apart by up to ~0.1 dB — the GPU test bounds the mean over one
    pass of the views there."""
    import torch

    import oracle_ops
    from helpers import TorchStats
    from oracle import c_oracle as CO
    from sgn_rast import step
    cfg = dict(C.SCHEDULE)
    cfg.update(n=12_000)

    def render_band(params, cam):
        oracle_ops.PIXEL_ROWS = cfg["band"]
        try:
            with torch.no_grad():
                return step.render(step.leaf_params(params), cam, 3, 16, ops=oracle_ops,
                                   caller_syncs=False).rgb[cfg["band"][0]:cfg["band"][1]]
        finally:
            oracle_ops.PIXEL_ROWS = None

    truth, start, gts = C.schedule_problem(render_band, cfg)
    runs = []
    old = CO.THREADS
    try:
        for th in (3, 8):
            CO.THREADS = th
            runs.append(C.fit_schedule(start, gts, ops=oracle_ops, loss_fn=C.oracle_loss, stats=TorchStats(), cfg=cfg,
                                       steps=300))
    finally:
        CO.THREADS = old
    a, b = runs
    assert [e[:2] for e in a["events"]] == [(200, "densify")] == [e[:2] for e in b["events"]]
    assert a["psnr"][149] > a["psnr"][0] + 5.0
    d = [abs(x - y) for x, y in zip(a["psnr"], b["psnr"])]
    worst_eval = max(abs(x[1] - y[1]) for x, y in zip(a["eval"], b["eval"]))
    assert max(d[:200]) <= 0.05 and worst_eval <= 0.05, (max(d[:200]), worst_eval)
    assert max(d[200:]) <= 0.15, max(d[200:])
