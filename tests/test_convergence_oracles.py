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
