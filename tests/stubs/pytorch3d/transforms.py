"""pytorch3d.transforms.quaternion_multiply (public behaviour): Hamilton product, real part first, result
standardised to a non-negative real part."""
import torch
def quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return torch.stack((ow, ox, oy, oz), -1)
def standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)
def quaternion_multiply(a, b):
    return standardize_quaternion(quaternion_raw_multiply(a, b))
