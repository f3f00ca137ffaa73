import torch
def to4x4(pose):
    last = torch.zeros_like(pose[..., :1, :]); last[..., :, 3] = 1
    return torch.cat([pose, last], dim=-2)
def multiply(a, b):
    R1, t1, R2, t2 = a[..., :3, :3], a[..., :3, 3:], b[..., :3, :3], b[..., :3, 3:]
    return torch.cat([R1 @ R2, t1 + R1 @ t2], dim=-1)
