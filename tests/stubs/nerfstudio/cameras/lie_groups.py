import torch


def exp_map_SO3xR3(tangent_vector):
    log_rot = tangent_vector[..., 3:]
    nrms = (log_rot * log_rot).sum(-1)
    ang = torch.clamp(nrms, 1e-4).sqrt()
    inv = 1.0 / ang
    fac1, fac2 = inv * ang.sin(), inv * inv * (1.0 - ang.cos())
    skews = torch.zeros(log_rot.shape[:-1] + (3, 3), dtype=log_rot.dtype, device=log_rot.device)
    skews[..., 0, 1], skews[..., 0, 2], skews[..., 1, 0] = -log_rot[..., 2], log_rot[..., 1], log_rot[..., 2]
    skews[..., 1, 2], skews[..., 2, 0], skews[..., 2, 1] = -log_rot[..., 0], -log_rot[..., 1], log_rot[..., 0]
    ret = torch.zeros(log_rot.shape[:-1] + (3, 4), dtype=log_rot.dtype, device=log_rot.device)
    ret[..., :3, :3] = (fac1[..., None, None] * skews + fac2[..., None, None] * (skews @ skews)
                        + torch.eye(3, dtype=log_rot.dtype, device=log_rot.device))
    ret[..., :3, 3] = tangent_vector[..., :3]
    return ret


def exp_map_SE3(tangent_vector):
    raise NotImplementedError("bbox optimiser mode SE3 is not exercised by the tests")
