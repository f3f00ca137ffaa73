"""nerfstudio.cameras.cameras.Cameras: the fields and methods the reference's model files touch."""
import torch


class Cameras:
    def __init__(self, camera_to_worlds, fx, fy, cx, cy, width, height, times=None):
        c2w = torch.as_tensor(camera_to_worlds, dtype=torch.float32)
        self.camera_to_worlds = c2w if c2w.dim() == 3 else c2w[None]
        n = self.camera_to_worlds.shape[0]
        col = lambda v, dt: torch.as_tensor(v, dtype=dt).reshape(-1, 1).expand(n, 1).clone()
        self.fx, self.fy = col(fx, torch.float32), col(fy, torch.float32)
        self.cx, self.cy = col(cx, torch.float32), col(cy, torch.float32)
        self.width, self.height = col(width, torch.int64), col(height, torch.int64)
        self.times = None if times is None else col(times, torch.float64)

    @property
    def shape(self):
        return self.camera_to_worlds.shape[:-2]

    @property
    def device(self):
        return self.camera_to_worlds.device

    def to(self, device):
        out = Cameras.__new__(Cameras)
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
        return out

    def rescale_output_resolution(self, scaling_factor):
        s = float(scaling_factor)
        self.fx, self.fy, self.cx, self.cy = self.fx * s, self.fy * s, self.cx * s, self.cy * s
        self.height = (self.height * s).to(torch.int64)
        self.width = (self.width * s).to(torch.int64)
