from dataclasses import dataclass
import torch


@dataclass
class SceneBox:
    aabb: torch.Tensor


@dataclass
class OrientedBox:
    R: torch.Tensor
    T: torch.Tensor
    S: torch.Tensor

    def within(self, pts):
        R, T, S = self.R.to(pts), self.T.to(pts), self.S.to(pts)
        local = (pts - T) @ R
        return ((local > -S / 2) & (local < S / 2)).all(dim=-1)
