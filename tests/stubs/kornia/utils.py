"""kornia.utils.create_meshgrid (public behaviour): [1,H,W,2] grid, last dim = (x, y)."""
import torch
def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=None):
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    base = torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1)   # W x H x 2
    return base.permute(1, 0, 2).unsqueeze(0)                             # 1 x H x W x 2
