import torch
class LearnedPerceptualImagePatchSimilarity(torch.nn.Module):
    """Needs pretrained network weights (no network here): constructible, not callable."""
    def __init__(self, normalize=False):
        super().__init__()
    def forward(self, a, b):
        raise NotImplementedError("LPIPS needs pretrained weights; not available offline")
