import torch
class PeakSignalNoiseRatio(torch.nn.Module):
    def __init__(self, data_range=1.0):
        super().__init__(); self.data_range = data_range
    def forward(self, preds, target):
        mse = torch.mean((preds - target) ** 2)
        return 10 * torch.log10(self.data_range ** 2 / mse)
