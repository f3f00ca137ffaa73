import torch
_C = {"black": [0.0, 0.0, 0.0], "white": [1.0, 1.0, 1.0], "red": [1.0, 0.0, 0.0], "green": [0.0, 1.0, 0.0],
      "blue": [0.0, 0.0, 1.0]}
def get_color(color):
    return torch.tensor(_C[color.lower()]) if isinstance(color, str) else torch.tensor(color)
