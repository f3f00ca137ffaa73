"""nerfstudio.models.base_model (public behaviour): Model / ModelConfig skeleton."""
from dataclasses import dataclass, field
from typing import Dict, Type
import torch
from torch import nn
from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class ModelConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: Model)
    enable_collider: bool = True
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
    eval_num_rays_per_chunk: int = 4096
    prompt: object = None


class Model(nn.Module):
    config: ModelConfig

    def __init__(self, config, scene_box=None, num_train_data=0, **kwargs):
        super().__init__()
        self.config, self.scene_box, self.num_train_data, self.kwargs = config, scene_box, num_train_data, kwargs
        self.render_aabb, self.collider = None, None
        self.populate_modules()
        self.callbacks = None
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self):
        pass

    def forward(self, ray_bundle):
        return self.get_outputs(ray_bundle)
