from dataclasses import dataclass, field
from typing import Literal, Type
import torch
from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class CameraOptimizerConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: CameraOptimizer)
    mode: Literal["off", "SO3xR3", "SE3"] = "off"


class CameraOptimizer(torch.nn.Module):
    def __init__(self, config, num_cameras, device, **kwargs):
        super().__init__()
        assert config.mode == "off", "the reference trains with the camera optimiser off (sgn_config.py:44)"
        self.config, self.num_cameras, self.device = config, num_cameras, device

    def apply_to_camera(self, camera):
        return camera.camera_to_worlds
