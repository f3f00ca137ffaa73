"""nerfstudio.configs.base_config (public behaviour): dataclass configs that instantiate their `_target`."""
from dataclasses import dataclass
from typing import Any, Type


@dataclass
class PrintableConfig:
    pass


@dataclass
class InstantiateConfig(PrintableConfig):
    _target: Type = None

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)
