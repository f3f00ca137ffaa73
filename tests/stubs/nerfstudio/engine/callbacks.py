"""nerfstudio.engine.callbacks (public behaviour)."""
from dataclasses import dataclass
from enum import Enum, auto
from typing import Callable, Dict, List, Optional, Tuple


@dataclass
class TrainingCallbackAttributes:
    optimizers: Optional[object] = None
    grad_scaler: Optional[object] = None
    pipeline: Optional[object] = None
    trainer: Optional[object] = None


class TrainingCallbackLocation(Enum):
    BEFORE_TRAIN_ITERATION = auto()
    AFTER_TRAIN_ITERATION = auto()
    AFTER_TRAIN = auto()


class TrainingCallback:
    def __init__(self, where_to_run: List[TrainingCallbackLocation], func: Callable,
                 update_every_num_iters: Optional[int] = None, iters: Optional[Tuple[int, ...]] = None,
                 args: Optional[List] = None, kwargs: Optional[Dict] = None):
        self.where_to_run, self.func = where_to_run, func
        self.update_every_num_iters, self.iters = update_every_num_iters, iters
        self.args = args if args is not None else []
        self.kwargs = kwargs if kwargs is not None else {}

    def run_callback(self, step: int) -> None:
        if self.update_every_num_iters is not None:
            if step % self.update_every_num_iters == 0:
                self.func(*self.args, **self.kwargs, step=step)
        elif self.iters is not None:
            if step in self.iters:
                self.func(*self.args, **self.kwargs, step=step)
        else:
            self.func(*self.args, **self.kwargs, step=step)

    def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
        if location in self.where_to_run:
            self.run_callback(step=step)
